#!/usr/bin/env python3
"""Fixture of the one volume file the reference snapshot holds: applications/volumes/RichtmyerMeshkov/ppm-t0020.cvol (old "cvol" format,
256^3 bytes, LZ4-compressed).  Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_cvol_fixture.py
Decodes the file with the pure-Python decoder below -- independent of the C++ loader it pins (fvsrn_volume_load_cvol) -- and writes
tests/golden/cvol_ppm_t0020.npz: resolution, voxel size, min / max / mean, a 64-bin histogram, the crc32 of the decoded bytes and a 32^3
block-mean downsample (the GPU test renders that).  Data only; the reference file itself is not copied.

Framing (recovered from the file, the reference's lz4cpp wrapper is an empty submodule): after the 64-byte header
    repeat:  int32 compressed size | one LZ4 block of a dependent-block stream, 65 536 decoded bytes per message
256 messages decode to 256^3 bytes and consume the file to its last byte."""
import os
import struct
import sys
import zlib

import numpy as np

SRC = "/root/reference/applications/volumes/RichtmyerMeshkov/ppm-t0020.cvol"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cvol_ppm_t0020.npz")


def lz4_block(src: bytes, out: bytearray) -> None:
    i, n = 0, len(src)
    while i < n:
        tok = src[i]; i += 1
        lit = tok >> 4
        if lit == 15:
            while True:
                b = src[i]; i += 1; lit += b
                if b != 255:
                    break
        out += src[i:i + lit]; i += lit
        if i >= n:
            break
        off = src[i] | (src[i + 1] << 8); i += 2
        ml = tok & 15
        if ml == 15:
            while True:
                b = src[i]; i += 1; ml += b
                if b != 255:
                    break
        ml += 4
        start = len(out) - off
        assert 0 < off <= len(out)
        if off >= ml:
            out += out[start:start + ml]
        else:
            out += (bytes(out[start:start + off]) * ((ml + off - 1) // off))[:ml]


def decode(path):
    d = open(path, "rb").read()
    assert d[:4] == b"cvol"
    sx, sy, sz = struct.unpack("<QQQ", d[4:28])
    voxel = struct.unpack("<ddd", d[28:52])
    dtype, = struct.unpack("<I", d[52:56])
    assert dtype == 0 and d[56] == 1, "expected a compressed uchar volume"
    body, pos, out, sizes = memoryview(d)[64:], 0, bytearray(), []
    while len(out) < sx * sy * sz:
        cs, = struct.unpack("<i", body[pos:pos + 4]); pos += 4
        before = len(out)
        lz4_block(bytes(body[pos:pos + cs]), out); pos += cs
        sizes.append(len(out) - before)
    assert pos == len(body) and len(out) == sx * sy * sz and set(sizes) == {65536}
    vol = np.frombuffer(bytes(out), np.uint8).reshape(sz, sy, sx).transpose(2, 1, 0)  # (X,Y,Z); file order: x fastest
    return vol, voxel, zlib.crc32(bytes(out))


def main():
    vol, voxel, crc = decode(SRC)
    small = vol.reshape(32, 8, 32, 8, 32, 8).astype(np.float64).mean(axis=(1, 3, 5)) / 255.0
    np.savez_compressed(OUT, resolution=np.array(vol.shape, np.int32), voxel_size=np.array(voxel, np.float64), minimum=int(vol.min()), maximum=int(vol.max()),
                        mean=float(vol.mean()), histogram=np.bincount(vol.reshape(-1) >> 2, minlength=64).astype(np.int64), crc32=np.uint32(crc),
                        block_mean_32=small.astype(np.float32))
    print("wrote", OUT, vol.shape, vol.min(), vol.max(), vol.mean(), hex(crc))


if __name__ == "__main__":
    if "--check" in sys.argv:
        vol, voxel, crc = decode(SRC)
        ref = np.load(OUT)
        assert int(ref["crc32"]) == crc and np.array_equal(ref["histogram"], np.bincount(vol.reshape(-1) >> 2, minlength=64))
        print("cvol fixture reproduces")
    else:
        main()
