"""Shared helpers of the test-suite: golden fixtures -> .volnet -> oracle / HIP path."""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import fvsrn_amd  # noqa: E402  (alias of the package directory fv-srn_amd/)
from fvsrn_amd import volnet_io  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def golden_names(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    meta = json.loads(bytes(d.pop("meta")).decode())
    return d, meta


def golden_to_volnet(d, meta, *, encoding=volnet_io.ENC_FLOAT, box_min=(0.0, 0.0, 0.0), box_size=(1.0, 1.0, 1.0)):
    """The export_to_pyrenderer hand-off (reference network.py:798-897) on the arrays of a fixture."""
    n = len(meta["layers"].split(":")) + 1
    weights = [d["W%d" % i] for i in range(n)]
    biases = [d["b%d" % i] for i in range(n)]
    time_grids = ensemble_grids = None
    if "grid" in d:  # static grid = one time key-frame (network.py:856-864)
        time_grids = [d["grid"][0]]
    if "grid_time" in d:
        time_grids = [g for g in d["grid_time"]]
    if "grid_ensemble" in d:
        ensemble_grids = [g for g in d["grid_ensemble"]]
        if time_grids is None:
            time_grids = []
    return volnet_io.build_volnet(
        fourier_B=d["B"], weights=weights, biases=biases, activation=meta["activation"],
        activation_param=meta["activation_param"], output_mode=meta["output_mode"], box_min=box_min, box_size=box_size,
        premultiplied=True, time_grids=time_grids, ensemble_grids=ensemble_grids, grid_encoding=encoding,
        has_time=meta.get("use_time_direct", False), has_direction=meta.get("use_direction", False))


from fvsrn_amd.synthetic import random_network  # noqa: E402,F401  (lives in the package: bench.py uses it too)


def lz4_messages(data: bytes, message: int = 65536) -> bytes:
    """Test-side LZ4 encoder in the framing of compressed .cvol bodies (int32 size + one LZ4 block per message of `message` bytes, matches
    may reach back into earlier messages: a dependent-block stream like the reference's lz4cpp wrapper writes).  Greedy 4-byte hash matcher;
    the last 5 bytes of a block are literals and no match starts in its last 12 bytes, as the block format requires."""
    import struct
    out, table = bytearray(), {}
    for base in range(0, len(data), message):
        end = min(base + message, len(data))
        blk, i, anchor = bytearray(), base, base

        def emit(lit_from, lit_to, match_len, offset):
            lit = lit_to - lit_from
            tok_l, tok_m = min(lit, 15), (min(match_len - 4, 15) if match_len else 0)
            blk.append((tok_l << 4) | tok_m)
            if lit >= 15:
                r = lit - 15
                while r >= 255:
                    blk.append(255); r -= 255
                blk.append(r)
            blk.extend(data[lit_from:lit_to])
            if match_len:
                blk.extend(struct.pack("<H", offset))
                if match_len - 4 >= 15:
                    r = match_len - 4 - 15
                    while r >= 255:
                        blk.append(255); r -= 255
                    blk.append(r)

        while i + 12 < end:
            key = data[i:i + 4]
            j = table.get(key)
            table[key] = i
            if j is not None and 0 < i - j <= 65535:
                n = 4
                while i + n < end - 5 and data[j + n] == data[i + n]:
                    n += 1
                emit(anchor, i, n, i - j)
                i += n
                anchor = i
            else:
                i += 1
        emit(anchor, end, 0, 0)
        out += struct.pack("<i", len(blk)) + blk
    return bytes(out)
