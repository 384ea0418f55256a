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
