#!/usr/bin/env python3
"""Throughput of fvsrn_evaluate_points (IVolumeInterpolation::evaluate, reference volume_interpolation.cpp:26-127) on one
MI355X: N random positions resident in HBM -> network values.  Prints one JSON line per configuration."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fvsrn_amd import synthetic as util  # noqa: E402
from fvsrn_amd import capi, volnet_io  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
    pos = torch.rand(n, 3, device="cuda")
    for name, kw in [("c32l4_fourier_relu", dict(C=32, layers=4, activation="ReLU")),
                     ("c32l4_fourier_snakealt", dict(C=32, layers=4, activation="SnakeAlt")),
                     ("c32l4_grid16_relu", dict(C=32, layers=4, activation="ReLU", grid=(16, 16))),
                     ("c64l6_grid16_relu", dict(C=64, layers=6, activation="ReLU", grid=(16, 32)))]:
        vn = util.random_network(output_mode="density:direct", seed=1234, **kw)
        net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
        out = net.evaluate(pos)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            out = net.evaluate(pos)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        info = net.info()
        print(json.dumps({"workload": "evaluate_points:" + name, "points": n, "ms": ms, "points_per_s": n / ms * 1e3,
                          "kernel": net.kernel_name(False), "algorithmic_tflops": info.flops_per_sample * n / ms / 1e9,
                          "hbm_GBps_algorithmic": 16.0 * n / ms / 1e6}))


if __name__ == "__main__":
    main()
