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


MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBPS = 8000.0


def main():
    """One JSON line per network and batch size.  evaluate_points sits between two rooflines: 12 B in + 4 B out per point against
    HBM, the network's algorithmic FLOP per point against the matrix cores -- both are reported, the larger fraction names the bound."""
    sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1 << 24, 1 << 26]
    only = [a for a in sys.argv[1:] if not a.isdigit() and a != "half"]  # workload names (default: all four)
    # "half": fp16 positions in, fp16 values out (fvsrn_evaluate_points_half, 8 B per point); the default is the reference's fp32 tensors (16 B)
    half = "half" in sys.argv[1:]
    for n in sizes:
        pos = torch.rand(n, 3, device="cuda")
        if half:
            pos = pos.half()
        for name, kw in [("c32l4_fourier_relu", dict(C=32, layers=4, activation="ReLU")),
                         ("c32l4_fourier_snakealt", dict(C=32, layers=4, activation="SnakeAlt")),
                         ("c32l4_grid16_relu", dict(C=32, layers=4, activation="ReLU", grid=(16, 16))),
                         ("c64l6_grid16_relu", dict(C=64, layers=6, activation="ReLU", grid=(16, 32))),
                         # (only when named: the LDS kernels without a latent grid)
                         ("c64l6_fourier_relu", dict(C=64, layers=6, activation="ReLU")),
                         ("c48l5_fourier_snakealt", dict(C=48, layers=5, activation="SnakeAlt"))]:
            if (only and name not in only) or (not only and name in ("c64l6_fourier_relu", "c48l5_fourier_snakealt")):
                continue
            vn = util.random_network(output_mode="density:direct", seed=1234, **kw)
            net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
            out = net.evaluate(pos)
            torch.cuda.synchronize()
            reps = 20 if n <= (1 << 24) else 6
            # clock spin-up like bench.py (r04: the shader clock of an idle MI355X needs a few hundred ms of load to settle -- three warm-up calls
            # of 0.2 ms left the r03 numbers 15 - 20 % below the steady state), then the best of three timed rounds
            import time
            t_end = time.perf_counter() + 0.25
            while time.perf_counter() < t_end:
                for _ in range(8):
                    out = net.evaluate(pos, out=out)
                torch.cuda.synchronize()
            ms = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    out = net.evaluate(pos, out=out)
                e1.record()
                torch.cuda.synchronize()
                ms = min(ms, e0.elapsed_time(e1) / reps)
            info = net.info()
            tflops = info.flops_per_sample * n / ms / 1e9
            bytes_per_point = 8 if half else 16
            gbps = bytes_per_point * n / ms / 1e6
            # the issue roofline of bench.py (vector issue port: MFMA 8, transcendental 8.4, convert 4.4, other 2.75 cycles per wave instruction) from the
            # committed PMC profile of this kernel (profiles/r*/evaluate_points_<name>_*pmc.csv), where there is one
            import importlib.util
            spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
            bench = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(bench)
            issue = bench.issue_roofline(bench.pmc_counters("evaluate_points_" + name))
            print(json.dumps({"workload": "evaluate_points:" + name + ("_half_io" if half else ""), "points": n, "ms": ms, "points_per_s": n / ms * 1e3, "issue": issue,
                              "kernel": net.kernel_name(False),
                              "roofline": {"bound": "mfma" if tflops / MFMA_F16_PEAK_TFLOPS > gbps / HBM_PEAK_GBPS else "hbm",
                                           "mfma": {"achieved": tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / MFMA_F16_PEAK_TFLOPS,
                                                    "flops_per_point": info.flops_per_sample},
                                           "hbm": {"achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                                                   "bytes_per_point": bytes_per_point}}}), flush=True)
        del pos


if __name__ == "__main__":
    main()
