#!/usr/bin/env python3
"""Throughput of the gradient entry points of IVolumeInterpolation on one MI355X: fvsrn_evaluate_points_adjoint (analytic gradients,
GRADIENT_MODE_ADJOINT_METHOD) and fvsrn_evaluate_points with FVSRN_EVAL_WITH_PREDICTED_CURVATURE, N random positions resident in HBM."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fvsrn_amd import synthetic, capi, volnet_io  # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
    pos = torch.rand(n, 3, device="cuda")
    for name, kw in [("c32l4_fourier_snakealt", dict(C=32, layers=4, activation="SnakeAlt")),
                     ("c32l4_grid16_relu", dict(C=32, layers=4, activation="ReLU", grid=(16, 16))),
                     ("c64l6_grid16_relu", dict(C=64, layers=6, activation="ReLU", grid=(16, 32)))]:
        net = capi.Network.from_volnet(volnet_io.save_volnet(synthetic.random_network(output_mode="density", seed=1234, **kw)))
        ms_f, _ = timed(lambda: net.evaluate(pos))
        ms_a, (v, g) = timed(lambda: net.evaluate_with_adjoint_gradient(pos))
        assert bool(torch.isfinite(g).all())
        print(json.dumps({"workload": "evaluate_points_adjoint:" + name, "points": n, "ms_forward": ms_f, "ms_adjoint": ms_a,
                          "points_per_s_adjoint": n / ms_a * 1e3, "adjoint_over_forward": ms_a / ms_f}))
    net = capi.Network.from_volnet(volnet_io.save_volnet(synthetic.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="densitycurvature", seed=1234)))
    ms_c, (d, g, c) = timed(lambda: net.evaluate_with_gradients_and_curvature(pos))
    assert tuple(c.shape) == (n, 2) and bool(torch.isfinite(c).all())
    print(json.dumps({"workload": "evaluate_with_gradients_and_curvature:c32l4_fourier_snakealt", "points": n, "ms": ms_c, "points_per_s": n / ms_c * 1e3}))


if __name__ == "__main__":
    main()
