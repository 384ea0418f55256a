#!/usr/bin/env python3
"""Frame time of the shaded renderer (Phong BRDF, normals from the network) on one MI355X: GRADIENT_MODE_FINITE_DIFFERENCES (6 extra network
evaluations per visible sample) against GRADIENT_MODE_ADJOINT_METHOD (one pass with three tangent tiles), 1024^2, step 1/512, early-out on."""
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fvsrn_amd import synthetic, capi, volnet_io  # noqa: E402


def main():
    W = H = 1024
    for name, kw in [("c32l4_fourier_snakealt", dict(C=32, layers=4, activation="SnakeAlt")),
                     ("c32l4_grid16_relu", dict(C=32, layers=4, activation="ReLU", grid=(16, 16))),
                     ("c64l6_grid16_relu", dict(C=64, layers=6, activation="ReLU", grid=(16, 32)))]:
        vn = synthetic.random_network(output_mode="density:direct", seed=1234, box_min=(-0.5, -0.5, -0.5), grid_scale=0.01, **kw)
        net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
        eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
        base = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1 / 512, early_out=True,
                    tf_kind=capi.TF_IDENTITY, tf_scale_absorption=10.0, tf_scale_emission=1.0)
        row = {"workload": "shaded:" + name}
        for mode, label in ((0, "unshaded"), (1, "finite_differences"), (2, "adjoint")):
            kwargs = dict(base, gradient_mode=mode)
            if mode:
                kwargs.update(finite_differences_stepsize=1 / 256,
                              brdf=dict(enable_phong=True, ambient=0.2, specular=0.4, magnitude_center=0.6, magnitude_radius=0.5, specular_exponent=8,
                                        light_type=0, light=tuple(float(v) for v in eye)))
            scene = capi.Scene(**kwargs)
            out = torch.zeros((1, 8, H, W), dtype=torch.float32, device="cuda")
            stats = torch.zeros(2, dtype=torch.int64, device="cuda")
            for _ in range(2):
                scene.render(net, W, H, out=out, stats=stats)
            torch.cuda.synchronize()
            stats.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 4
            e0.record()
            for _ in range(reps):
                scene.render(net, W, H, out=out, stats=stats)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            row[label + "_ms"] = ms
            row[label + "_Gsamples_per_s"] = float(stats[0]) / reps / ms / 1e6
        print(json.dumps(row))


if __name__ == "__main__":
    main()
