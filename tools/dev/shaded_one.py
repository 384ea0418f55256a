#!/usr/bin/env python3
"""Developer tool: ONE workload of tools/bench_shaded.py in ONE gradient mode, a few frames (for rocprofv3 runs).
usage: tools/dev/shaded_one.py <c32l4_fourier_snakealt|c32l4_grid16_relu|c64l6_grid16_relu> <0|1|2> [frames]"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fvsrn_amd import synthetic, capi, volnet_io  # noqa: E402

NETS = {"c32l4_fourier_snakealt": dict(C=32, layers=4, activation="SnakeAlt"),
        "c32l4_grid16_relu": dict(C=32, layers=4, activation="ReLU", grid=(16, 16)),
        "c64l6_grid16_relu": dict(C=64, layers=6, activation="ReLU", grid=(16, 32))}
name, mode = sys.argv[1], int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W = H = 1024
vn = synthetic.random_network(output_mode="density:direct", seed=1234, box_min=(-0.5, -0.5, -0.5), grid_scale=0.01, **NETS[name])
net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
kw = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1 / 512, early_out=True,
          tf_kind=capi.TF_IDENTITY, tf_scale_absorption=10.0, tf_scale_emission=1.0, gradient_mode=mode)
if mode:
    kw.update(finite_differences_stepsize=1 / 256,
              brdf=dict(enable_phong=True, ambient=0.2, specular=0.4, magnitude_center=0.6, magnitude_radius=0.5, specular_exponent=8,
                        light_type=0, light=tuple(float(v) for v in eye)))
scene = capi.Scene(**kw)
out = torch.zeros((1, 8, H, W), dtype=torch.float32, device="cuda")
stats = torch.zeros(2, dtype=torch.int64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
scene.render(net, W, H, out=out, stats=stats)
torch.cuda.synchronize()
stats.zero_()
e0.record()
for _ in range(frames):
    scene.render(net, W, H, out=out, stats=stats)
e1.record()
torch.cuda.synchronize()
print("%s mode %d: %.3f ms / frame, %.0f samples / frame, kernel %s" % (name, mode, e0.elapsed_time(e1) / frames, float(stats[0]) / frames, net.kernel_name(mode == 0)))
