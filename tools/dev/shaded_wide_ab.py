#!/usr/bin/env python3
"""Developer A/B (GPU box): shaded frames (finite differences + Phong) of 80 / 128-wide networks, one against two waves per SIMD (FVSRN_LIBRARY selects the variant library)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fvsrn_amd import capi, synthetic, volnet_io
for C, L in ((128, 2), (80, 3)):
    for grid, opt in (((16, 32), None), ((16, 32), 0), (None, None)):
        vn = synthetic.random_network(C=C, layers=L, activation="ReLU", output_mode="density:direct", grid=grid, seed=1234, box_min=(-0.5, -0.5, -0.5), fourier_std=0.5, grid_scale=0.01)
        net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
        eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.3, 1.6)
        sc = capi.Scene(eye=eye, right=right, up=up, fov_y_radians=math.radians(45), stepsize=1 / 256, early_out=True, tf_kind=capi.TF_IDENTITY, tf_scale_absorption=10.0, tf_scale_emission=1.0,
                        gradient_mode=capi.GRADIENT_FINITE_DIFFERENCES, finite_differences_stepsize=1 / 256, brdf=dict(enable_phong=True, ambient=0.2, specular=0.4))
        if opt is not None:
            sc.set_option("cell_table", opt)
        out = torch.zeros((1, 8, 512, 512), device="cuda")
        for _ in range(2):
            sc.render(net, 512, 512, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            sc.render(net, 512, 512, out=out)
        e1.record(); torch.cuda.synchronize()
        print("%dx%d grid %-9s cell_table %-4s shaded (finite differences) 512^2 x 256: %.3f ms  %s" % (C, L, grid, opt, e0.elapsed_time(e1) / 3, sc.last_kernel_name()), flush=True)
