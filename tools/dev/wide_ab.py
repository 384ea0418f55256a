#!/usr/bin/env python3
"""Developer A/B (GPU box): frame time of wide study networks through the ctypes binding, so that FVSRN_LIBRARY selects the library variant (tools/variant.sh):
112 / 128 channels with one or two waves per SIMD (-DFVSRN_WAVES_PER_EU_WIDE), by latent-grid path, and evaluate_points."""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fvsrn_amd import capi, synthetic, volnet_io
NETS = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]] or [(128, 2), (112, 2), (128, 3)]
for C, L in NETS:
    for act in ("ReLU", "SnakeAlt"):
        for grid, enc, opt in (((16, 32), 0, None), ((16, 32), 0, 0), ((16, 32), 2, None), (None, 0, None)):
            vn = synthetic.random_network(C=C, layers=L, activation=act, output_mode="density:direct", grid=grid, seed=1234, box_min=(-0.5, -0.5, -0.5), fourier_std=0.5, grid_scale=0.01,
                                          encoding=enc)
            net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
            eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.3, 1.6)
            sc = capi.Scene(eye=eye, right=right, up=up, fov_y_radians=math.radians(45), stepsize=1 / 512, early_out=False, tf_kind=capi.TF_IDENTITY, tf_scale_absorption=10.0, tf_scale_emission=1.0)
            if opt is not None:
                sc.set_option("cell_table", opt)
            out = torch.zeros((1, 8, 1024, 1024), device="cuda")
            for _ in range(4):
                sc.render(net, 1024, 1024, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                sc.render(net, 1024, 1024, out=out)
            e1.record(); torch.cuda.synchronize()
            pos = torch.rand(1 << 22, 3, device="cuda")
            o = net.evaluate(pos)
            for _ in range(3):
                o = net.evaluate(pos, out=o)
            torch.cuda.synchronize()
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
            for _ in range(5):
                o = net.evaluate(pos, out=o)
            g1.record(); torch.cuda.synchronize()
            print("%dx%d %-8s grid %-9s enc %d cell_table %-4s render %.3f ms  %-44s evaluate 2^22: %.3f ms" % (
                C, L, act, grid, enc, opt, e0.elapsed_time(e1) / 6, sc.last_kernel_name()[:44], g0.elapsed_time(g1) / 5), flush=True)
