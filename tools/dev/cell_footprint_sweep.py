"""Box script: cell-table path against the gather path as a function of the footprint of an 8 x 8 pixel tile in grid cells
(f = 8 pixels x pixel size at the box centre x (N - 1) / box size).  Prints one line per (network, grid, image)."""
import os, sys, time, math
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa
from fvsrn_amd import capi, volnet_io, synthetic  # noqa
kw = bench.build_scene_kwargs(capi, 0.7, 1 / 512, False)
dist, fov = 1.6, math.radians(45.0)
for C, layers in ((32, 4), (64, 6)):
    for res in (16, 24, 32, 48, 64):
        vn = synthetic.random_network(C=C, layers=layers, activation="ReLU", param=1.0, output_mode="density:direct", grid=(16, res), seed=1234, box_min=(-0.5, -0.5, -0.5), grid_scale=0.01)
        net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
        for W in (512, 1024, 2048):
            if W == 2048 and C == 64:
                continue
            t = {}
            for name, opt in (("cells", 1), ("gather", 0), ("auto", -1)):
                sc = capi.Scene(**kw).set_option("cell_table", opt)
                sc.render(net, W, W)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 3 if C == 64 else 6
                for _ in range(n):
                    sc.render(net, W, W)
                torch.cuda.synchronize()
                t[name] = (time.perf_counter() - t0) / n * 1e3
                t[name + "_used"] = sc.last_render_info()["cell_table"]
            f = 8 * (2 * math.tan(fov / 2) / W * dist) * (res - 1)
            print("%dx%d grid %2d^3 image %4d^2: footprint %.2f cells  cells %.3f ms  gather %.3f ms  ratio %.2f  automatic: %s %.3f ms" % (C, layers, res, W, f, t["cells"], t["gather"], t["cells"] / t["gather"], "cells" if t["auto_used"] else "gather", t["auto"]), flush=True)
        del net
