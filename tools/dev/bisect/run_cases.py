#!/usr/bin/env python3
"""Developer tool (hazard bisect, r04): launch-to-launch bit-equality of the kernels of ONE translation unit under test, for the library
FVSRN_LIBRARY points at.  One line per case: how many of N launches differ from the first.
usage: run_cases.py <stripe|small|lds32> [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from fvsrn_amd import capi, synthetic, volnet_io  # noqa: E402

FAMILY = sys.argv[1] if len(sys.argv) > 1 else "stripe"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
kw = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1 / 128, early_out=False,
          tf_kind=capi.TF_IDENTITY, tf_scale_absorption=20.0, tf_scale_emission=1.0)
W, H = 1024, 512
G = dict(grid=(16, 8))
if FAMILY == "stripe":
    CASES = [("relu frame", dict(C=64, layers=3, activation="ReLU", **G), "frame", dict(overlap_kernel=1)),
             ("relu frame notpers", dict(C=64, layers=3, activation="ReLU", **G), "frame", dict(overlap_kernel=1, persistent=0)),
             ("relu frame 1wave/simd", dict(C=64, layers=3, activation="ReLU", **G), "frame", dict(overlap_kernel=1, waves_per_block=4, max_blocks_per_cu=1)),
             ("snakealt stripes", dict(C=64, layers=3, activation="SnakeAlt", **G), "stripes", dict(overlap_kernel=1)),
             ("snakealt frame notpers", dict(C=64, layers=3, activation="SnakeAlt", **G), "frame", dict(overlap_kernel=1, persistent=0)),
             ("snakealt frame 1wave/simd", dict(C=64, layers=3, activation="SnakeAlt", **G), "frame", dict(overlap_kernel=1, waves_per_block=4, max_blocks_per_cu=1))]
elif FAMILY == "small":
    CASES = [("relu resident", dict(C=32, layers=4, activation="ReLU", **G), "frame", {}),
             ("snakealt resident", dict(C=32, layers=4, activation="SnakeAlt", **G), "frame", {}),
             ("relu resident notpers", dict(C=32, layers=4, activation="ReLU", **G), "frame", dict(persistent=0))]
else:
    CASES = [("snakealt lds32", dict(C=32, layers=4, activation="SnakeAlt", **G), "frame", dict(small_kernel=0)),
             ("snakealt lds32 stripes notpers", dict(C=32, layers=4, activation="SnakeAlt", **G), "stripes", dict(small_kernel=0, persistent=0, unit_quota=0))]
res = []
for name, net_kw, what, opts in CASES:
    vn = synthetic.random_network(output_mode="density", seed=62, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4, grid_scale=0.3, **net_kw)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    scene = capi.Scene(**kw).set_option("depth_segments", 1)
    for k, v in opts.items():
        scene.set_option(k, v)
    first, bad, nvals, lanes = None, 0, 0, set()
    for i in range(N):
        img = capi.render_stripes(scene, net, W, H, 16, 1, 2) if what == "stripes" else scene.render(net, W, H)[0]
        img = torch.nan_to_num(img, nan=-7.0).clone()
        if first is None:
            first = img
        elif not torch.equal(first, img):
            d = (first != img).nonzero().cpu().numpy()
            bad += 1
            nvals += len(d)
            lanes |= set(int((y % 8) * 8 + (x % 8)) for _, y, x in d)
    res.append("%s %d/%d%s" % (name, bad, N - 1, (" (%d vals, lanes %d-%d)" % (nvals, min(lanes), max(lanes))) if bad else ""))
print(" | ".join(res))
