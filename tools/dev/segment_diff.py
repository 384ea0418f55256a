"""Box script: how far do frames rendered with different depth-segment counts / as stripes differ (rotation restart points)?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa
from fvsrn_amd import capi, volnet_io, tiles  # noqa
for C, layers, grid, act in ((32, 4, None, "ReLU"), (32, 4, None, "SnakeAlt"), (32, 4, (16, 16), "ReLU"), (64, 6, (16, 32), "ReLU")):
    vn = bench.bench_network(C, layers, grid, act)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    kw = bench.build_scene_kwargs(capi, 0.7, 1 / 512, False)
    W = H = 512
    ref = torch.nan_to_num(capi.Scene(**kw).set_option("depth_segments", 1).render(net, W, H)[0], nan=0.0).clone()
    out = []
    for K in (2, 4, 8):
        img = torch.nan_to_num(capi.Scene(**kw).set_option("depth_segments", K).render(net, W, H)[0], nan=0.0)
        out.append("K=%d %.2e" % (K, float((img[:4] - ref[:4]).abs().max())))
    rows = tiles.owned_rows(H, 16, 3, 8)
    st = torch.nan_to_num(capi.render_stripes(capi.Scene(**kw), net, W, H, 16, 3, 8), nan=0.0)
    out.append("stripes(rank 3 of 8) %.2e" % float((st[:4] - ref[:4][:, rows]).abs().max()))
    exact = torch.nan_to_num(capi.Scene(**kw).set_option("depth_segments", 1).set_option("fourier_resync", 1).render(net, W, H)[0], nan=0.0)
    out.append("rotation vs per-step features %.2e" % float((exact[:4] - ref[:4]).abs().max()))
    print("%dx%d %s %s: %s" % (C, layers, grid, act, "  ".join(out)))
