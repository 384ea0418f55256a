#!/usr/bin/env python3
"""Developer tool (hazard bisect, r04): the kernel under test at ONE wave per SIMD, next to aggressor waves of one instruction class
(aggressor.hip) on a second stream.  One line per class: launches of N that differ from a reference rendered alone.
usage: run_aggr.py [N] [blocks]          (FVSRN_LIBRARY selects the build under test, AGGR_KINDS=2,14 a subset of the classes)"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from fvsrn_amd import capi, synthetic, volnet_io  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BLOCKS = int(sys.argv[2]) if len(sys.argv) > 2 else 512
agg = ctypes.CDLL(os.path.join(ROOT, "tools", "dev", "bin", "libaggressor.so"))
agg.aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
agg.aggressor_name.restype = ctypes.c_char_p
eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
kw = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1 / 128, early_out=False,
          tf_kind=capi.TF_IDENTITY, tf_scale_absorption=20.0, tf_scale_emission=1.0)
W, H = 1024, 512
sB = torch.cuda.Stream()
for act in ("ReLU", "SnakeAlt"):
    vn = synthetic.random_network(output_mode="density", seed=62, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4, grid_scale=0.3, C=64, layers=3,
                                  activation=act, grid=(16, 8))
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    scene = capi.Scene(**kw).set_option("depth_segments", 1)
    for k, v in dict(overlap_kernel=1, waves_per_block=4, max_blocks_per_cu=1).items():
        scene.set_option(k, v)
    ref = torch.nan_to_num(scene.render(net, W, H)[0], nan=-7.0).clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.render(net, W, H)
    torch.cuda.synchronize()
    alone_ms = (time.perf_counter() - t0) * 1e3
    for kind in ([int(k) for k in os.environ["AGGR_KINDS"].split(",")] if os.environ.get("AGGR_KINDS") else range(agg.aggressor_kinds())):
        bad, nvals, lanes, ms = 0, 0, set(), 0.0
        for i in range(N):
            torch.cuda.synchronize()
            with torch.cuda.stream(sB):
                agg.aggressor(kind, BLOCKS, int(alone_ms * 4000) + 3000, ctypes.c_void_p(sB.cuda_stream))
            time.sleep(0.0005)
            t0 = time.perf_counter()
            img = torch.nan_to_num(scene.render(net, W, H)[0], nan=-7.0)
            torch.cuda.current_stream().synchronize()
            ms += (time.perf_counter() - t0) * 1e3
            if not torch.equal(ref, img):
                d = (ref != img).nonzero().cpu().numpy()
                bad += 1
                nvals += len(d)
                lanes |= set(int((y % 8) * 8 + (x % 8)) for _, y, x in d)
        print("%-8s next to %-28s %2d/%d differ%s   (render %.2f ms, alone %.2f)" % (
            act, agg.aggressor_name(kind).decode(), bad, N, (" (%d vals, lanes %d-%d)" % (nvals, min(lanes), max(lanes))) if bad else "", ms / N, alone_ms))
