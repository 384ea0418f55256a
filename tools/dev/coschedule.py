#!/usr/bin/env python3
"""Developer tool: does a small kernel on another stream get onto the chip while a render launch holds it?  One rank's share of
configs[3] at world 8 (1.7 ms) on stream A; 300 us later a stand-in for a collective (tools/dev/occupy.hip: 24 workgroups x 512 threads
spinning 150 us) on stream B, timed with events on B.  usage: tools/dev/coschedule.py   (FVSRN_PERSISTENT / FVSRN_PERSISTENT_RESERVE / ...)"""
import ctypes
import importlib.util
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import torch  # noqa: E402
from fvsrn_amd import capi, tiles, volnet_io  # noqa: E402

occ = ctypes.CDLL(os.path.join(ROOT, "tools", "dev", "bin", "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
name = "c64l6_grid16_1024x512"
cfg = b.CONFIGS[name]
_, net = b.make_network(volnet_io, capi, cfg, "ReLU", 1)
_, _, _, W, H, steps = cfg
kw = b.build_scene_kwargs(capi, 0.3, 1.0 / steps, False)
pipe = tiles.StripeRenderer(net, W, H, kw, rank=3, world=8, stripe=b.STRIPE, pipelined=False)
sB = torch.cuda.Stream()
for blocks, threads, us in ((24, 512, 150), (24, 256, 150), (8, 512, 150)):
    rows = []
    for rep in range(6):
        pipe.submit(rep, kw, gather=False)
        pipe.finish()
        torch.cuda.synchronize()
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        pipe.submit(rep, kw, gather=False)
        pipe.finish()
        r1.record()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 300e-6:
            pass
        with torch.cuda.stream(sB):
            g0.record()
            occ.occupy(blocks, threads, us, ctypes.c_void_p(sB.cuda_stream))
            g1.record()
        torch.cuda.synchronize()
        rows.append((r0.elapsed_time(r1), g0.elapsed_time(g1), r0.elapsed_time(g1)))
    rows = rows[1:]
    print("stand-in %2d x %3d threads x %d us: render %.3f ms, stand-in from its submission to its end %.3f ms, render start to stand-in end %.3f ms"
          % (blocks, threads, us, sum(r[0] for r in rows) / len(rows), sum(r[1] for r in rows) / len(rows), sum(r[2] for r in rows) / len(rows)))
