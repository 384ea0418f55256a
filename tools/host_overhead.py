#!/usr/bin/env python3
"""Host-side cost of one bench frame (scene update + launch through the C ABI) measured on a frame so small that the GPU
is idle: the floor below which more GPUs cannot shorten a frame."""
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import torch  # noqa: E402
from fvsrn_amd import synthetic as util  # noqa: E402
from fvsrn_amd import capi, volnet_io  # noqa: E402

cfg = (32, 4, None, 64, 64, 64)
vn, net = b.make_network(volnet_io, capi, cfg, "ReLU")
r = b.Runner(capi, net, cfg, 0, 1, False)
for i in range(20):
    r.frame(i)
torch.cuda.synchronize()
n = 500
t0 = time.perf_counter()
for i in range(n):
    r.frame(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host time per frame: %.1f us (submit), %.1f us incl. final sync" % (1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
