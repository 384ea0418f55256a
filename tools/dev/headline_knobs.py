"""Box script: BASELINE the headline frame (32x4 Fourier-only, 1024^2 x 512 steps) under the launch-shape options -- which one does the automatic choice leave on the table?"""
import os, sys, time, itertools
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa
from fvsrn_amd import capi, volnet_io  # noqa
vn = bench.bench_network(32, 4, None, "ReLU")
net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
kw = bench.build_scene_kwargs(capi, 0.7, 1 / 512, False)
W = H = 1024
def run(opts, n=12):
    sc = capi.Scene(**kw)
    for k, v in opts.items():
        sc.set_option(k, v)
    for _ in range(5):
        sc.render(net, W, H)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            sc.render(net, W, H)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best, sc.last_render_info()
# spin the clock up
t_end = time.perf_counter() + 0.3
sc0 = capi.Scene(**kw)
while time.perf_counter() < t_end:
    sc0.render(net, W, H)
torch.cuda.synchronize()
base, plan = run({})
print("automatic: %.4f ms  %s" % (base, plan))
for seg in (1, 2, 4, 8):
    for wpb in (0, 1, 2, 4):
        for pers in (-1, 0):
            t, plan = run(dict(depth_segments=seg, waves_per_block=wpb, persistent=pers))
            print("segments %d waves_per_block %d persistent %2d: %.4f ms (%.2f of automatic)" % (seg, wpb, pers, t, t / base))
for order in (0, 1):
    t, _ = run(dict(tile_order=order))
    print("tile_order %d: %.4f ms" % (order, t))
for res in (32, 128, 256):
    t, _ = run(dict(fourier_resync=res))
    print("fourier_resync %d: %.4f ms" % (res, t))
