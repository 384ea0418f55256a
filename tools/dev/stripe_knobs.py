"""Box script: one rank's stripes of the headline frame at world 8 (1024^2 x 512, 16-row stripes) under the launch-shape options."""
import os, sys, time
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
def run(opts, world=8, rank=3, n=60):
    sc = capi.Scene(**kw)
    for k, v in opts.items():
        sc.set_option(k, v)
    for _ in range(8):
        capi.render_stripes(sc, net, W, H, 16, rank, world)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            capi.render_stripes(sc, net, W, H, 16, rank, world)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best, sc.last_render_info()
t_end = time.perf_counter() + 0.3
sc0 = capi.Scene(**kw)
while time.perf_counter() < t_end:
    sc0.render(net, W, H)
torch.cuda.synchronize()
full, _ = run({}, world=1, rank=0, n=20)
print("whole frame %.4f ms -> ideal share at world 8: %.4f ms" % (full, full / 8))
base, plan = run({})
print("automatic: %.4f ms (%.0f %%)  %s" % (base, 100 * full / 8 / base, plan))
for pers in (0, 1):
    for seg in (0, 1, 2, 4, 8):
        for wpb in (0, 1, 2, 4):
            t, plan = run(dict(depth_segments=seg, waves_per_block=wpb, persistent=pers))
            print("persistent %d segments %d waves_per_block %d: %.4f ms (%.0f %%) K=%d" % (pers, seg, wpb, t, 100 * full / 8 / t, plan["segments"]))
for q in (0, 1, 2, 4):
    t, _ = run(dict(persistent=0, unit_quota=q))
    print("bounded, unit_quota %d: %.4f ms (%.0f %%)" % (q, t, 100 * full / 8 / t))
for r in (0, 16, 64):
    t, _ = run(dict(persistent=1, persistent_reserve=r))
    print("persistent, reserve %d: %.4f ms (%.0f %%)" % (r, t, 100 * full / 8 / t))
