#!/usr/bin/env python3
"""How long does ONE rank's share of a multi-GPU frame take on its GPU, against frame_time / world?  (The communication is
not part of this: one GPU, no collective.)  usage: tools/stripe_efficiency.py [config]"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import torch  # noqa: E402
from fvsrn_amd import synthetic as util  # noqa: E402
from fvsrn_amd import capi, volnet_io  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c32l4_fourier_1024x512"
cfg = b.CONFIGS[name]
vn, net = b.make_network(volnet_io, capi, cfg, "ReLU")
_, _, _, W, H, steps = cfg
scene = capi.Scene(**b.build_scene_kwargs(capi, 0.3, 1.0 / steps, False))


def timed(fn, reps=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


full = torch.zeros((1, 8, H, W), dtype=torch.float32, device="cuda")
t_full = timed(lambda: scene.render(net, W, H, out=full))
print("%s: full frame %.3f ms" % (name, t_full))
for world in (2, 4, 8):
    worst = 0.0
    for rank in range(world):
        rows = capi.stripe_rows(H, b.STRIPE, rank, world)
        out = torch.zeros((8, rows, W), dtype=torch.float32, device="cuda")
        worst = max(worst, timed(lambda: capi.render_stripes(scene, net, W, H, b.STRIPE, rank, world, out=out)))
    print("  world %d: slowest rank %.3f ms, ideal %.3f ms -> render-only efficiency %.0f%%" % (world, worst, t_full / world, 100 * t_full / world / worst))
