#!/usr/bin/env python3
"""Developer stress (GPU box): the SAME frames rendered alone and two at a time on two streams (static network, no time change)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from fvsrn_amd import capi, tiles, volnet_io
from test_gpu_stripes import _scene_kw

def run(enc, has_time, rounds, segs=None, W=256, H=192, opts=None):
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="density", grid=(16, 8), seed=52, box_min=(-0.5, -0.5, -0.5),
                             fourier_std=0.4, encoding=enc, time_grids=5, has_time=has_time)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    net.set_time_and_ensemble(1.3, 0)
    yaws = [0.3 + 0.37 * i for i in range(8)]
    scene = capi.Scene(**_scene_kw(0.0))
    for k, v in (opts or {}).items():
        scene.set_option(k, v)
    refs = []
    for yaw in yaws:
        scene.update(**_scene_kw(yaw))
        refs.append(torch.nan_to_num(scene.render(net, W, H).clone(), nan=-7.0))
    torch.cuda.synchronize()
    pipe = tiles.StripeRenderer(net, W, H, _scene_kw(0.0), pipelined=True)
    for sc in pipe.scenes:
        for k, v in (opts or {}).items():
            sc.set_option(k, v)
    bad = {}
    where = None
    for r in range(rounds):
        for i in range(0, len(yaws), 2):
            for j in (i, i + 1):
                pipe.submit(j, _scene_kw(yaws[j]))
            pipe.finish(); torch.cuda.synchronize()
            for j in (i, i + 1):
                g = torch.nan_to_num(pipe.frame(j & 1), nan=-7.0)
                if not torch.equal(g, refs[j]):
                    d = (g - refs[j]).abs()
                    bad.setdefault(j, []).append((int((d > 0).sum()), float(d.max())))
                    if where is None:
                        ys, xs = torch.nonzero(d[0].amax(0) > 0, as_tuple=True)
                        where = sorted(set((int(y) // 8, int(x) // 8) for y, x in zip(ys, xs)))[:6], sorted(set((int(y) % 8, int(x) % 8) for y, x in zip(ys, xs)))[:20]
    print("enc", enc, "has_time", has_time, "opts", opts, scene.last_render_info(), net.kernel_name(True)[:44], "| differing frames:", {k: (len(v), v[0]) for k, v in bad.items()}, "| tiles, in-tile pixels:", where, flush=True)

if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    run(2, False, R)
    run(2, False, R, opts={"depth_segments": 1})
    run(2, False, R, opts={"small_kernel": 0})
    run(1, True, R)
    run(0, True, R)
    run(0, False, R)
    run(0, False, R, opts={"small_kernel": 0})
