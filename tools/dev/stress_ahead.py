#!/usr/bin/env python3
"""Developer stress test (GPU box): determinism of serial renders, and the pipelined renderer against them, many rounds."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from fvsrn_amd import capi, tiles, volnet_io
from test_gpu_stripes import _scene_kw

def run(enc, has_time, slots, ahead, rounds, W=256, H=192):
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="density", grid=(16, 8), seed=52, box_min=(-0.5, -0.5, -0.5),
                             fourier_std=0.4, encoding=enc, time_grids=5, has_time=has_time)
    blob = volnet_io.save_volnet(vn)
    times = [0.0, 0.75, 1.5, 3.9, 2.25, 2.25, 4.0, 0.1, 0.1, 3.3]
    yaws = [0.3 + 0.37 * i for i in range(len(times))]
    bad_serial, bad_pipe = {}, {}
    refs = None
    for r in range(rounds):
        net, serial = capi.Network.from_volnet(blob), capi.Network.from_volnet(blob)
        if slots:
            net.set_option("keyframe_slots", slots)
        pipe = tiles.StripeRenderer(net, W, H, _scene_kw(0.0), pipelined=True)
        got = {}
        for i in range(0, len(times), 2):
            for j in (i, i + 1):
                pipe.submit(j, _scene_kw(yaws[j]), time=times[j], next_time=times[j + 1] if ahead and j + 1 < len(times) else None)
            pipe.finish(); torch.cuda.synchronize()
            for j in (i, i + 1):
                got[j] = torch.nan_to_num(pipe.frame(j & 1).clone(), nan=-7.0)
        ref_scene = capi.Scene(**_scene_kw(0.0))
        cur = []
        for t, yaw in zip(times, yaws):
            serial.set_time_and_ensemble(t, 0); ref_scene.update(**_scene_kw(yaw))
            cur.append(torch.nan_to_num(ref_scene.render(serial, W, H).clone(), nan=-7.0))
        torch.cuda.synchronize()
        if refs is None:
            refs = cur
        for j in range(len(times)):
            if not torch.equal(cur[j], refs[j]):
                d = (cur[j] - refs[j]).abs(); bad_serial.setdefault(j, []).append((r, int((d > 0).sum()), float(d.max())))
            if not torch.equal(got[j], refs[j]):
                d = (got[j] - refs[j]).abs(); bad_pipe.setdefault(j, []).append((r, int((d > 0).sum()), float(d.max())))
    print("enc", enc, "has_time", has_time, "slots", slots, "ahead", ahead, "rounds", rounds, "| serial vs first serial:", {k: (len(v), v[:1]) for k, v in bad_serial.items()},
          "| pipeline vs first serial:", {k: (len(v), v[:1]) for k, v in bad_pipe.items()}, "|", net.kernel_name(True)[:40], flush=True)

if __name__ == "__main__":
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for enc, ht, slots in [(0, False, 0), (0, True, 0), (2, False, 0), (0, False, 2), (1, True, 3)]:
        for ah in (False, True):
            run(enc, ht, slots, ah, R)
