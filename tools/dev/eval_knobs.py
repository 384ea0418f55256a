#!/usr/bin/env python3
"""Developer tool: evaluate_points of the 32x4 ReLU bench network over launch-shape options and sizes (interleaved, repeated: the clock of the
chip moves with the load history)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fvsrn_amd import capi, synthetic, volnet_io
vn = synthetic.random_network(output_mode="density:direct", seed=1234, C=32, layers=4, activation=sys.argv[1] if len(sys.argv) > 1 else "ReLU")
CONF = [dict(), dict(waves_per_block=4), dict(waves_per_block=2), dict(relu_clamp=0), dict(relu_clamp=0, waves_per_block=4), dict(relu_clamp=0, waves_per_block=2),
        dict(waves_per_block=4, max_blocks_per_cu=24), dict(relu_clamp=0, waves_per_block=4, max_blocks_per_cu=24), dict(relu_clamp=0, waves_per_block=4, max_blocks_per_cu=8)]
for logn in (20, 22, 24, 26):
    n = 1 << logn
    pos = torch.rand(n, 3, device="cuda")
    nets = []
    for opts in CONF:
        net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
        for k, v in opts.items():
            net.set_option(k, v)
        nets.append((opts, net, net.evaluate(pos)))
    best = {i: 1e9 for i in range(len(CONF))}
    for rep in range(4):
        for i, (opts, net, out) in enumerate(nets):
            for _ in range(3):
                net.evaluate(pos, out=out)
            torch.cuda.synchronize()
            reps = 20 if logn <= 24 else 6
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                net.evaluate(pos, out=out)
            e1.record()
            torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / reps)
    for i, (opts, net, out) in enumerate(nets):
        print("2^%d %-70s %.4f ms  %.1f G points/s" % (logn, opts, best[i], n / best[i] / 1e6))
    del pos, nets
