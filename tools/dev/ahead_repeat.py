#!/usr/bin/env python3
"""Developer measurement (GPU box): frame period of one rank's pipeline for a time-dependent network, blend inside the render call vs
blend ahead on a side stream, several repetitions each (the two-stream pipeline has run-to-run spread)."""
import importlib.util, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
spec = importlib.util.spec_from_file_location("se", os.path.join(ROOT, "tools", "stripe_efficiency.py")); se = importlib.util.module_from_spec(spec); spec.loader.exec_module(se)
from fvsrn_amd import capi, volnet_io
b = se.b
name = "c64l6_grid16_time16_1024x512"
cfg, keys = b.CONFIGS[name], b.TIME_KEYS[name]
_, net = b.make_network(volnet_io, capi, cfg, "ReLU", keys)
for world in (2, 4, 8):
    for ahead in (False, True):
        vals = [round(se.frame_period(net, cfg, keys, r % world, world, frames=32, ahead=ahead), 3) for r in range(6)]
        print("world", world, "ahead", ahead, "ranks 0..5 (mod world):", vals, flush=True)
