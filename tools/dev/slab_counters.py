#!/usr/bin/env python3
"""Developer tool (GPU box): how the wave steps of a latent-grid frame go through the slab path (srn_device.hpp) -- needs a library built with
-DFVSRN_SLAB_COUNTERS (tools/variant.sh slabcount "-DFVSRN_SLAB_COUNTERS=1" kernels_small_cells; FVSRN_LIBRARY=fv-srn_amd/ablate/libfvsrn_slabcount.so).
Prints, per workload: wave steps, and the share that ran on slot 1 alone / needed slot 2 / picked slot 1 / promoted slot 2 / picked slot 2 / further slabs."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import torch  # noqa: E402
from fvsrn_amd import capi, volnet_io  # noqa: E402

for name in sys.argv[1:] or ["c32l4_grid16_1024x512", "c32l4_grid16r32_1024x512"]:
    cfg = b.CONFIGS[name]
    _, net = b.make_network(volnet_io, capi, cfg, "ReLU", 1)
    sc = capi.Scene(**b.build_scene_kwargs(capi, 0.3, 1.0 / cfg[5], False))
    stats = torch.zeros(16, dtype=torch.int64, device="cuda")
    sc.render(net, cfg[3], cfg[4], stats=stats)
    torch.cuda.synchronize()
    st = stats.cpu().tolist()
    steps = st[1] / 64
    names = ["slot1_only", "two_slots", "pick_slot1", "promote", "pick_slot2", "further_slabs"]
    print(json.dumps({"workload": name, "kernel": sc.last_kernel_name(), "wave_steps": steps, **{n: st[2 + i] / steps for i, n in enumerate(names)}}))
