#!/usr/bin/env python3
"""Developer tool: cycle split of the render loop's wave step from a -DFVSRN_PROF_SECTIONS build
(tools/ablate.sh "-DFVSRN_PROF_SECTIONS", then FVSRN_LIBRARY=fv-srn_amd/ablate/libfvsrn__DFVSRN_PROF_SECTIONS.so).
usage: tools/section_profile.py [config]     (FVSRN_MAX_BLOCKS_PER_CU=4 -> one wave per SIMD: pure latencies)
The marks sit in render_kernel's pipelined layer loop: run with FVSRN_SMALL_KERNEL=0 for networks the register-resident kernel would take."""
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
out = torch.zeros((1, 8, H, W), dtype=torch.float32, device="cuda")
stats = torch.zeros(16, dtype=torch.int64, device="cuda")
for _ in range(3):
    scene.render(net, W, H, out=out, stats=stats)
torch.cuda.synchronize()
stats.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
scene.render(net, W, H, out=out, stats=stats)
e1.record()
torch.cuda.synchronize()
st = stats.cpu().numpy()
wave_steps = st[1] / 64
labels = ["tail (output, TF, blend)", "loop head + LDS reads issued", "pre (features -> B, rotation)", "first layer", "hidden layers",
          "last layer (+ rotation)"]
print("%s: %.3f ms, %.2f M wave steps" % (name, e0.elapsed_time(e1), wave_steps / 1e6))
tot = 0.0
for k, l in enumerate(labels):
    c = st[2 + k] / wave_steps
    tot += c
    print("  %-34s %8.1f cycles / wave step" % (l, c))
print("  %-34s %8.1f" % ("sum (s_memtime counts at 100 MHz x ? -- compare ratios)", tot))
