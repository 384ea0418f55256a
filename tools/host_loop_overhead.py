"""Developer tool: host time and wall time per frame of bench.py's frame loop (Runner) for one rank of a 1/2/4/8-GPU run,
on ONE GPU with the collective stubbed out.  usage: tools/host_loop_overhead.py [config]"""
import importlib.util, os, sys, time, types
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import torch
from fvsrn_amd import synthetic as util
import torch.distributed as dist
from fvsrn_amd import capi, volnet_io
name = sys.argv[1] if len(sys.argv) > 1 else "c32l4_fourier_1024x512"
cfg = b.CONFIGS[name]
print(name)
vn, net = b.make_network(volnet_io, capi, cfg, "ReLU")
dist.all_gather_into_tensor = lambda out, inp: None   # host cost of the collective call itself is not in this number
for world in (1, 2, 4, 8):
    r = b.Runner(capi, net, cfg, 0, world, False)
    for i in range(8): r.frame(i)
    r.finish(); torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n): r.frame(8 + i, record=True)
    t_host = time.perf_counter() - t0
    r.finish(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("world %d: host loop %.1f us/frame, wall %.1f us/frame" % (world, 1e6 * t_host / n, 1e6 * t_all / n))
