"""Box script: a 64^3 x 16 latent grid behind 64x4 (cell table 250 MB + the plain image's) and a 128^3 grid (table above the 1 GiB cap: gather path)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa
from fvsrn_amd import capi, volnet_io, synthetic  # noqa
for C, layers, res in ((64, 4, 64), (32, 4, 64), (64, 4, 128)):
    vn = synthetic.random_network(C=C, layers=layers, activation="ReLU", param=1.0, output_mode="density:direct", grid=(16, res), seed=7, box_min=(-0.5, -0.5, -0.5), grid_scale=0.05)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    kw = bench.build_scene_kwargs(capi, 0.7, 1 / 512, False)
    W = H = 512
    res_ = {}
    for name, opt in (("cells", -1), ("gather", 0)):
        sc = capi.Scene(**kw).set_option("cell_table", opt)
        img = sc.render(net, W, H)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            sc.render(net, W, H)
        torch.cuda.synchronize()
        res_[name] = (torch.nan_to_num(img, nan=0.0).clone(), (time.perf_counter() - t0) / 5 * 1e3, sc.last_render_info()["cell_table"])
    d = float((res_["cells"][0][:4] - res_["gather"][0][:4]).abs().max())
    print("%dx%d + %d^3 grid: cells %.3f ms (table used: %s) gather %.3f ms, max |diff| %.2e, alpha max %.3f, mem %.0f MB" % (
        C, layers, res, res_["cells"][1], res_["cells"][2], res_["gather"][1], d, float(res_["cells"][0][3].max()), torch.cuda.mem_get_info()[1] / 1e6 - torch.cuda.mem_get_info()[0] / 1e6))
    del net
