import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa
from fvsrn_amd import capi, volnet_io, synthetic  # noqa
kw = bench.build_scene_kwargs(capi, 0.7, 1 / 512, False)
for C, layers, res, seed, gs in ((64, 4, 64, 7, 0.05), (64, 4, 64, 1234, 0.01), (64, 6, 64, 7, 0.05), (64, 4, 32, 7, 0.05)):
    vn = synthetic.random_network(C=C, layers=layers, activation="ReLU", param=1.0, output_mode="density:direct", grid=(16, res), seed=seed, box_min=(-0.5, -0.5, -0.5), grid_scale=gs)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    out = []
    for name, opt in (("cells", -1), ("gather", 0), ("cells", -1)):
        sc = capi.Scene(**kw).set_option("cell_table", opt)
        ts = []
        for _ in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            img = sc.render(net, 512, 512)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        out.append("%s %s alpha %.3f plan %s" % (name, ["%.2f" % t for t in ts], float(torch.nan_to_num(img[0, 3]).max()), sc.last_render_info()["segments"]))
    print(C, layers, res, seed, gs, net.kernel_name(True)[:40], " | ".join(out), flush=True)
