"""Box script: where the rotating cell-table kernel differs from the LDS kernel / the oracle in the scene of
test_register_resident_kernel_matches_lds_kernel[3-SnakeAlt-density-texture-grid4]."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import util  # noqa
import torch
from test_gpu_parity import make_scene_kwargs, oracle
from fvsrn_amd import capi, volnet_io

for early in (True, False):
    vn = util.random_network(C=32, layers=3, activation="SnakeAlt", output_mode="density", seed=77, box_min=(-0.5, -0.5, -0.5), grid=(16, 12), grid_scale=0.3)
    kw = make_scene_kwargs(stepsize=1 / 128, early_out=early, tf_scale_absorption=10.0, density_min=-1.0, density_max=1.0)
    rng = np.random.RandomState(5)
    tab = rng.uniform(0.0, 1.0, (64, 4)).astype(np.float32)
    tab[:, 3] *= 20.0
    kw.update(tf_kind=oracle.TF_TEXTURE, tf_table=tab)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    W = H = 96
    imgs = {}
    for name, opts in dict(cells_rot=dict(), cells_exact=dict(fourier_resync=1), gather=dict(cell_table=0), lds=dict(small_kernel=0)).items():
        sc = capi.Scene(**kw)
        for k, v in opts.items():
            sc.set_option(k, v)
        imgs[name] = torch.nan_to_num(sc.render(net, W, H)[0].clone(), nan=-7.0).cpu().numpy()
        print(name, sc.last_render_info())
    ref, _ = oracle.OracleScene(**kw).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT), W, H, 0, H)
    ref = np.nan_to_num(ref, nan=-7.0)
    print("early_out", early)
    for a in imgs:
        print("  %-12s vs oracle %.2e" % (a, np.abs(imgs[a][:4] - ref[:4]).max()), " ".join("vs %s %.2e" % (b, np.abs(imgs[a][:4] - imgs[b][:4]).max()) for b in imgs if b != a))
