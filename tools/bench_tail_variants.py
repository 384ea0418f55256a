import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fvsrn_amd
from fvsrn_amd import synthetic as util
from fvsrn_amd import capi, volnet_io
def run(mode, act, tf_kind, tf_table=None):
    vn = util.random_network(C=32, layers=4, activation=act, param=1.0, output_mode=mode, seed=1234, box_min=(-0.5, -0.5, -0.5))
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.3, 1.6)
    kw = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1/512, early_out=False, tf_kind=tf_kind, tf_scale_absorption=10.0, tf_scale_emission=1.0)
    if tf_table is not None: kw["tf_table"] = tf_table
    scene = capi.Scene(**kw)
    out = torch.zeros((1, 8, 1024, 1024), dtype=torch.float32, device="cuda"); stats = torch.zeros(2, dtype=torch.int64, device="cuda")
    for _ in range(4): scene.render(net, 1024, 1024, out=out)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(16): scene.render(net, 1024, 1024, out=out, stats=stats)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 16
    print("%-16s %-9s tf %d: %.3f ms  %.1f Gsamples/s  %s" % (mode, act, tf_kind, ms, stats.cpu()[0].item() / 16 / ms / 1e6, net.kernel_name(True)[:40]))
rng = np.random.RandomState(0)
gauss = np.array([[0.9, 0.2, 0.1, 30.0, 0.3, 0.1], [0.1, 0.8, 0.3, 60.0, 0.7, 0.08]], np.float32)
pw = np.array([[0, 0, 0, 0, -1.0], [0.2, 0.4, 0.9, 5.0, 0.3], [0.9, 0.5, 0.1, 40.0, 0.7], [1, 1, 1, 10.0, 2.0]], np.float32)
tex = rng.uniform(0, 1, (256, 4)).astype(np.float32); tex[:, 3] *= 30
run("density:direct", "ReLU", capi.TF_IDENTITY)
run("density:direct", "ReLU", capi.TF_TEXTURE, tex)
run("density", "ReLU", capi.TF_TEXTURE, tex)
run("density:direct", "ReLU", capi.TF_PIECEWISE, pw)
run("density:direct", "ReLU", capi.TF_GAUSSIAN, gauss)
run("rgbo:direct", "ReLU", capi.TF_NONE)
run("rgbo", "ReLU", capi.TF_NONE)
run("rgbo", "SnakeAlt", capi.TF_NONE)
run("density", "SnakeAlt", capi.TF_TEXTURE, tex)
