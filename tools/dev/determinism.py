#!/usr/bin/env python3
"""Developer tool: is a render bit-identical from launch to launch?  Renders one scene N times per configuration and reports every
launch whose image differs from the first one: how many values, which planes, which pixels (tile and lane inside the 8x8 tile).
usage: tools/dev/determinism.py [repeats]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fvsrn_amd import capi, synthetic, volnet_io  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
POISON = None
if len(sys.argv) > 2:  # a bit pattern (hex) written into every vector register of the chip before each launch
    import ctypes
    _occ = ctypes.CDLL(os.path.join(ROOT, "tools", "dev", "bin", "liboccupy.so"))
    _occ.poison.argtypes = [ctypes.c_uint, ctypes.c_void_p]
    POISON = int(sys.argv[2], 16)
eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
kw = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1 / 128, early_out=False,
          tf_kind=capi.TF_IDENTITY, tf_scale_absorption=20.0, tf_scale_emission=1.0)
W, H = 1024, 512
G = dict(grid=(16, 8))
CASES = [("64x3 ReLU + grid, stripes (render_kernel, fragment-major order)", dict(C=64, layers=3, activation="ReLU", **G), "stripes", {}),
         ("64x3 ReLU + grid, stripes, render_kernel", dict(C=64, layers=3, activation="ReLU", **G), "stripes", dict(overlap_kernel=0)),
         ("64x3 ReLU + grid, frame, render_kernel", dict(C=64, layers=3, activation="ReLU", **G), "frame", {}),
         ("64x3 ReLU + grid, frame, render_kernel, fragment-major order", dict(C=64, layers=3, activation="ReLU", **G), "frame", dict(overlap_kernel=1)),
         ("64x3 ReLU + grid, frame, render_kernel, fragment-major order, not persistent", dict(C=64, layers=3, activation="ReLU", **G), "frame", dict(overlap_kernel=1, persistent=0)),
         ("64x3 SnakeAlt + grid, stripes (render_kernel, fragment-major order)", dict(C=64, layers=3, activation="SnakeAlt", **G), "stripes", {}),
         ("64x3 SnakeAlt + grid, stripes, render_kernel", dict(C=64, layers=3, activation="SnakeAlt", **G), "stripes", dict(overlap_kernel=0)),
         ("64x3 SnakeAlt + grid, stripes, not persistent", dict(C=64, layers=3, activation="SnakeAlt", **G), "stripes", dict(persistent=0, unit_quota=0)),
         ("64x3 SnakeAlt, no grid, stripes", dict(C=64, layers=3, activation="SnakeAlt"), "stripes", {}),
         ("64x3 ReLU, no grid, frame", dict(C=64, layers=3, activation="ReLU"), "frame", {}),
         ("48x3 SnakeAlt + grid, stripes", dict(C=48, layers=3, activation="SnakeAlt", **G), "stripes", {}),
         ("32x4 ReLU + grid, frame (resident kernel)", dict(C=32, layers=4, activation="ReLU", **G), "frame", {}),
         ("32x4 SnakeAlt + grid, frame (resident kernel)", dict(C=32, layers=4, activation="SnakeAlt", **G), "frame", {}),
         ("32x4 SnakeAlt + grid, frame, LDS kernel", dict(C=32, layers=4, activation="SnakeAlt", **G), "frame", dict(small_kernel=0)),
         ("32x4 SnakeAlt + grid, stripes, LDS kernel, not persistent", dict(C=32, layers=4, activation="SnakeAlt", **G), "stripes", dict(small_kernel=0, persistent=0, unit_quota=0)),
         ("32x4 ReLU + grid, shaded, adjoint gradients", dict(C=32, layers=4, activation="ReLU", **G), "adjoint", {}),
         ("64x3 SnakeAlt + grid, shaded, adjoint gradients", dict(C=64, layers=3, activation="SnakeAlt", **G), "adjoint", {}),
         ("64x3 ReLU + grid, shaded, finite differences", dict(C=64, layers=3, activation="ReLU", **G), "fd", {}),
         ("32x4 SnakeAlt + grid, evaluate_points_adjoint", dict(C=32, layers=4, activation="SnakeAlt", **G), "eval_adjoint", {}),
         ("64x3 ReLU + grid, evaluate_points", dict(C=64, layers=3, activation="ReLU", **G), "eval", {}),
         ("32x4 ReLU Fourier-only, Gaussian TF, early-out", dict(C=32, layers=4, activation="ReLU"), "gauss", {}),
         ("32x4 SnakeAlt + grid, Gaussian TF, early-out", dict(C=32, layers=4, activation="SnakeAlt", **G), "gauss", {}),
         ("64x3 ReLU + grid, piecewise TF", dict(C=64, layers=3, activation="ReLU", **G), "piecewise", {}),
         ("32x4 SnakeAlt rgbo Fourier-only", dict(C=32, layers=4, activation="SnakeAlt", output_mode="rgbo"), "frame", {}),
         ("128x3 ReLU + grid", dict(C=128, layers=3, activation="ReLU", **G), "frame", {}),
         ("96x3 SnakeAlt + grid", dict(C=96, layers=3, activation="SnakeAlt", **G), "frame", {}),
         # r04: the cases above take the cell table (a 1024 x 512 image of an 8^3 grid: the automatic rule picks it); the gather kernels of the same frames (grid_tap: where hipcc emitted the
         # packed-fp32 selection of profiles/r04/nondeterminism_r04.md)
         ("64x3 ReLU + grid, frame, gather path", dict(C=64, layers=3, activation="ReLU", **G), "frame", dict(cell_table=0)),
         ("64x3 SnakeAlt + grid, stripes, gather path", dict(C=64, layers=3, activation="SnakeAlt", **G), "stripes", dict(overlap_kernel=0, cell_table=0)),
         ("32x4 ReLU + grid, frame, resident kernel, gather path", dict(C=32, layers=4, activation="ReLU", **G), "frame", dict(cell_table=0)),
         ("32x4 SnakeAlt + grid, frame, LDS kernel, gather path", dict(C=32, layers=4, activation="SnakeAlt", **G), "frame", dict(small_kernel=0, cell_table=0)),
         ("96x3 SnakeAlt + grid, gather path", dict(C=96, layers=3, activation="SnakeAlt", **G), "frame", dict(cell_table=0)),
         ("128x3 ReLU + grid, gather path", dict(C=128, layers=3, activation="ReLU", **G), "frame", dict(cell_table=0)),
         ("64x3 ReLU + grid, shaded, finite differences, gather path", dict(C=64, layers=3, activation="ReLU", **G), "fd", dict(cell_table=0)),
         ("32x4 SnakeAlt + grid, shaded, finite differences", dict(C=32, layers=4, activation="SnakeAlt", **G), "fd", {})]
GAUSS = np.array([[0.9, 0.2, 0.1, 30.0, 0.25, 0.08], [0.1, 0.7, 0.9, 20.0, 0.6, 0.1], [0.9, 0.9, 0.2, 40.0, 0.85, 0.05]], np.float32)
PIECE = np.array([[0.0, 0, 0, 0, 0], [0.2, 1, 0, 0, 10], [0.5, 0, 1, 0, 30], [0.8, 0, 0, 1, 5], [1.0, 1, 1, 1, 40]], np.float32)
PHONG = dict(enable_phong=True, ambient=0.2, specular=0.4, magnitude_center=0.6, magnitude_radius=0.5, specular_exponent=8, light_type=0, light=tuple(float(v) for v in eye))
POINTS = torch.from_numpy(np.random.RandomState(3).uniform(-0.5, 0.5, (1 << 18, 3)).astype(np.float32)).cuda()
# the dense grid-volume renderer (BASELINE configs[0] on the GPU): trilinear fetches of its own
_ax = np.linspace(-1, 1, 96, dtype=np.float32)
_x, _y, _z = np.meshgrid(_ax, _ax, _ax, indexing="ij")
_vol = capi.Volume.from_array(np.clip(np.exp(-3 * (_x * _x + _y * _y + _z * _z)) + 0.1 * np.sin(9 * _x) * np.cos(7 * _y) * np.sin(5 * _z), 0, 1).astype(np.float32),
                              (-0.5, -0.5, -0.5), (1, 1, 1))
_vscene = capi.Scene(**dict(kw, stepsize=1 / 96, tf_scale_absorption=10.0))
_first, _bad = None, 0
for _i in range(N):
    _img = torch.nan_to_num(_vol.render(_vscene, 512, 512, 1), nan=-7.0).clone()
    if _first is None:
        _first = _img
    elif not torch.equal(_first, _img):
        _bad += 1
print("%-56s %d of %d launches differ from the first" % ("dense 96^3 grid volume, trilinear, frame", _bad, N - 1))
for name, net_kw, what, opts in CASES:
    net_kw = dict(net_kw)
    vn = synthetic.random_network(output_mode=net_kw.pop("output_mode", "density"), seed=62, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4, grid_scale=0.3, **net_kw)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    skw = dict(kw)
    if what in ("adjoint", "fd"):
        skw.update(gradient_mode=2 if what == "adjoint" else 1, finite_differences_stepsize=1 / 256, brdf=PHONG, stepsize=1 / 64)
    if what == "gauss":
        skw.update(tf_kind=capi.TF_GAUSSIAN, tf_table=GAUSS, early_out=True)
    if what == "piecewise":
        skw.update(tf_kind=capi.TF_PIECEWISE, tf_table=PIECE)
    if name.find("rgbo") >= 0:
        skw.update(tf_kind=capi.TF_NONE)
    scene = capi.Scene(**skw).set_option("depth_segments", 1)
    for k, v in opts.items():
        scene.set_option(k, v)
    first, bad = None, []
    for i in range(N):
        if POISON is not None and i > 0:
            _occ.poison(POISON, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if what == "eval_adjoint":
            v, g = net.evaluate_with_adjoint_gradient(POINTS, None, world=True)
            img = torch.cat([v, g], dim=1).t().reshape(4, 512, 512)
        elif what == "eval":
            img = net.evaluate(POINTS, None, world=True).t().reshape(1, 512, 512)
        else:
            img = capi.render_stripes(scene, net, W, H, 16, 1, 2) if what == "stripes" else scene.render(net, W, H)[0]
        img = torch.nan_to_num(img, nan=-7.0).clone()
        if first is None:
            first = img
        elif not torch.equal(first, img):
            d = (first != img).nonzero().cpu().numpy()
            planes = sorted(set(d[:, 0].tolist()))
            px = sorted(set((int(y), int(x)) for _, y, x in d))
            lanes = sorted(set((y % 8) * 8 + (x % 8) for y, x in px))
            tiles = sorted(set((y // 8, x // 8) for y, x in px))
            bad.append((i, len(d), float((first - img).abs().max()), planes, tiles[:4], lanes))
    print("%-56s %d of %d launches differ from the first" % (name, len(bad), N - 1))
    for b in bad[:2]:
        print("     launch %d: %d values, max |diff| %.2e, planes %s, tiles %s, lanes %s" % b)
