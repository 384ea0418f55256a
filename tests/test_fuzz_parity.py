"""Randomised GPU <-> oracle parity: networks, scenes and image sizes drawn from a seeded generator, each rendered through the
C ABI and through the C restatement.  Complements the hand-picked cases of test_gpu_parity.py (every kernel family is reached:
register-resident, LDS, latent grid, generic / scalar / colour tails, depth segments)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import util  # noqa: E402
from oracle import oracle  # noqa: E402

pytestmark = pytest.mark.gpu
TOL_IMG = 3e-3  # the image tolerance of test_gpu_parity.py


def draw_case(seed):
    rng = np.random.RandomState(1000 + seed)
    C = int(rng.choice([32, 32, 48, 64]))
    layers = int(rng.randint(2, 6))
    act = str(rng.choice(["ReLU", "SnakeAlt", "Sine", "Snake", "Sigmoid"]))
    out = str(rng.choice(["density", "density:direct", "rgbo", "rgbo:direct", "densitygrad"]))
    grid = (16, int(rng.choice([4, 8]))) if rng.rand() < 0.35 else None
    # NeRF frequency ladder (2 pi 2^k, network.py:55-63) or a smooth random Fourier matrix, for every activation.  Random weights
    # behind the ladder and a periodic activation are sensitive to the position at the 1e-4 level: there the reference's own
    # fp16 arithmetic (phases in half, renderer_volume_tensorcores.cuh:797-806) moves colours by percents against an fp32
    # evaluation of the same network, see the tolerance in test_random_scene_matches_oracle.
    net = dict(C=C, layers=layers, activation=act, param=1.0, output_mode=out, grid=grid, seed=int(rng.randint(1 << 20)),
               box_min=(-0.5, -0.5, -0.5), fourier_std=0.4 if rng.rand() < 0.5 else None)
    eye, right, up = oracle.camera_on_a_sphere(str(rng.choice(["Ym", "Zp", "Xm"])), (0, 0, 0), float(rng.uniform(-0.6, 0.6)),
                                               float(rng.uniform(0, 6.28)), float(rng.uniform(1.2, 2.2)))
    scene = dict(eye=eye, right=right, up=up, fov_y_radians=float(rng.uniform(0.5, 1.0)), stepsize=float(1.0 / rng.choice([24, 48, 160])),
                 early_out=bool(rng.rand() < 0.5), blend_mode=int(rng.choice([oracle.BLEND_ALPHA, oracle.BLEND_BEER_LAMBERT])))
    if out.startswith("rgbo"):
        scene.update(tf_kind=oracle.TF_NONE)
    else:
        kind = int(rng.choice([oracle.TF_IDENTITY, oracle.TF_TEXTURE, oracle.TF_PIECEWISE, oracle.TF_GAUSSIAN]))
        scene.update(tf_kind=kind, density_min=-0.5 if "direct" in out else 0.1, density_max=0.5 if "direct" in out else 0.9)
        if kind == oracle.TF_IDENTITY:
            scene.update(tf_scale_absorption=float(rng.uniform(5, 40)), tf_scale_emission=float(rng.uniform(0.5, 1.5)))
        elif kind == oracle.TF_TEXTURE:
            # a smooth table (6 random control values per channel): i.i.d. texels would have slope R per unit density
            R = int(rng.choice([16, 64, 256]))
            ctrl = rng.uniform(0, 1, (6, 4))
            t = np.stack([np.interp(np.linspace(0, 5, R), np.arange(6), ctrl[:, c]) for c in range(4)], axis=1).astype(np.float32)
            t[:, 3] *= 30
            scene.update(tf_table=t)
        elif kind == oracle.TF_PIECEWISE:
            n = int(rng.randint(2, 6))
            pos = np.concatenate([[-1.0], np.sort((np.arange(n) + rng.uniform(0.2, 0.8, n)) / n), [2.0]])  # >= 0.4 / n apart
            t = np.concatenate([rng.uniform(0, 1, (n + 2, 3)), rng.uniform(0, 40, (n + 2, 1)), pos[:, None]], axis=1).astype(np.float32)
            scene.update(tf_table=t)
        else:
            n = int(rng.randint(1, 4))
            t = np.stack([rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.uniform(0, 1, n), rng.uniform(10, 60, n), rng.uniform(0.1, 0.9, n),
                          rng.uniform(0.04, 0.2, n)], axis=1).astype(np.float32)
            scene.update(tf_table=t)
    W, H = int(rng.choice([24, 33, 40, 57])), int(rng.choice([16, 21, 32]))
    return net, scene, W, H


def compare_case(seed, scene_options=None):
    """-> (max |rgba+normal| difference GPU vs oracle fp32-accumulate model, the same between the oracle's fp16-accumulate
    (reference CUDA arithmetic) and fp32-accumulate models, info string, img, ref, stats, count)"""
    import torch
    from fvsrn_amd import capi, volnet_io
    net_kw, scene_kw, W, H = draw_case(seed)
    vn = util.random_network(**net_kw)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    stats = torch.zeros(2, dtype=torch.int64, device="cuda")
    scene = capi.Scene(**scene_kw)
    for k, v in (scene_options or {}).items():
        scene.set_option(k, v)
    img = scene.render(net, W, H, stats=stats)[0].cpu().numpy()
    ref, count = oracle.OracleScene(**scene_kw).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT), W, H)
    ref_h, _ = oracle.OracleScene(**scene_kw).render(oracle.OracleNetwork(vn, oracle.ACC_HALF), W, H)
    info = "%s | %s | %s" % ({k: v for k, v in net_kw.items() if k != "box_min"},
                             {k: v for k, v in scene_kw.items() if k not in ("eye", "right", "up", "tf_table")}, net.kernel_name(True)[:32])
    return float(np.abs(img[:7] - ref[:7]).max()), float(np.abs(ref_h[:7] - ref[:7]).max()), info, img, ref, stats.cpu().numpy(), count, scene_kw


@pytest.mark.parametrize("seed", range(32))
def test_random_scene_matches_oracle(seed):
    """Tolerance: the image tolerance of test_gpu_parity.py (3e-3), or -- for networks on which the reference's own two arithmetic
    models disagree by more than that -- the distance between the oracle's fp16-accumulate model (the reference's CUDA arithmetic:
    phases and sums in half) and its fp32-accumulate model on this very image: the HIP path is never farther from the fp32-accumulate
    restatement than the reference's own renderer is.  (The one modelled difference: between two exact re-derivations the kernels'
    rotated Fourier features follow the un-rounded ray position, the reference rounds every position to fp16 -- scene option
    fourier_resync = 1 removes it, see test_random_scene_with_per_step_features.)"""
    err, spread, info, img, ref, stats, count, scene_kw = compare_case(seed)
    solid = ref[3] > 1e-4
    assert err < max(TOL_IMG, spread), "%s: |gpu - oracle| %.2e, oracle fp16 vs fp32 model %.2e" % (info, err, spread)
    if err < TOL_IMG:
        assert np.array_equal(np.isnan(img[7])[solid], np.isnan(ref[7])[solid]), info
        if solid.any():
            assert np.nanmax(np.abs(img[7] - ref[7])[solid]) < 3e-2, info
    if not scene_kw["early_out"]:
        assert int(stats[0]) == count, info  # lane-exact evaluated samples (early-out may stop a step apart at the threshold)


# Seeds of a 400-scene run of this fuzz (profiles/r02/fuzz_report_gpu_vs_oracle.txt) that miss the criterion above: 6 of 400, all 32-wide
# NeRF-ladder networks whose features the kernels advance by rotation.
ROTATION_OUTLIERS = [88, 92, 114, 136, 226, 288]


@pytest.mark.parametrize("seed", ROTATION_OUTLIERS)
def test_rotation_outliers_are_bounded_and_vanish_with_per_step_features(seed):
    """What the rotation costs in parity, at its worst: the reference rounds every sample position to fp16 before the Fourier stage, a
    phase error of up to 0.4 rad in the top octave of a 10-octave ladder -- noise that a rotated feature (exact increments from the last
    re-derived position) does not follow.  On 6 of 400 random scenes that exceeds both 3e-3 and the distance between the reference's own
    two arithmetic models (measured 3.5e-3 .. 1.6e-2 against model spreads of 1.9e-3 .. 8.6e-3); with the features re-derived at every
    step (scene option fourier_resync = 1, the reference's arithmetic) the same scenes agree to <= 7e-4."""
    err, spread, info, *_ = compare_case(seed)
    assert err < 2e-2, "%s: |gpu - oracle| %.2e (model spread %.2e)" % (info, err, spread)
    err1 = compare_case(seed, {"fourier_resync": 1})[0]
    assert err1 < 1e-3, "%s: |gpu - oracle| %.2e with per-step features" % (info, err1)


@pytest.mark.parametrize("seed", [1, 6, 13, 14, 21, 30, 31])
def test_random_scene_with_per_step_features(seed):
    """The 32-wide Fourier-only cases of the fuzz (the networks whose Fourier features the kernels advance by rotation), rendered
    with exact features at every step (scene option fourier_resync = 1: the fp16 position of every sample, like the reference):
    the plain image tolerance holds with a wide margin (measured r02: <= 5e-4 where the default gives up to 4e-3), i.e. the whole
    deviation of those cases is the un-rounded position between two re-derivations.  (The 64-wide ladder cases 23 / 24 / 26 do
    not rotate; their 4e-3 .. 8e-3 against a model spread of 4e-2 .. 2e-1 is the 2^-22 phase accuracy of the hi/lo phase MFMA
    at 512 revolutions behind a chaotic network.)"""
    err, spread, info, *_ = compare_case(seed, {"fourier_resync": 1})
    assert err < TOL_IMG, "%s: |gpu - oracle| %.2e with per-step features (oracle fp16 vs fp32 model %.2e)" % (info, err, spread)


if __name__ == "__main__":  # developer report: python tests/test_fuzz_parity.py [n]  (GPU box)
    for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 32):
        err, spread, info, *_ = compare_case(seed)
        err1 = compare_case(seed, {"fourier_resync": 1})[0]
        print("%2d  gpu-oracle %.2e (per-step features %.2e)  fp16-vs-fp32 models %.2e  %s  %s" % (seed, err, err1, spread, "OK" if err < max(TOL_IMG, spread) else "FAIL", info))
