"""The oracle's arithmetic models against each other, on the CPU (no GPU involved): the fuzz scenes of test_fuzz_parity.py.

  EXACT   the network itself (fp16 weights, nothing else rounded to half): what the reference's PyTorch model computes
  FLOAT   the reference's CUDA algorithm with fp32 accumulation (fp16 position at every sample, fp16 features / activations)
  HALF    the reference's CUDA arithmetic (fp16 accumulation, half Fourier chain)
  DEVICE  FLOAT + the MI355X kernels' Fourier stage: hi + lo phase matrix, and for 32-wide Fourier-only networks features that are
          re-derived every 64 steps and rotated in between (they follow the un-rounded ray there)

What is asserted: the DEVICE model -- the arithmetic the GPU parity tests hold the HIP kernels to with one absolute tolerance -- is, on
every scene, inside max(3e-3, twice the distance of the reference's own models from the exact network): the feature rotation adds a
noise term of the size of the fp16 roundings every model shares, it does not compute something else.  """
import numpy as np
import pytest

import util
from oracle import oracle
from test_fuzz_parity import REFERENCE_SIDE_FACTOR, TOL_IMG, draw_case

# the 12 scenes of a 400-scene sweep (seeds 0..399, r03) on which DEVICE is farthest from FLOAT (3.5e-3 .. 2.2e-2), + the first 60
WORST_OF_400 = [266, 226, 136, 288, 190, 88, 92, 80, 332, 21, 289, 114]


def _models(seed):
    net_kw, scene_kw, W, H = draw_case(seed)
    vn = util.random_network(**net_kw)
    rotates = net_kw["C"] == 32 and net_kw["grid"] is None  # kRotate of fv-srn_amd/csrc/kernels.hpp (the GPU tests ask the library instead)
    out = {}
    for name, acc, kw in (("exact", oracle.ACC_EXACT, {}), ("float", oracle.ACC_FLOAT, {}), ("half", oracle.ACC_HALF, {}),
                          ("device", oracle.ACC_DEVICE, dict(rotation_resync=64 if rotates else 0))):
        out[name], _ = oracle.OracleScene(**kw, **scene_kw).render(oracle.OracleNetwork(vn, acc), W, H)
    return out, rotates


@pytest.mark.parametrize("seed", WORST_OF_400 + list(range(60)) + [1000, 1001, 1002, 1003])  # (1000 ...: 96 / 128 wide, r04)
def test_device_model_is_bounded_by_the_reference_models(seed):
    m, rotates = _models(seed)
    d = lambda a, b: float(np.abs(m[a][:7] - m[b][:7]).max())  # noqa: E731
    dev, flt, hlf = d("device", "exact"), d("float", "exact"), d("half", "exact")
    assert dev < max(TOL_IMG, REFERENCE_SIDE_FACTOR * max(flt, hlf)), (seed, dev, flt, hlf)
    if not rotates:  # without the rotation the DEVICE model differs from FLOAT by the 2^-22 phase matrix and by the last bit of the
        # sample position (one fma per axis instead of the reference's expression): an fp16 rounding flips here and there, which only the
        # networks that are chaotic in the position (FLOAT vs HALF of several percent) turn into more than 1e-3
        assert d("device", "float") < max(1e-3, 0.5 * d("half", "float")), (seed, d("device", "float"), d("half", "float"))
    assert np.array_equal(np.isnan(m["device"][7]), np.isnan(m["float"][7]))


def test_rotation_model_restarts_in_every_depth_segment():
    """segments = K: the step count of the rotation restarts in each of the K step ranges of a ray (kernels.hpp, render_body): with as
    many segments as resync periods the features are re-derived at the same steps plus the segment starts -- a different, equally
    valid image; with resync 1 the segments change nothing at all."""
    net_kw, scene_kw, W, H = draw_case(13)
    vn = util.random_network(**net_kw)
    net = oracle.OracleNetwork(vn, oracle.ACC_DEVICE)
    a, _ = oracle.OracleScene(rotation_resync=1, segments=1, **scene_kw).render(net, W, H)
    b, _ = oracle.OracleScene(rotation_resync=1, segments=4, **scene_kw).render(net, W, H)
    c, _ = oracle.OracleScene(rotation_resync=0, **scene_kw).render(net, W, H)
    assert np.array_equal(np.nan_to_num(a, nan=-1), np.nan_to_num(b, nan=-1)) and np.array_equal(np.nan_to_num(a, nan=-1), np.nan_to_num(c, nan=-1))
    d1, _ = oracle.OracleScene(rotation_resync=64, segments=1, **scene_kw).render(net, W, H)
    d4, _ = oracle.OracleScene(rotation_resync=64, segments=4, **scene_kw).render(net, W, H)
    assert 0 < np.abs(d1[:4] - d4[:4]).max() < 2e-2 and 0 < np.abs(d1[:4] - a[:4]).max() < 2e-2


def test_transfer_function_restatement_matches_library_interpolation():
    """The TF rows are restatement-pinned (the reference holds no TF output): hold the restatement of EvaluateTF (renderer_tf_kernels.cuh:11-70
    over renderer_tf_{identity,texture,piecewise,gaussian}.cuh) to numpy's own interpolation -- a Texture TF is np.interp over the texel centres
    (i + 0.5) / R with clamped ends, a Piecewise TF np.interp over its control points, a Gaussian TF the plain sum of its Gaussians -- with the
    density mapped by (d - min) / (max - min), clamped, nothing below `min`, and the opacity times the step size."""
    rng = np.random.RandomState(12)
    eye, right, up = oracle.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
    base = dict(eye=eye, right=right, up=up, fov_y_radians=0.7, stepsize=1 / 37, density_min=0.15, density_max=0.85)
    d = np.concatenate([rng.uniform(-0.2, 1.2, 4000), [0.15, 0.85, 0.0, 1.0]]).astype(np.float32)
    x = np.clip((d.astype(np.float64) - 0.15) / (0.85 - 0.15), 0.0, 1.0)
    inside = (d >= np.float32(0.15))[:, None]
    step = 1 / 37
    # Texture
    R = 48
    tab = rng.uniform(0, 1, (R, 4)).astype(np.float32)
    tab[:, 3] *= 30
    got = oracle.OracleScene(tf_kind=oracle.TF_TEXTURE, tf_table=tab, **base).evaluate_tf(d)
    want = np.stack([np.interp(x, (np.arange(R) + 0.5) / R, tab[:, k]) for k in range(4)], axis=1)
    want[:, 3] *= step
    assert np.abs(got - want * inside).max() < 2e-5
    # Piecewise: rows (r, g, b, opacity, position), positions ascending, the outer two beyond [0, 1]
    pw = np.array([[0, 0, 0, 0, -1], [0.2, 0.1, 0.8, 0, 0.2], [0.9, 0.5, 0.1, 40, 0.5], [1, 1, 1, 120, 0.9], [1, 1, 1, 120, 2]], np.float32)
    got = oracle.OracleScene(tf_kind=oracle.TF_PIECEWISE, tf_table=pw, **base).evaluate_tf(d)
    want = np.stack([np.interp(x, pw[:, 4], pw[:, k]) for k in range(4)], axis=1)
    want[:, 3] *= step
    assert np.abs(got - want * inside).max() < 2e-4
    # Gaussian: rows (r, g, b, opacity, mean, sigma): sum of colour x exp(-(x - mean)^2 / sigma^2)
    ga = np.array([[0.9, 0.1, 0.1, 30.0, 0.25, 0.08], [0.1, 0.9, 0.2, 60.0, 0.5, 0.05], [0.2, 0.3, 0.95, 90.0, 0.8, 0.1]], np.float32)
    got = oracle.OracleScene(tf_kind=oracle.TF_GAUSSIAN, tf_table=ga, **base).evaluate_tf(d)
    w = np.exp(-((x[:, None] - ga[None, :, 4]) ** 2) / (ga[None, :, 5].astype(np.float64) ** 2))
    want = w @ ga[:, :4].astype(np.float64)
    want[:, 3] *= step
    assert np.abs(got - want * inside).max() < 2e-4
    # Identity: emission = scale x density, opacity = absorption x density x step
    got = oracle.OracleScene(tf_kind=oracle.TF_IDENTITY, tf_scale_absorption=25.0, tf_scale_emission=0.7, **base).evaluate_tf(d)
    want = np.stack([0.7 * x, 0.7 * x, 0.7 * x, 25.0 * x * step], axis=1)
    assert np.abs(got - want * inside).max() < 2e-5
