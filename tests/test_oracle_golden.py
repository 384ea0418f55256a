"""
Pins the CPU oracle (oracle/srn_oracle.c) against golden vectors produced by the reference's own
Python implementation (tests/golden/make_golden.py imports /root/reference/applications/volnet).

Tolerances
  * oracle FLOAT model (fp32 accumulate, what the MI355X kernels compute) vs reference fp32: 5e-4
    -- the only differences left are activations stored as fp16 between layers
  * oracle HALF model (the reference CUDA arithmetic: fp16 accumulate) vs reference fp16 torch: 1e-2,
    the bar of the reference's own unit test (unittests/testSRN.cpp:409-411); 1.5e-2 for the 96- and 128-wide fixtures of round 4
"""
import numpy as np
import pytest

import util
from oracle import oracle


def expected_output(d, meta, key):
    ref = d[key]
    if meta["output_mode"] == "rgbo:direct":  # CUDA eval returns the clamped value (tensorcores.cuh:1063-1072)
        ref = np.concatenate([np.clip(ref[..., :3], 0, 1), np.maximum(ref[..., 3:], 0)], axis=-1)
    return ref


@pytest.mark.parametrize("name", util.golden_names("g1_"))
def test_network_outputs_match_reference_python(name):
    d, meta = util.load_golden(name)
    vn = util.golden_to_volnet(d, meta)
    out_f = oracle.OracleNetwork(vn, oracle.ACC_FLOAT).evaluate(d["positions"], d.get("directions"))
    out_h = oracle.OracleNetwork(vn, oracle.ACC_HALF).evaluate(d["positions"], d.get("directions"))
    assert np.isfinite(out_f).all() and np.isfinite(out_h).all()
    spread = np.abs(expected_output(d, meta, "out_fp32") - expected_output(d, meta, "out_fp16")).max()
    # r05, the deep networks of the reference's study grid (8 .. 22 weight matrices, eval_NetworkConfigsGrid.py:37): the fp16 rounding of the stored
    # activations is amplified layer by layer (measured 0.8e-3 .. 3.0e-3 against 1.0e-3 .. 4.6e-3 between the reference's own fp32 and fp16 passes), so
    # the FLOAT model is held to that spread -- and the EXACT model (the network itself: nothing but the weights rounded to half) pins the restatement
    # of layer loop, latent grid and output parametrisation to the reference's fp32 pass at 1e-5 (measured <= 2e-6)
    deep = len(meta["layers"].split(":")) + 1 >= 8
    assert np.abs(out_f - expected_output(d, meta, "out_fp32")).max() < (max(5e-4, 1.5 * spread) if deep else 5e-4)
    if deep:
        out_e = oracle.OracleNetwork(vn, oracle.ACC_EXACT).evaluate(d["positions"], d.get("directions"))
        assert np.abs(out_e - expected_output(d, meta, "out_fp32")).max() < 1e-5
        assert expected_output(d, meta, "out_fp32").std(axis=0).max() > 1e-2, "the output does not depend on the position: a vacuous fixture"
    # (fp16 accumulation noise grows with the layer width: the reference's bar is stated for its 32-wide test networks; 96 / 128 wide: 1.5e-2)
    wide = int(meta["layers"].split(":")[0]) > 64
    assert np.abs(out_h - expected_output(d, meta, "out_fp16")).max() < (1.5e-2 if wide else 1e-2)
    # the two arithmetic models differ by no more than the reference's fp16 and fp32 paths do themselves
    assert np.abs(out_f - out_h).max() < max(1.5e-2 if wide else 1e-2, 1.5 * spread)


@pytest.mark.parametrize("name", util.golden_names("g2_"))
def test_time_and_ensemble_grids_match_reference_python(name):
    d, meta = util.load_golden(name)
    vn = util.golden_to_volnet(d, meta)
    tes = [(t, 0) for t in meta["times"]] if "times" in meta else meta["time_ensemble"]
    for i, (t, e) in enumerate(tes):
        out_f = oracle.OracleNetwork(vn, oracle.ACC_FLOAT, time=t, ensemble=e).evaluate(d["positions"])
        out_h = oracle.OracleNetwork(vn, oracle.ACC_HALF, time=t, ensemble=e).evaluate(d["positions"])
        assert np.abs(out_f - d["out_fp32"][i]).max() < 5e-4, (t, e)
        assert np.abs(out_h - d["out_fp32"][i]).max() < 1e-2, (t, e)


def test_time_is_clamped_to_keyframe_range():
    d, meta = util.load_golden("g2_time3_c32l4_grid16r8")
    vn = util.golden_to_volnet(d, meta)
    lo = oracle.OracleNetwork(vn, oracle.ACC_FLOAT, time=-3.0).evaluate(d["positions"])
    hi = oracle.OracleNetwork(vn, oracle.ACC_FLOAT, time=17.0).evaluate(d["positions"])
    assert np.array_equal(lo, oracle.OracleNetwork(vn, oracle.ACC_FLOAT, time=0.0).evaluate(d["positions"]))
    assert np.array_equal(hi, oracle.OracleNetwork(vn, oracle.ACC_FLOAT, time=2.0).evaluate(d["positions"]))


def trace_scene_kwargs(d, meta):
    return dict(eye=d["eye"], right=d["right"], up=d["up"], fov_y_radians=meta["fov_y"], stepsize=meta["stepsize"],
                early_out=False, tf_kind=oracle.TF_NONE)


# (fixture, FLOAT tolerance, HALF tolerance): g3 = 48 steps per ray; g3b = step 1/512 through the unit box with the NeRF
# ladder (the sampling of the headline benchmark, rays of up to ~800 steps), SnakeAlt and ReLU
TRACE_FIXTURES = [("g3_trace_rgbo_32x32", 1e-4, 5e-3), ("g3b_trace_rgbo_64x64_s512_snakealt", 2e-4, 1e-2),
                  ("g3b_trace_rgbo_64x64_s512_relu", 2e-4, 1e-2)]


@pytest.mark.parametrize("name,tol_f,tol_h", TRACE_FIXTURES)
def test_ray_march_matches_reference_python_trace(name, tol_f, tol_h):
    """DVR loop + box clipping + Beer-Lambert blending vs Raytracing._full_trace_forward (rgbo mode).
    Pixel (0,0) carries the sentinel ray of the generator and is ignored."""
    d, meta = util.load_golden(name)
    vn = util.golden_to_volnet(d, meta, box_min=meta["box_min"], box_size=meta["box_size"])
    scene = oracle.OracleScene(**trace_scene_kwargs(d, meta))
    for mode, tol in ((oracle.ACC_FLOAT, tol_f), (oracle.ACC_HALF, tol_h)):
        img, count = scene.render(oracle.OracleNetwork(vn, mode), meta["W"], meta["H"])
        diff = np.abs(img[:4] - d["image"])
        diff[:, 0, 0] = 0
        assert diff.max() < tol, (mode, diff.max())
        assert count == scene.count_samples(oracle.OracleNetwork(vn, mode), meta["W"], meta["H"])
    assert d["image"][3].max() > 0.5  # the fixture is not an empty image


@pytest.mark.parametrize("name", util.golden_names("g4_"))
def test_adjoint_gradient_matches_reference_autograd(name):
    """GRADIENT_MODE_ADJOINT_METHOD (renderer_volume_tensorcores.cuh:1198-1540) as restated in the oracle, against torch.autograd on the
    reference's own PyTorch model in fp32 (tests/golden/make_golden.py, G4): pins the restatement of the backward pass.  Smooth
    activations agree everywhere (measured <= 7e-4 of the largest gradient in the FLOAT model, 3e-3 in the HALF model); ReLU networks
    have samples whose fp16-rounded pre-activation sits on the other side of the kink: by quantile there."""
    d, meta = util.load_golden(name)
    vn = util.golden_to_volnet(d, meta)
    ref = d["grad_fp32"]
    scale = float(np.abs(ref).max())
    assert scale > 1e-2
    for mode, tol in ((oracle.ACC_FLOAT, 2e-3), (oracle.ACC_HALF, 8e-3)):
        err = np.abs(oracle.OracleNetwork(vn, mode).adjoint_gradient(d["positions"]) - ref)
        if meta["activation"] == "ReLU":
            assert np.median(err) < 0.1 * tol * scale and np.percentile(err, 95) < tol * scale, (mode, np.median(err), np.percentile(err, 95))
        else:
            assert err.max() < tol * scale, (mode, err.max(), scale)
    assert np.abs(oracle.OracleNetwork(vn, oracle.ACC_FLOAT).evaluate(d["positions"])[:, :1] - d["out_fp32"]).max() < 1e-3


def test_camera_on_a_sphere_matches_generator_frame():
    d, meta = util.load_golden("g3_trace_rgbo_32x32")
    eye, right, up = oracle.camera_on_a_sphere(meta["orientation"], (0, 0, 0), meta["pitch"], meta["yaw"], meta["distance"])
    assert np.allclose(eye, d["eye"], atol=1e-7) and np.allclose(right, d["right"], atol=1e-7) and np.allclose(up, d["up"], atol=1e-7)
    # rays generated from (eye,right,up) reproduce the stored explicit rays (except the sentinel)
    W, H = meta["W"], meta["H"]
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    front = np.cross(up, right)
    ty = np.float32(np.tan(meta["fov_y"] / 2))
    dirs = front + ((2 * (xs + 0.5) / W - 1) * ty * (W / H))[..., None] * right + ((2 * (ys + 0.5) / H - 1) * ty)[..., None] * up
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    err = np.abs(dirs - d["ray_dir"])
    err[0, 0] = 0
    assert err.max() < 1e-6


def test_half_conversion_round_trip_and_ties():
    l = oracle.lib()
    for bits in list(range(0, 0x7c00, 37)) + [0x0001, 0x03ff, 0x0400, 0x7bff, 0x8001, 0xfbff]:
        assert l.oracle_float_to_half(l.oracle_half_to_float(bits)) == bits
    assert l.oracle_float_to_half(np.float32(1.0 + 2.0 ** -11)) == 0x3c00  # tie -> even
    assert l.oracle_float_to_half(np.float32(1.0 + 3 * 2.0 ** -11)) == 0x3c02
    assert l.oracle_float_to_half(np.float32(65520.0)) == 0x7c00
    a = np.random.RandomState(0).randn(4096).astype(np.float32)
    ours = np.array([l.oracle_float_to_half(v) for v in a], np.uint16)
    assert np.array_equal(ours, a.astype(np.float16).view(np.uint16))
