"""
The C++ pybind11 `pyrenderer` module (fv-srn_amd/pyrenderer/) -- the reference's own Python-visible surface for
this path -- driven exactly like the reference's callers drive it:
  * export_to_pyrenderer's call sequence (applications/volnet/network.py:798-897)
  * LoadedModel.render_network's call sequence (applications/volnet/inference.py:529-625)
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

import util

sys.path.insert(0, os.path.join(util.ROOT, "fv-srn_amd", "pyrenderer"))
pr = pytest.importorskip("pyrenderer", reason="pyrenderer module not built (python fv-srn_amd/pyrenderer/build.py)")
from fvsrn_amd import volnet_io  # noqa: E402


def export_like_the_reference(d, meta, encoding=None):
    n = pr.SceneNetwork()
    n.input.has_direction = False
    n.input.set_fourier_matrix_from_tensor(torch.from_numpy(d["B"]), True)
    n.input.has_time = meta.get("use_time_direct", False)
    n.output.output_mode = pr.SceneNetwork.OutputParametrization.OutputModeFromString(meta["output_mode"])
    enc = encoding if encoding is not None else pr.SceneNetwork.LatentGrid.Float
    if "grid" in d:
        g = pr.SceneNetwork.LatentGridTimeAndEnsemble(time_min=0, time_num=1, time_step=1, ensemble_min=0, ensemble_num=0)
        g.set_time_grid_from_torch(0, torch.from_numpy(d["grid"]), enc)
        n.latent_grid = g
    elif "grid_time" in d:
        tg, eg = d["grid_time"], d.get("grid_ensemble", np.zeros((0,)))
        g = pr.SceneNetwork.LatentGridTimeAndEnsemble(time_min=0, time_num=len(tg), time_step=1, ensemble_min=0, ensemble_num=len(eg))
        for i in range(len(tg)):
            g.set_time_grid_from_torch(i, torch.from_numpy(tg[i:i + 1]), enc)
        for i in range(len(eg)):
            g.set_ensemble_grid_from_torch(i, torch.from_numpy(eg[i:i + 1]), enc)
        assert g.is_valid()
        n.latent_grid = g
    act = pr.SceneNetwork.Layer.ActivationFromString(meta["activation"])
    k = len(meta["layers"].split(":"))
    for i in range(k):
        n.add_layer(torch.from_numpy(d["W%d" % i]), torch.from_numpy(d["b%d" % i]), act, meta["activation_param"])
    n.add_layer(torch.from_numpy(d["W%d" % k]), torch.from_numpy(d["b%d" % k]), pr.SceneNetwork.Layer.Activation.NONE)
    n.box_min = pr.float3(0, 0, 0)
    n.box_size = pr.float3(1, 1, 1)
    assert n.valid()
    return n


@pytest.mark.parametrize("name", ["g1_c32l4_snakealt_density", "g1_c32l4_grid16r8_snakealt_rgbo", "g2_time3_ens2_c32l4_grid32r8",
                                  "g2_time3_passtime_c32l4_grid16r8"])
def test_export_sequence_writes_reference_format(tmp_path, name):
    d, meta = util.load_golden(name)
    n = export_like_the_reference(d, meta)
    path = str(tmp_path / "net.volnet")
    n.save(path)
    assert open(path, "rb").read() == volnet_io.save_volnet(util.golden_to_volnet(d, meta))
    m = pr.SceneNetwork.load(path)
    assert m.num_layers() == n.num_layers() and m.num_parameters() == n.num_parameters()
    assert m.input.num_fourier_features() == d["B"].shape[0]
    assert m.get_layer(0).channels_out == int(meta["layers"].split(":")[0])
    assert m.get_layer(m.num_layers() - 1).activation == pr.SceneNetwork.Layer.Activation.NONE
    assert m.output.output_mode == n.output.output_mode


def test_value_types_and_enums():
    a = pr.double3(1, 2, 3) + pr.double3(1, 1, 1)
    assert (a.x, a.y, a.z) == (2, 3, 4) and str(pr.float3(1, 2, 3)) == "(1, 2, 3)"
    assert pr.SceneNetwork.OutputParametrization.OutputModeFromString("rgbo:direct") == pr.SceneNetwork.OutputParametrization.RGBO_DIRECT
    assert pr.SceneNetwork.Layer.ActivationFromString("SnakeAlt") == pr.SceneNetwork.Layer.SnakeAlt
    with pytest.raises(RuntimeError):
        pr.SceneNetwork.Layer.ActivationFromString("Tanh")
    lg = pr.SceneNetwork.LatentGridTimeAndEnsemble(2, 3, 2, 5, 2)
    assert lg.time_max_inclusive == 6 and lg.ensemble_max_inclusive == 6
    assert lg.interpolate_time(5.0) == 1.5 and lg.interpolate_time(99) == 2 and lg.interpolate_ensemble(0) == 0


def test_entries_outside_the_hot_path_exist_and_raise():
    """world2screen (OpenGL matrices), importance_sampling* (training-data samplers) and the curvature
    evaluation of volumes that do not estimate it are present on the surface and raise a RuntimeError, rather than being absent attributes."""
    cam = pr.CameraOnASphere()
    with pytest.raises(RuntimeError, match="world2screen"):
        cam.world2screen(64, 64, [])
    vol = pr.VolumeInterpolationNetwork()
    with pytest.raises(RuntimeError, match="importance_sampling"):
        vol.importance_sampling(10, None, 0.1, 1, 0, 0.0, 1.0, "float")
    with pytest.raises(RuntimeError, match="importance_sampling_with_probability_grid"):
        vol.importance_sampling_with_probability_grid(10, None, None, 1.0, 0.1, 1, 0, 0.0, 1.0)
    with pytest.raises(RuntimeError, match="curvature"):   # (networks that predict the curvature provide it: see below)
        pr.VolumeInterpolationGrid().evaluate_with_gradients_and_curvature(torch.zeros(1, 3))


def test_load_from_json_builds_the_module_tree(tmp_path):
    scene = {"version": 1, "root": "Simple",
             "ImageEvaluator": {"Simple": {"selectedCamera": "Sphere", "selectedVolume": "SRN", "selectedRayEvaluator": "DVR",
                                           "samplesPerIterationLog2": 0, "useTonemapping": False}},
             "camera": {"Sphere": {"orientation": "Zp", "center": [0.1, 0.2, 0.3], "pitch": 0.3, "yaw": 1.1, "distance": 2.5, "fovY": 0.7}},
             "RayEvaluation": {"DVR": {"stepsize": 0.5, "stepsizeIsObjectSpace": True, "minDensity": 0.1, "maxDensity": 0.9,
                                       "earlyOut": False, "selectedTF": "Gaussian", "selectedBRDF": "Lambert"}},
             "tf": {"Gaussian": {"absorptionScaling": 2.0, "points": [[1, 0, 0, 10, 0.3, 0.1], [0, 1, 0, 20, 0.7, 0.05]]}},
             "brdf": {"Lambert": {"enablePhong": False}}, "blending": {"blending": {"blending": "Alpha"}},
             "volume": {"SRN": {}}}
    p = tmp_path / "scene.json"
    p.write_text(json.dumps(scene))
    ev = pr.load_from_json(str(p))
    cam = ev.camera
    assert cam.orientation == pr.CameraOnASphere.Zp and cam.pitchYawDistance.value.z == 2.5 and cam.center.value.y == 0.2
    assert abs(cam.fov_y_radians - 0.7) < 1e-12
    r = ev.ray_evaluator
    assert abs(r.stepsize - 0.5 / 256) < 1e-12 and r.min_density == 0.1 and r.max_density == 0.9 and not r.early_out
    assert r.blending.blendMode == pr.Blending.Alpha
    t = r.tf.tensor
    assert tuple(t.shape) == (1, 2, 6) and float(t[0, 0, 3]) == 20.0 and float(t[0, 1, 3]) == 40.0
    assert isinstance(ev.volume, pr.VolumeInterpolationNetwork)
    with pytest.raises(RuntimeError, match="No network loaded"):
        ev.volume.current_network()


def test_gaussian_tf_flags_of_scene_files(tmp_path):
    """scaleWithGradient / usePiecewiseAnalyticIntegration (transfer_function_gaussian.cpp:238-239) load into the TF object; both at
    once are refused when the kernel variant is selected, like the reference's getDefines (:296-297)."""
    body = {"absorptionScaling": 2.0, "points": [[1, 0, 0, 10, 0.3, 0.1], [0, 1, 0, 20, 0.7, 0.05]]}
    for flags, want in (({}, (False, False)), ({"scaleWithGradient": True}, (True, False)), ({"usePiecewiseAnalyticIntegration": True}, (False, True))):
        p = tmp_path / "g.json"
        p.write_text(json.dumps(_scene_json("Gaussian", dict(body, **flags), volume="SRN")))
        tf = pr.load_from_json(str(p)).ray_evaluator.tf
        assert (tf.scale_with_gradient, tf.piecewise_analytic_integraton) == want
        assert tf.requires_gradients() == want[0]
    tf.scale_with_gradient = True
    with pytest.raises(RuntimeError, match="incompatible"):
        tf.get_max_absorption()  # (any call that builds the TF's device description)


def _scene_json(tf_name, tf_body, brdf=None, volume="Grid"):
    return {"version": 1, "root": "Simple",
            "ImageEvaluator": {"Simple": {"selectedCamera": "Sphere", "selectedVolume": volume, "selectedRayEvaluator": "DVR",
                                          "samplesPerIterationLog2": 0, "useTonemapping": False}},
            "camera": {"Sphere": {"orientation": "Ym", "center": [0, 0, 0], "pitch": 0.4, "yaw": 0.7, "distance": 1.6, "fovY": 0.7853981633974483}},
            "RayEvaluation": {"DVR": {"stepsize": 1 / 48, "stepsizeIsObjectSpace": False, "minDensity": 0.0, "maxDensity": 1.0,
                                      "earlyOut": True, "selectedTF": tf_name, "selectedBRDF": "Lambert"}},
            "tf": {tf_name: tf_body}, "brdf": {"Lambert": brdf or {"enablePhong": False}},
            "blending": {"blending": {"blending": "BeerLambert"}}, "volume": {volume: {}}}


PIECEWISE_JSON = {"absorptionScaling": 25.0, "colorPoints": [[0.0033, 1e-6, 1e-6, 1e-6], [0.44, 1.0, 0.04, 0.04], [0.7558, 0.9167, 0.9167, 0.1168]],
                  "opacityPoints": [[0.0568, 0.0], [0.3645, 0.66], [0.7, 0.0], [0.75, 0.0], [0.8, 0.0], [1.0, 0.9]]}
TEXTURE_JSON = {"absorptionScaling": 40.0, "preintegrationMode": "None",
                "colorPoints": [[0.1, 0.2, 0.1, 0.9], [0.5, 1.0, 0.5, 0.0], [0.9, 1.0, 1.0, 1.0]],
                "opacityPoints": [float(0.5 + 0.5 * np.sin(i / 40.0)) for i in range(256)]}


def test_scene_files_with_piecewise_and_texture_tfs_load(tmp_path):
    """Scene files of the reference select the ground-truth volume and a Piecewise / Texture / Gaussian TF; the device
    tables must equal the restatement of TransferFunctionPiecewiseLinear::computeTensor / TransferFunctionTexture::computeTexture."""
    from oracle import oracle
    p = tmp_path / "pw.json"
    p.write_text(json.dumps(_scene_json("Piecewise", PIECEWISE_JSON)))
    ev = pr.load_from_json(str(p))
    assert isinstance(ev.volume, pr.VolumeInterpolationGrid) and ev.volume.source() == pr.VolumeInterpolationGrid.Empty
    t = ev.ray_evaluator.tf.tensor.numpy()[0]
    ref = oracle.tf_piecewise_table(PIECEWISE_JSON["colorPoints"], PIECEWISE_JSON["opacityPoints"], 25.0)
    assert t.shape == ref.shape and np.abs(t - ref).max() < 1e-6
    assert ref[0, 4] == -1.0 and ref.shape[0] < 3 + 6 + 2  # sentinel at -1, a run of zero-absorption points was purged
    p = tmp_path / "tex.json"
    p.write_text(json.dumps(_scene_json("Texture", TEXTURE_JSON)))
    t = pr.load_from_json(str(p)).ray_evaluator.tf.tensor.numpy()[0]
    ref = oracle.tf_texture_table(TEXTURE_JSON["colorPoints"], TEXTURE_JSON["opacityPoints"], 40.0)
    assert t.shape == (256, 4) and np.abs(t - ref).max() < 1e-6
    pre = dict(TEXTURE_JSON, preintegrationMode="Preintegrate2D")
    p.write_text(json.dumps(_scene_json("Texture", pre)))
    assert pr.load_from_json(str(p)).ray_evaluator.tf.preintegration_mode == pr.TransferFunctionTexture.PreintegrationMode.Preintegrate2D


REFERENCE_SCENES = "/root/reference/applications/config-files"


@pytest.mark.skipif(not os.path.isdir(REFERENCE_SCENES), reason="reference checkout not present (it is not on the GPU box)")
def test_reference_scene_files_load_unchanged():
    """Every DVR scene file the reference ships loads; the ones outside this build fail with a precise message."""
    import glob
    from oracle import oracle
    loaded = 0
    for f in sorted(glob.glob(os.path.join(REFERENCE_SCENES, "*.json"))):
        d = json.load(open(f))
        sel = d["ImageEvaluator"]["Simple"].get("selectedRayEvaluator", "DVR")
        tfsel = d["RayEvaluation"].get("DVR", {}).get("selectedTF")
        jt = d.get("tf", {}).get(tfsel, {})
        expect_error = None
        if sel != "DVR":
            expect_error = "ray evaluator"
        if expect_error:
            with pytest.raises(RuntimeError, match=expect_error):
                pr.load_from_json(f)
            continue
        ev = pr.load_from_json(f)
        loaded += 1
        if tfsel == "Piecewise":
            ref = oracle.tf_piecewise_table(jt["colorPoints"], jt["opacityPoints"], jt["absorptionScaling"])
            assert np.abs(ev.ray_evaluator.tf.tensor.numpy()[0] - ref).max() < 1e-5, f
        elif tfsel == "Texture":
            ref = oracle.tf_texture_table(jt["colorPoints"], jt["opacityPoints"], jt["absorptionScaling"])
            assert np.abs(ev.ray_evaluator.tf.tensor.numpy()[0] - ref).max() < 1e-5, f
        b = d.get("brdf", {}).get("Lambert", {})
        assert ev.ray_evaluator.brdf.enable_phong == b.get("enablePhong", False), f
    assert loaded >= 22


def test_extract_color_rejects_host_tensors():
    # the reference checks CHECK_CUDA(inputTensor) (iimage_evaluator.cpp:29); there is no CPU path here either
    with pytest.raises(RuntimeError, match="GPU"):
        pr.ImageEvaluatorSimple.Extract_color(torch.rand(1, 8, 5, 7), False, 1.0, pr.ImageEvaluatorSimple.ChannelMode.Color)


@pytest.mark.gpu
def test_extract_color_channels():
    raw = torch.rand(2, 8, 5, 7, device="cuda")
    ch = pr.ImageEvaluatorSimple.ChannelMode
    c = pr.ImageEvaluatorSimple.Extract_color(raw, False, 1.0, ch.Color)
    assert torch.equal(c, raw[:, :4])
    n = pr.ImageEvaluatorSimple.Extract_color(raw, False, 1.0, ch.Normal)
    assert torch.allclose(n[:, :3], raw[:, 4:7] * 0.5 + 0.5) and torch.equal(n[:, 3], raw[:, 3])
    m = pr.ImageEvaluatorSimple.Extract_color(raw, False, 1.0, ch.Mask)
    assert torch.equal(m[:, 0], raw[:, 3]) and bool((m[:, 3] == 1).all())
    dpt = pr.ImageEvaluatorSimple.Extract_color(raw[:1], False, 1.0, ch.Depth)
    assert abs(float(dpt[:, 0].min())) < 1e-6 and abs(float(dpt[:, 0].max()) - 1) < 1e-6
    t = pr.ImageEvaluatorSimple.Extract_color(raw, True, 2.0, ch.Color)
    x = raw[:, :3] / 2.0
    ref = ((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14)).clamp(0, 1) ** (1 / 2.4)
    assert torch.allclose(t[:, :3], ref, atol=1e-6) and torch.equal(t[:, 3], raw[:, 3])


@pytest.mark.gpu
def test_render_network_sequence_matches_oracle():
    """inference.py:529-625: set camera / stepsize / volume, render, extract_color -- vs the CPU oracle."""
    from oracle import oracle
    d, meta = util.load_golden("g3_trace_rgbo_32x32")
    n = export_like_the_reference(d, meta)
    n.box_min = pr.float3(*meta["box_min"])
    n.box_size = pr.float3(*meta["box_size"])
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(n)
    ev = pr.ImageEvaluatorSimple()
    ev.volume = vol
    ev.camera.orientation = pr.CameraOnASphere.Ym
    ev.camera.pitchYawDistance.value = pr.double3(meta["pitch"], meta["yaw"], meta["distance"])
    ev.camera.fov_y_radians = meta["fov_y"]
    ev.ray_evaluator.stepsize = meta["stepsize"]
    ev.ray_evaluator.early_out = False
    timer = pr.GPUTimer()
    timer.start()
    img = ev.render(meta["W"], meta["H"])
    rgba = ev.extract_color(img)
    timer.stop()
    pr.sync()
    assert timer.elapsed_milliseconds() > 0
    assert tuple(img.shape) == (1, 8, 32, 32) and tuple(rgba.shape) == (1, 4, 32, 32)
    diff = (rgba[0].cpu().numpy() - d["image"])
    diff[:, 0, 0] = 0
    assert np.abs(diff).max() < 3e-3
    # evaluate(): unit-box positions like IVolumeInterpolation::evaluate
    pos = torch.rand(500, 3, device="cuda")
    out = vol.evaluate(pos)
    vn = util.golden_to_volnet(d, meta, box_min=meta["box_min"], box_size=meta["box_size"])
    ref = oracle.OracleNetwork(vn, oracle.ACC_FLOAT).evaluate(pos.cpu().numpy() * np.array(meta["box_size"], np.float32) + np.array(meta["box_min"], np.float32))
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-3


@pytest.mark.gpu
def test_batched_camera_matrices_render_one_image_per_batch_entry():
    """The batch dimension B of ImageEvaluatorSimple::render (renderer_image_evaluator_simple.cuh:36-127 `virtual_size.z`,
    iimage_evaluator.cpp:138-165 computeBatchCount) through externally set (B,3,3) camera matrices (camera.cpp:242-258): batch entry b is
    the image of camera b; the module loops over launches."""
    d, meta = util.load_golden("g3_trace_rgbo_32x32")
    n = export_like_the_reference(d, meta)
    n.box_min = pr.float3(*meta["box_min"])
    n.box_size = pr.float3(*meta["box_size"])
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(n)
    ev = pr.ImageEvaluatorSimple()
    ev.volume = vol
    ev.camera.orientation = pr.CameraOnASphere.Ym
    ev.camera.fov_y_radians = meta["fov_y"]
    ev.ray_evaluator.stepsize = meta["stepsize"]
    singles, mats = [], []
    for yaw in (0.2, 1.4, 2.9):
        ev.camera.pitchYawDistance.value = pr.double3(meta["pitch"], yaw, meta["distance"])
        mats.append(ev.camera.get_parameters())
        singles.append(ev.render(40, 24).clone())
    assert ev.compute_batch_count() == 1
    ev.camera.set_parameters(torch.cat(mats, dim=0))
    assert ev.compute_batch_count() == 3 and tuple(ev.camera.get_parameters().shape) == (3, 3, 3)
    img = ev.render(40, 24)
    assert tuple(img.shape) == (3, 8, 24, 40) and tuple(ev.extract_color(img).shape) == (3, 4, 24, 40)
    for b in range(3):
        # (r05: the batch is ONE call into the library and, up to eight entries, one launch -- without the depth segments a small single launch is cut
        # into: the same samples, re-associated sums)
        assert torch.equal(torch.isnan(img[b]), torch.isnan(singles[b][0]))
        assert float((torch.nan_to_num(img[b]) - torch.nan_to_num(singles[b][0])).abs().max()) < 2e-4
    assert float((img[0, :4] - img[1, :4]).abs().max()) > 1e-2
    ev.camera.set_parameters(torch.empty(0))  # back to pitch / yaw / distance
    assert ev.compute_batch_count() == 1 and tuple(ev.render(40, 24).shape) == (1, 8, 24, 40)


@pytest.mark.gpu
@pytest.mark.parametrize("world,stripe", [(2, 16), (4, 8)])
def test_render_stripes_through_the_module_api(world, stripe):
    """Multi-GPU frames behind the module API (VERDICT r04 missing 2): ImageEvaluatorSimple.render_stripes(width, height, rank, world, stripe) is this
    rank's round-robin row stripes of the frame render() gives, compact; the stripes of all ranks -- stacked like all_gather_into_tensor / gather deliver
    them -- go back into image order with Assemble_stripes and equal the whole frame bit for bit (batched cameras included)."""
    d, meta = util.load_golden("g3_trace_rgbo_32x32")
    n = export_like_the_reference(d, meta)
    n.box_min = pr.float3(*meta["box_min"])
    n.box_size = pr.float3(*meta["box_size"])
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(n)
    ev = pr.ImageEvaluatorSimple()
    ev.volume = vol
    ev.camera.orientation = pr.CameraOnASphere.Ym
    ev.camera.fov_y_radians = meta["fov_y"]
    ev.ray_evaluator.stepsize = meta["stepsize"]
    mats = []
    for yaw in (0.2, 1.4):
        ev.camera.pitchYawDistance.value = pr.double3(meta["pitch"], yaw, meta["distance"])
        mats.append(ev.camera.get_parameters())
    ev.camera.set_parameters(torch.cat(mats, dim=0))
    W, H = 40, 64
    full = ev.render(W, H).clone()
    parts = [ev.render_stripes(W, H, r, world, stripe).clone() for r in range(world)]
    for r in range(world):
        rows = pr.ImageEvaluatorSimple.stripe_rows(H, stripe, r, world)
        assert tuple(parts[r].shape) == (2, 8, rows, W) and rows == H // world
    frame = pr.ImageEvaluatorSimple.Assemble_stripes(torch.stack(parts), H, stripe)
    assert torch.equal(torch.nan_to_num(frame, nan=-7.0), torch.nan_to_num(full, nan=-7.0))  # (both sides: two cameras per launch, one depth segment)
    assert float(full[:, 3].max()) > 0.05
    with pytest.raises(Exception):
        ev.render_stripes(W, H, world, world, stripe)
    with pytest.raises(Exception):
        ev.render_stripes(W, H, 0, world, 12)


@pytest.mark.gpu
def test_shaded_configuration_through_the_module_api():
    """The shaded scene files of the reference (e.g. config-files/ejecta1024-v7-shaded.json) select finite-difference
    gradients on the volume and Phong shading on the BRDF: same image as the C ABI driven directly, and as the oracle."""
    from oracle import oracle
    from fvsrn_amd import capi
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="density", seed=31, box_min=(-0.5, -0.5, -0.5), fourier_std=0.35)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "shaded_test.volnet")
    open(path, "wb").write(volnet_io.save_volnet(vn))
    ev = pr.ImageEvaluatorSimple()
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(pr.SceneNetwork.load(path))
    vol.gradient_mode = pr.VolumeInterpolationNetwork.GradientMode.FINITE_DIFFERENCES
    vol.finite_differences_stepsize = 1 / 16
    ev.volume = vol
    ev.camera.orientation = pr.CameraOnASphere.Ym
    ev.camera.pitchYawDistance.value = pr.double3(0.4, 0.7, 1.6)
    ev.camera.fov_y_radians = float(np.deg2rad(45.0))
    ev.ray_evaluator.stepsize = 1 / 48
    tf = pr.TransferFunctionIdentity()
    tf.absorption_emission.value = pr.double2(20.0, 1.0)
    ev.ray_evaluator.tf = tf
    b = ev.ray_evaluator.brdf
    b.enable_phong = True
    b.ambient.value, b.specular.value, b.magnitude_center.value, b.magnitude_radius.value = 0.2, 0.4, 0.6, 0.5
    b.specular_exponent.value = 8
    b.light_follows_camera = True
    b.light_type = pr.BRDFLambert.LightType.Point
    img = ev.render(40, 24).cpu().numpy()[0]
    eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
    kw = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1 / 48, early_out=True,
              tf_kind=oracle.TF_IDENTITY, tf_scale_absorption=20.0, tf_scale_emission=1.0, gradient_mode=1,
              finite_differences_stepsize=1 / 16,
              brdf=dict(enable_phong=True, ambient=0.2, specular=0.4, magnitude_center=0.6, magnitude_radius=0.5, specular_exponent=8,
                        light_type=0, light=tuple(float(v) for v in eye)))
    ref, _ = oracle.OracleScene(**kw).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT), 40, 24)
    assert np.abs(ref[4:7]).max() > 0.05 and img[3].max() > 0.05
    assert np.abs(img[:7] - ref[:7]).max() < 1.2e-2
    # GradientMode.ADJOINT_METHOD through the same module tree: the analytic gradient, against the oracle's backward pass
    vol.gradient_mode = pr.VolumeInterpolationNetwork.GradientMode.ADJOINT_METHOD
    img2 = ev.render(40, 24).cpu().numpy()[0]
    kw.update(gradient_mode=2)
    ref2, _ = oracle.OracleScene(**kw).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT), 40, 24)
    assert np.abs(ref2[4:7]).max() > 0.05
    assert np.abs(img2[:7] - ref2[:7]).max() < 6e-3
    assert vol.current_network().compute_max_warps(False, True) <= vol.current_network().compute_max_warps(False, False)
    # evaluate_with_gradients in the same mode: analytic gradients w.r.t. the unit-box position (volume_interpolation.cpp:128-243)
    import torch
    pos = torch.rand(500, 3, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    dens, grad = vol.evaluate_with_gradients(pos)
    assert tuple(dens.shape) == (500, 1) and tuple(grad.shape) == (500, 3)
    world = pos.cpu().numpy() - 0.5   # the box of this network is [-0.5, 0.5]^3; the tensor API takes unit-box positions
    gref = oracle.OracleNetwork(vn, oracle.ACC_FLOAT).adjoint_gradient(world)
    assert np.abs(gref).max() > 0.01
    assert np.abs(grad.cpu().numpy() - gref).max() < 3e-3 * max(1.0, np.abs(gref).max())
    assert float((dens - vol.evaluate(pos)).abs().max()) < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("tf_name", ["Piecewise", "Texture"])
def test_scene_file_render_matches_oracle(tmp_path, tf_name):
    """A scene file in the reference's format (ground-truth volume selected, Piecewise / Texture TF, Phong-shaded BRDF with
    a light that follows the camera) + a .volnet: load_from_json, attach the network like inference.py:598, render."""
    from oracle import oracle
    from fvsrn_amd import capi
    brdf = {"enablePhong": True, "enableMagnitudeScaling": False, "ambient": 0.3, "specular": 0.5, "magnitudeCenter": 0.6,
            "magnitudeRadius": 0.5, "specularExponent": 4, "lightFollowsCamera": True, "lightType": "Directional"}
    body = PIECEWISE_JSON if tf_name == "Piecewise" else TEXTURE_JSON
    p = tmp_path / "scene.json"
    p.write_text(json.dumps(_scene_json(tf_name, body, brdf)))
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="densitygrad", seed=8, box_min=(-0.5, -0.5, -0.5), fourier_std=0.35)
    path = str(tmp_path / "net.volnet")
    open(path, "wb").write(volnet_io.save_volnet(vn))
    ev = pr.load_from_json(str(p))
    with pytest.raises(RuntimeError, match="No volume specified"):
        ev.render(40, 24)  # the scene's ground-truth grid volume (.cvol) is not on disk: empty source, like the reference
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(pr.SceneNetwork.load(path))
    ev.volume = vol
    img = ev.render(40, 24).cpu().numpy()[0]
    eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
    front = np.cross(up, right)
    table = (oracle.tf_piecewise_table(body["colorPoints"], body["opacityPoints"], body["absorptionScaling"]) if tf_name == "Piecewise"
             else oracle.tf_texture_table(body["colorPoints"], body["opacityPoints"], body["absorptionScaling"]))
    kw = dict(eye=eye, right=right, up=up, fov_y_radians=0.7853981633974483, stepsize=1 / 48, early_out=True,
              tf_kind=oracle.TF_PIECEWISE if tf_name == "Piecewise" else oracle.TF_TEXTURE, tf_table=table,
              brdf=dict(enable_phong=True, ambient=0.3, specular=0.5, magnitude_center=0.6, magnitude_radius=0.5, specular_exponent=4,
                        light_type=1, light=tuple(float(v) for v in front)))
    ref, _ = oracle.OracleScene(**kw).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT), 40, 24)
    assert img[3].max() > 0.05
    assert np.abs(img[:7] - ref[:7]).max() < 3e-3


@pytest.mark.gpu
def test_tensor_apis_generate_rays_and_tf_evaluate():
    """camera.generate_rays + tf.evaluate: the calls the reference's Python ray tracer makes (raytracing.py:206-224, :286-297)."""
    from oracle import oracle
    cam = pr.CameraOnASphere()
    cam.orientation = pr.CameraOnASphere.Ym
    cam.pitchYawDistance.value = pr.double3(0.4, 0.7, 1.6)
    start, direction = cam.generate_rays(48, 32)
    assert tuple(start.shape) == (1, 32, 48, 3) and tuple(direction.shape) == (1, 32, 48, 3)
    assert torch.allclose(direction.norm(dim=-1), torch.ones(1, 32, 48, device="cuda"), atol=1e-6)
    origin = cam.get_origin()
    assert torch.allclose(start[0, 5, 7].cpu(), torch.tensor([origin.x, origin.y, origin.z], dtype=torch.float32))
    tf = pr.TransferFunctionGaussian()
    table = np.array([[0.9, 0.1, 0.1, 30.0, 0.25, 0.08], [0.1, 0.9, 0.2, 60.0, 0.5, 0.05]], np.float32)
    tf.tensor = torch.from_numpy(table[None])
    dens = torch.rand(1000, 1, device="cuda")
    out = tf.evaluate(dens, 0.1, 0.9).cpu().numpy()
    eye, right, up = oracle.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
    ref = oracle.OracleScene(eye=eye, right=right, up=up, fov_y_radians=0.78, stepsize=1.0, density_min=0.1, density_max=0.9,
                             tf_kind=oracle.TF_GAUSSIAN, tf_table=table).evaluate_tf(dens.cpu().numpy())
    assert np.abs(out - ref).max() < 2e-5 * np.abs(ref).max()


def test_protocol_png_writer(tmp_path):
    sys.path.insert(0, os.path.join(util.ROOT, "tools"))
    import render_protocol
    img = (np.arange(6 * 5 * 4) % 256).astype(np.uint8).reshape(6, 5, 4)
    path = str(tmp_path / "a.png")
    render_protocol.write_png(path, img)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and data[12:16] == b"IHDR"
    import struct
    import zlib
    w, h, depth, ctype = struct.unpack(">IIBB", data[16:26])
    assert (w, h, depth, ctype) == (5, 6, 8, 6)
    i = data.index(b"IDAT")
    n = struct.unpack(">I", data[i - 4:i])[0]
    raw = zlib.decompress(data[i + 4:i + 4 + n])
    rows = [raw[y * (1 + 5 * 4) + 1:(y + 1) * (1 + 5 * 4)] for y in range(6)]
    assert b"".join(rows) == img.tobytes()


@pytest.mark.gpu
def test_reference_timing_protocol_runs(tmp_path):
    """tools/render_protocol.py = eval_NetworkConfigsGrid.py:100-140 on the drop-in module: loads a .volnet, rotates the
    camera, times render + extract_color, writes frames and statistics."""
    sys.path.insert(0, os.path.join(util.ROOT, "tools"))
    import render_protocol
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="density", seed=2, box_min=(-0.5, -0.5, -0.5))
    path = str(tmp_path / "net.volnet")
    open(path, "wb").write(volnet_io.save_volnet(vn))
    out = str(tmp_path / "out")
    args = render_protocol.argparse.Namespace(volnet=path, scene=None, out=out, frames=True, width=64, height=48, cameras=4,
                                              stepsize=1 / 64, timestep=0.0, ensemble=0)
    stats = render_protocol.run(args)
    assert stats["num_cameras"] == 4 and stats["ms_mean"] > 0 and stats["num_parameters"] > 3000
    assert sorted(os.listdir(out)) == ["frame000.png", "frame001.png", "frame002.png", "frame003.png", "stats.json"]
    assert json.load(open(os.path.join(out, "stats.json")))["width"] == 64


# ------------------------------------------------------------------------------------------------ grid volumes
def _test_volume(shape=(20, 16, 12)):
    x, y, z = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
    return np.clip(np.exp(-3 * (x * x + y * y + z * z)) + 0.1 * np.sin(7 * x) * np.cos(5 * y), 0, 1).astype(np.float32)


def capi_volume_data(path, feature=0):
    from fvsrn_amd import capi
    return capi.Volume.load(path, feature).data()


def test_volume_container_round_trip(tmp_path):
    """pyrenderer.Volume (renderer/volume.cpp:1244-1400): features from tensors, .cvol save / load (uncompressed)."""
    data = _test_volume()
    v = pr.Volume()
    v.worldX, v.worldY, v.worldZ = 1.0, 0.8, 0.6
    f = v.add_feature_from_tensor("density", torch.from_numpy(data)[None])
    assert f.name() == "density" and f.channels() == 1 and f.base_resolution() == data.shape and f.type() == pr.Volume.DataType.TypeFloat
    v.add_feature_from_tensor("velocity", torch.zeros(3, *data.shape))
    path = str(tmp_path / "vol.cvol")
    v.save(path)
    w = pr.Volume(path)
    assert w.num_features() == 2 and (w.worldX, w.worldY, w.worldZ) == pytest.approx((1.0, 0.8, 0.6))
    assert w.get_feature(0).name() == "density" and w.get_feature("velocity").channels() == 3 and w.get_feature("nope") is None
    # Volume::save(filename, compression) (volume.cpp:623-682): LZ4 messages, both features of one file; read back bit for bit
    packed = str(tmp_path / "vol_lz4.cvol")
    v.save(packed, 5)
    assert os.path.getsize(packed) < os.path.getsize(path) // 2  # (a zero velocity field and a smooth density)
    u = pr.Volume(packed)
    assert u.num_features() == 2 and u.get_feature(1).channels() == 3 and u.get_feature(0).base_resolution() == data.shape
    got = capi_volume_data(packed)
    assert np.array_equal(got, data)
    with pytest.raises(RuntimeError, match="compression"):
        v.save(path, 10)
    # the same file through the C ABI loader
    from fvsrn_amd import capi
    res, bmin, bsize = capi.Volume.load(path).info()
    assert res == data.shape and np.allclose(bsize, (1.0, 0.8, 0.6))
    g = pr.VolumeInterpolationGrid()
    assert g.source() == pr.VolumeInterpolationGrid.Empty and g.interpolation() == pr.VolumeInterpolationGrid.Trilinear
    g.setSource(w)
    assert g.source() == pr.VolumeInterpolationGrid.Volume and g.box_min().x == pytest.approx(-0.5) and g.box_size().z == pytest.approx(0.6)
    g.setSource(torch.from_numpy(data)[None])
    assert g.source() == pr.VolumeInterpolationGrid.TorchTensor and g.maxDensity() == pytest.approx(float(data.max()))
    assert g.box_size().x == pytest.approx(1.0) and g.box_size().y == pytest.approx(0.8)  # res / max(res)


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["volume", "tensor"])
def test_grid_volume_renders_like_the_restatement(tmp_path, source):
    """ImageEvaluatorSimple with a VolumeInterpolationGrid (BASELINE.json configs[0]: grid DVR) against oracle_render_volume."""
    from oracle import oracle
    data = _test_volume()
    grid = pr.VolumeInterpolationGrid()
    if source == "volume":
        v = pr.Volume()
        v.worldX, v.worldY, v.worldZ = 1.0, 0.8, 0.6
        v.add_feature_from_tensor("density", torch.from_numpy(data)[None])
        grid.setSource(v)
        osrc = oracle.VOLUME_SOURCE_TEXTURE
    else:
        grid.setSource(torch.from_numpy(data)[None])
        osrc = oracle.VOLUME_SOURCE_TENSOR
    grid.setInterpolation(pr.VolumeInterpolationGrid.Tricubic)
    ev = pr.ImageEvaluatorSimple()
    ev.volume = grid
    ev.camera.orientation = pr.CameraOnASphere.Ym
    ev.camera.pitchYawDistance.value = pr.double3(0.5, 0.8, 1.7)
    ev.camera.fov_y_radians = float(np.deg2rad(45.0))
    tf = pr.TransferFunctionIdentity()
    tf.absorption_emission.value = pr.double2(30.0, 1.0)
    ev.ray_evaluator.tf = tf
    ev.ray_evaluator.stepsize = 1 / 96
    ev.ray_evaluator.min_density, ev.ray_evaluator.max_density = 0.1, 0.9
    W, H = 64, 48
    img = ev.render(W, H)[0].cpu().numpy()
    eye, right, up = oracle.camera_on_a_sphere("Ym", (0, 0, 0), 0.5, 0.8, 1.7)
    scene = oracle.OracleScene(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1 / 96, early_out=True,
                               tf_kind=oracle.TF_IDENTITY, tf_scale_absorption=30.0, tf_scale_emission=1.0, density_min=0.1, density_max=0.9)
    bmin, bsize = (-0.5, -0.4, -0.3), (1.0, 0.8, 0.6)
    ref, _ = oracle.OracleVolume(data, bmin, bsize, oracle.VOLUME_TRICUBIC, osrc).render(scene, W, H)
    assert ref[3].max() > 0.5
    assert np.abs(img[:4] - ref[:4]).max() < (3e-3 if source == "volume" else 2e-4)
    # IVolumeInterpolation::evaluate takes unit-box positions (the reference sets the box to [0,1]^3, volume_interpolation.cpp:46-49)
    pos = torch.rand(1000, 3, device="cuda")
    world = np.array(bmin, np.float32) + pos.cpu().numpy() * np.array(bsize, np.float32)
    ov = oracle.OracleVolume(data, bmin, bsize, oracle.VOLUME_TRICUBIC, osrc)
    got = grid.evaluate(pos).cpu().numpy()[:, 0]
    assert np.abs(got - ov.evaluate(world)).max() < 2e-5
    # evaluate_with_gradients: evalNormal of the grid = central differences one voxel to either side (renderer_volume_grid.cuh:234-283)
    dens, grad = grid.evaluate_with_gradients(pos)
    assert dens.shape == (1000, 1) and grad.shape == (1000, 3)
    for k in range(3):
        h = 1.0 / (data.shape[k] - 1)
        off = np.zeros(3, np.float32)
        off[k] = h * bsize[k]
        want = (ov.evaluate(world + off) - ov.evaluate(world - off)) * (0.5 / h)
        assert np.abs(grad[:, k].cpu().numpy() - want).max() < 2e-3 * max(1.0, np.abs(want).max())
    # refine (image_evaluator_simple.cpp:351-356): the DVR is deterministic, the running average equals the render
    img2 = ev.refine(W, H, ev.render(W, H))
    img3 = ev.refine(W, H, img2)
    assert float((img3[0, :4].cpu() - torch.from_numpy(img[:4])).abs().max()) < 1e-6
    assert ev.get_module_for_tag("volume") is grid and ev.get_module_for_tag("tf") is tf
    with pytest.raises(RuntimeError, match="no module"):
        ev.get_module_for_tag("nope")


def test_convert_to_texture_tf(tmp_path):
    """RayEvaluationSteppingDvr.convert_to_texture_tf (ray_evaluation_stepping.cpp:767-779, what inference.py:335 calls before it
    times a network): the TF sampled at the 256 texel centres becomes a TransferFunctionTexture."""
    from oracle import oracle
    p = tmp_path / "pw.json"
    p.write_text(json.dumps(_scene_json("Piecewise", PIECEWISE_JSON)))
    ev = pr.load_from_json(str(p))
    table = ev.ray_evaluator.tf.tensor.numpy()[0]  # (R,5): r,g,b,absorption,pos
    ev.ray_evaluator.convert_to_texture_tf()
    assert isinstance(ev.ray_evaluator.tf, pr.TransferFunctionTexture)
    tex = ev.ray_evaluator.tf.tensor.numpy()[0]
    assert tex.shape == (256, 4)
    d = ((np.arange(256) + 0.5) / 256).astype(np.float32)
    want = np.zeros((256, 4), np.float32)
    for i, x in enumerate(d):  # sampleTF, renderer_tf_piecewise.cuh:31-52
        k = 0
        while k < table.shape[0] - 2 and not table[k + 1, 4] > x:
            k += 1
        a, b = table[k], table[k + 1]
        f = (min(max(x, a[4]), b[4]) - a[4]) / (b[4] - a[4])
        want[i] = a[:4] + f * (b[:4] - a[:4])
    assert np.abs(tex[:, :3] - want[:, :3]).max() < 1e-6
    assert np.abs(tex[:, 3] - want[:, 3]).max() < 1e-5 * max(1.0, want[:, 3].max())
    before = ev.ray_evaluator.tf
    ev.ray_evaluator.convert_to_texture_tf()  # already a texture TF: unchanged
    assert ev.ray_evaluator.tf is before
    # Identity and Gaussian sources
    re = pr.RayEvaluationSteppingDvr()
    re.tf.absorption_emission.value = pr.double2(25.0, 0.5)
    re.convert_to_texture_tf()
    t = re.tf.tensor.numpy()[0]
    assert np.allclose(t[:, 0], d * 0.5, atol=1e-6) and np.allclose(t[:, 3], d * 25.0, rtol=1e-5)
    p = tmp_path / "g.json"
    gaussian_json = {"absorptionScaling": 2.0, "points": [[1, 0, 0, 10, 0.3, 0.1], [0, 1, 0, 20, 0.7, 0.05]]}
    p.write_text(json.dumps(_scene_json("Gaussian", gaussian_json)))
    ev = pr.load_from_json(str(p))
    g = ev.ray_evaluator.tf.tensor.numpy()[0]
    ev.ray_evaluator.convert_to_texture_tf()
    t = ev.ray_evaluator.tf.tensor.numpy()[0]
    want = sum(np.outer(np.exp(-(d - r[4]) ** 2 / r[5] ** 2), r[:4]) for r in g)
    assert np.abs(t - want).max() < 2e-5 * max(1.0, want.max())


@pytest.mark.gpu
def test_render_with_converted_texture_tf_stays_close(tmp_path):
    """inference.py:335: the renders with the piecewise TF and with its 256-texel texture agree to the resolution of the texture."""
    p = tmp_path / "pw.json"
    p.write_text(json.dumps(_scene_json("Piecewise", PIECEWISE_JSON)))
    vn = util.random_network(C=32, layers=4, activation="ReLU", output_mode="density", seed=8, box_min=(-0.5, -0.5, -0.5))
    path = str(tmp_path / "net.volnet")
    open(path, "wb").write(volnet_io.save_volnet(vn))
    ev = pr.load_from_json(str(p))
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(pr.SceneNetwork.load(path))
    ev.volume = vol
    a = ev.render(64, 48).clone()
    ev.ray_evaluator.convert_to_texture_tf()
    b = ev.render(64, 48)
    assert float(a[0, 3].max()) > 0.2
    assert float((a[0, :4] - b[0, :4]).abs().max()) < 2e-2


@pytest.mark.gpu
def test_network_evaluate_with_gradients_and_ray_multisampling(tmp_path):
    """IVolumeInterpolation.evaluate_with_gradients for a network in FINITE_DIFFERENCES mode (central differences of evaluate, unit box)
    and ICamera.generate_rays_multisampling (jittered rays, batch = sample)."""
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="density", seed=3, box_min=(-0.5, -0.5, -0.5))
    path = str(tmp_path / "net.volnet")
    open(path, "wb").write(volnet_io.save_volnet(vn))
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(pr.SceneNetwork.load(path))
    pos = torch.rand(512, 3, device="cuda") * 0.8 + 0.1
    with pytest.raises(RuntimeError, match="gradient"):
        vol.evaluate_with_gradients(pos)  # OFF_OR_DIRECT: a density network predicts no gradient
    vol.gradient_mode = pr.VolumeInterpolationNetwork.GradientMode.FINITE_DIFFERENCES
    vol.finite_differences_stepsize = 1 / 64
    dens, grad = vol.evaluate_with_gradients(pos)
    assert torch.equal(dens, vol.evaluate(pos))
    # fp16 positions: the scalar-type dispatch takes fvsrn_evaluate_points_half (fp16 in, fp16 out, no conversion passes) -- the fp32 result, rounded once
    out16 = vol.evaluate(pos.half())
    assert out16.dtype == torch.float16 and torch.equal(out16, vol.evaluate(pos.half().float()).half())
    h = 1 / 64
    for k in range(3):
        off = torch.zeros(1, 3, device="cuda")
        off[0, k] = h
        want = (vol.evaluate(pos + off) - vol.evaluate(pos - off))[:, 0] / (2 * h)
        assert float((grad[:, k] - want).abs().max()) < 1e-5
    assert float(grad.abs().max()) > 1e-3
    layer = vol.current_network().get_layer(0)
    assert layer.valid(False) and vol.object_resolution().x == 256 and vol.voxel_size().x == pytest.approx(1 / 255)
    cam = pr.CameraOnASphere()
    cam.pitchYawDistance.value = pr.double3(0.4, 0.7, 1.6)
    start, direction = cam.generate_rays_multisampling(24, 16, 5)
    c_start, c_dir = cam.generate_rays(24, 16)
    assert start.shape == (5, 16, 24, 3) and direction.shape == (5, 16, 24, 3)
    assert float((start - c_start).abs().max()) < 1e-6
    assert float((direction.norm(dim=3) - 1).abs().max()) < 1e-5
    # jitter stays inside the pixel: directions differ from the pixel-centre ray by less than one pixel's angle, and differ between samples
    assert float((direction - c_dir).abs().max()) < 2 * float((c_dir[0, 0, 1] - c_dir[0, 0, 0]).abs().max())
    assert float((direction[0] - direction[1]).abs().max()) > 0
    again, _ = cam.generate_rays_multisampling(24, 16, 5), None
    assert torch.equal(again[1], direction)  # seeded like the reference: (42, time)


@pytest.mark.gpu
@pytest.mark.parametrize("mode,C,layers,grid", [("densitycurvature", 32, 4, None), ("densitycurvature:direct", 64, 3, (16, 8))])
def test_evaluate_with_gradients_and_curvature_of_a_curvature_predicting_network(tmp_path, mode, C, layers, grid):
    """IVolumeInterpolation.evaluate_with_gradients_and_curvature (volume_interpolation.cpp:245-360) on a densitycurvature network:
    eval + evalNormal + evalCurvature, all three from the network's six outputs -- against oracle_eval_points_full; any other
    network raises (the reference traps)."""
    from oracle import oracle
    vn = util.random_network(C=C, layers=layers, activation="SnakeAlt", output_mode=mode, seed=6, fourier_std=0.4, grid=grid)
    path = str(tmp_path / "net.volnet")
    open(path, "wb").write(volnet_io.save_volnet(vn))
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(pr.SceneNetwork.load(path))
    pos = torch.rand(1000, 3, device="cuda")
    dens, grad, curv = vol.evaluate_with_gradients_and_curvature(pos)
    full = oracle.OracleNetwork(vn, oracle.ACC_FLOAT).evaluate_full(pos.cpu().numpy())  # value[4], normal[3], curvature[2]
    assert dens.shape == (1000, 1) and grad.shape == (1000, 3) and curv.shape == (1000, 2)
    assert np.abs(dens.cpu().numpy()[:, 0] - full[:, 0]).max() < 2e-3
    assert np.abs(grad.cpu().numpy() - full[:, 4:7]).max() < 2e-3 * max(1.0, np.abs(full[:, 4:7]).max())
    assert np.abs(curv.cpu().numpy() - full[:, 7:9]).max() < 2e-3 * max(1.0, np.abs(full[:, 7:9]).max())
    assert np.abs(full[:, 7:9]).max() > 1e-3 and np.abs(full[:, 7] - full[:, 8]).max() > 1e-3
    # same through the ctypes binding, and the other entry points of this network are unchanged
    from fvsrn_amd import capi
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    d2, g2, c2 = net.evaluate_with_gradients_and_curvature(pos)
    assert torch.equal(c2, curv) and torch.equal(g2, grad)
    assert float((net.evaluate(pos) - d2).abs().max()) < 5e-4
    vn2 = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="densitygrad", seed=6, fourier_std=0.4)
    open(path, "wb").write(volnet_io.save_volnet(vn2))
    vol.set_network(pr.SceneNetwork.load(path))
    with pytest.raises(RuntimeError, match="curvature"):
        vol.evaluate_with_gradients_and_curvature(pos)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["densitygrad", "densitygrad:cubic"])
def test_evaluate_with_gradients_of_a_gradient_predicting_network(tmp_path, mode):
    """GRADIENT_MODE_OFF_OR_DIRECT on a densitygrad network: evaluate_with_gradients returns the network's own gradient outputs
    (evalNormal, renderer_volume_tensorcores.cuh:1166-1183; cubed in densitygrad:cubic) -- against oracle_eval_points_full."""
    from oracle import oracle
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode=mode, seed=5, fourier_std=0.4)
    path = str(tmp_path / "net.volnet")
    open(path, "wb").write(volnet_io.save_volnet(vn))
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(pr.SceneNetwork.load(path))
    pos = torch.rand(777, 3, device="cuda")
    dens, grad = vol.evaluate_with_gradients(pos)
    full = oracle.OracleNetwork(vn, oracle.ACC_FLOAT).evaluate_full(pos.cpu().numpy())  # value[4], normal[3], curvature[2]
    assert dens.shape == (777, 1) and grad.shape == (777, 3)
    assert np.abs(dens.cpu().numpy()[:, 0] - full[:, 0]).max() < 2e-3
    assert np.abs(grad.cpu().numpy() - full[:, 4:7]).max() < 2e-3 * max(1.0, np.abs(full[:, 4:7]).max())
    assert np.abs(full[:, 4:7]).max() > 1e-3


# ---- the reference's own scene files (tests/golden/scenes: applications/config-files/*.json trimmed by tests/golden/make_scene_fixtures.py) ----
REFERENCE_SCENE_FIXTURES = sorted(os.path.basename(p)[:-5] for p in __import__("glob").glob(os.path.join(util.GOLDEN_DIR, "scenes", "*.json")))


def oracle_scene_from_reference_json(d, oracle):
    """The test's OWN reading of a reference scene file (independent of pyrenderer.load_from_json): camera (camera.cpp:349-362), DVR ray evaluator
    (ray_evaluation_stepping.cpp:62-72,740-755), transfer function, BRDF (brdf.cpp:208-225), blending -> OracleScene keyword arguments."""
    cam = d["camera"]["Sphere"]
    eye, right, up = oracle.camera_on_a_sphere(cam["orientation"], cam.get("center", (0, 0, 0)), cam["pitch"], cam["yaw"], cam["distance"])
    dvr = d["RayEvaluation"]["DVR"]
    step = dvr["stepsize"] / 256.0 if dvr.get("stepsizeIsObjectSpace", False) else dvr["stepsize"]
    kw = dict(eye=eye, right=right, up=up, fov_y_radians=cam["fovY"], stepsize=step, density_min=dvr.get("minDensity", 0.0), density_max=dvr.get("maxDensity", 1.0),
              early_out=dvr.get("earlyOut", True),
              blend_mode=oracle.BLEND_ALPHA if d.get("blending", {}).get("blending", {}).get("blending", "BeerLambert") == "Alpha" else oracle.BLEND_BEER_LAMBERT)
    name = dvr["selectedTF"]
    t = d["tf"][name]
    if name == "Piecewise":
        kw.update(tf_kind=oracle.TF_PIECEWISE, tf_table=oracle.tf_piecewise_table(t["colorPoints"], t["opacityPoints"], t["absorptionScaling"]))
    elif name == "Texture":
        kw.update(tf_kind=oracle.TF_TEXTURE, tf_table=oracle.tf_texture_table(t["colorPoints"], t["opacityPoints"], t["absorptionScaling"]),
                  tf_preintegration={"None": 0, "Preintegrate1D": 1, "Preintegrate2D": 2}[t.get("preintegrationMode", "None")])
    elif name == "Gaussian":
        rows = [[p[0], p[1], p[2], p[3] * t["absorptionScaling"], p[4], p[5]] for p in t["points"]]
        kw.update(tf_kind=oracle.TF_GAUSSIAN, tf_table=np.asarray(rows, np.float32),
                  tf_gaussian_mode=2 if t.get("usePiecewiseAnalyticIntegration") else (1 if t.get("scaleWithGradient") else 0))
    else:
        kw.update(tf_kind=oracle.TF_IDENTITY, tf_scale_absorption=t["absorptionScaling"], tf_scale_emission=t["emissionScaling"])
    b = d.get("brdf", {}).get("Lambert", {})
    shaded = bool(b.get("enablePhong") or b.get("enableMagnitudeScaling"))
    if shaded:
        point = b.get("lightType") == "Point"
        if b.get("lightFollowsCamera", True):  # updateLightFromCamera, brdf.cpp:490-508: the camera's position / viewing direction
            light = tuple(float(v) for v in (eye if point else np.cross(up, right)))
        else:
            light = tuple(b["lightPosition"] if point else b["lightDirection"])
        kw.update(brdf=dict(enable_phong=b.get("enablePhong", False), enable_magnitude_scaling=b.get("enableMagnitudeScaling", False),
                            magnitude_scaling=b.get("magnitudeScaling", 1.0), ambient=b.get("ambient", 1.0), specular=b.get("specular", 1.0),
                            magnitude_center=b.get("magnitudeCenter", 1.0), magnitude_radius=b.get("magnitudeRadius", 1.0),
                            specular_exponent=b.get("specularExponent", 1), light_type=0 if point else 1, light=light))
    return kw, shaded


@pytest.mark.gpu
@pytest.mark.parametrize("scene", REFERENCE_SCENE_FIXTURES)
def test_reference_scene_files_render_like_the_oracle(scene):
    """The reference's scene files rendered, not just loaded: pyrenderer.load_from_json on the file, the ground-truth volume
    replaced by a network like the reference's scripts do (inference.py:598; here the golden fixture g1_c32l4_snakealt_density, reference
    weights), 48 x 48 -- against the oracle set up by this test's own reading of the same JSON.  One file per transfer-function kind
    (Piecewise, Texture, Gaussian), both step-size conventions, two shaded ones (Phong; Phong + magnitude scaling + 2D pre-integrated
    texture TF) with finite-difference normals."""
    from oracle import oracle
    path = os.path.join(util.GOLDEN_DIR, "scenes", scene + ".json")
    d = json.load(open(path))
    kw, shaded = oracle_scene_from_reference_json(d, oracle)
    g, meta = util.load_golden("g1_c32l4_snakealt_density")
    vn = util.golden_to_volnet(g, meta, box_min=(-0.5, -0.5, -0.5), box_size=(1.0, 1.0, 1.0))  # the box of the scenes' unit-sized volumes
    if oracle.OracleScene(**kw).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT), 24, 24)[0][3].max() < 0.05:
        # the randomly initialised golden networks put out densities of 0.45 .. 0.57; the ejecta transfer functions are opaque at 0.26 .. 0.43 only:
        # a seeded network with a 12 x larger last layer and a bias of -0.75 (densities 0.34 .. 0.59) stands in there
        from fvsrn_amd import synthetic
        a = synthetic.random_arrays(C=32, layers=4, output_mode="density", fourier_std=0.35, seed=77)
        a["weights"][-1] = a["weights"][-1] * 12.0
        a["biases"][-1] = a["biases"][-1] * 0 - 0.75
        vn = volnet_io.build_volnet(fourier_B=a["B"], weights=a["weights"], biases=a["biases"], activation="SnakeAlt", activation_param=1.0, output_mode="density",
                                    box_min=(-0.5, -0.5, -0.5), box_size=(1, 1, 1), time_grids=None, grid_encoding=volnet_io.ENC_FLOAT)
    volnet = os.path.join(os.environ.get("TMPDIR", "/tmp"), "scene_fixture_%d.volnet" % os.getpid())
    open(volnet, "wb").write(volnet_io.save_volnet(vn))
    ev = pr.load_from_json(path)
    vol = pr.VolumeInterpolationNetwork()
    vol.set_network(pr.SceneNetwork.load(volnet))
    if shaded:
        vol.gradient_mode = pr.VolumeInterpolationNetwork.GradientMode.FINITE_DIFFERENCES
        vol.finite_differences_stepsize = 1 / 64
        kw.update(gradient_mode=1, finite_differences_stepsize=1 / 64)
    ev.volume = vol
    img = ev.render(48, 48).cpu().numpy()[0]
    ref, _ = oracle.OracleScene(**kw).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT), 48, 48)
    assert ref[3].max() > 0.05, "the scene is empty with this network: the comparison would be vacuous"
    tol = 1.2e-2 if shaded else 3e-3  # (the tolerances of the shaded / unshaded module tests above)
    # a Gaussian TF with sigma 0.016 (ejecta70) turns 1e-4 of density into 1e-2 of opacity: there the bar is the distance between the reference's
    # own two arithmetic models (fp32 / fp16 accumulation) on this image, like the randomised parity suite of r02
    ref_h, _ = oracle.OracleScene(**kw).render(oracle.OracleNetwork(vn, oracle.ACC_HALF), 48, 48)
    tol = max(tol, float(np.abs(ref_h[:7] - ref[:7]).max()))
    assert np.abs(img[:7] - ref[:7]).max() < tol, (scene, float(np.abs(img[:7] - ref[:7]).max()), tol)
    solid = ref[3] > 1e-3
    assert np.array_equal(np.isnan(img[7])[solid], np.isnan(ref[7])[solid])


@pytest.mark.skipif(not os.path.isdir(REFERENCE_SCENES), reason="reference checkout not present (it is not on the GPU box)")
def test_scene_fixtures_are_the_reference_files_and_its_own_volume_loads():
    """tests/golden/scenes/*.json reproduce from the reference's files (make_scene_fixtures.py), and the one scene whose ground-truth volume the
    snapshot holds -- RichtmyerMeshkov-t20-v1-dvr.json -> volumes/RichtmyerMeshkov/ppm-t0020.cvol, old format, LZ4 -- loads WITH that volume."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_scene_fixtures", os.path.join(util.GOLDEN_DIR, "make_scene_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    assert sorted(m.SCENES) == REFERENCE_SCENE_FIXTURES
    for name in m.SCENES:
        assert m.trim(json.load(open(os.path.join(REFERENCE_SCENES, name + ".json")))) == json.load(open(os.path.join(util.GOLDEN_DIR, "scenes", name + ".json"))), name
    ev = pr.load_from_json(os.path.join(REFERENCE_SCENES, "RichtmyerMeshkov-t20-v1-dvr.json"))
    f = ev.volume.volume().get_feature(0)
    assert f.name() == "density" and tuple(f.base_resolution()) == (256, 256, 256) and f.type() == pr.Volume.DataType.TypeUChar
    bmin, bsize = ev.volume.box_min(), ev.volume.box_size()
    assert (bmin.x, bmin.y, bmin.z) == (-0.5, -0.5, -0.5) and (bsize.x, bsize.y, bsize.z) == (1.0, 1.0, 1.0)
