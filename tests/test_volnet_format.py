"""
.volnet container and host model (no GPU): the C++ reader/writer behind the C ABI against the independent
pure-Python implementation, mirroring the round-trip assertions of the reference's unit test
(unittests/testSRN.cpp:413-430) and the layer re-layout rules of SceneNetwork::addLayer.
"""
import struct

import numpy as np
import pytest

import util
from fvsrn_amd import capi, volnet_io


def build_via_c_abi(d, meta, encoding=0):
    """The export_to_pyrenderer call sequence (reference network.py:798-897) through the C ABI."""
    net = capi.Network.create()
    net.set_input(d["B"], has_time=meta.get("use_time_direct", False), has_direction=meta.get("use_direction", False), premultiplied=True)
    net.set_output_mode(meta["output_mode"])
    net.set_box((0, 0, 0), (1, 1, 1))
    if "grid" in d:
        net.set_latent_grid_layout(0, 1, 1, 0, 0)
        net.set_latent_grid(False, 0, d["grid"], encoding)
    elif "grid_time" in d or "grid_ensemble" in d:
        tg = d.get("grid_time", np.zeros((0,)))
        eg = d.get("grid_ensemble", np.zeros((0,)))
        net.set_latent_grid_layout(0, len(tg), 1, 0, len(eg))
        for i, g in enumerate(tg):
            net.set_latent_grid(False, i, g, encoding)
        for i, g in enumerate(eg):
            net.set_latent_grid(True, i, g, encoding)
    n = len(meta["layers"].split(":")) + 1
    for i in range(n):
        last = i == n - 1
        net.add_layer(d["W%d" % i], d["b%d" % i], "None" if last else meta["activation"], 1.0 if last else meta["activation_param"])
    return net


@pytest.mark.parametrize("name", util.golden_names("g"))
def test_c_abi_builder_writes_the_same_bytes_as_the_python_writer(name):
    d, meta = util.load_golden(name)
    net = build_via_c_abi(d, meta)
    assert net.valid(), net.last_error()
    py_bytes = volnet_io.save_volnet(util.golden_to_volnet(d, meta))
    assert net.save() == py_bytes


@pytest.mark.parametrize("name", ["g1_c32l4_snakealt_rgbo", "g1_c64l6_grid16r8_snakealt_density-direct",
                                  "g2_time3_ens2_c32l4_grid32r8", "g2_time3_passtime_c32l4_grid16r8"])
def test_save_load_round_trip_keeps_every_field(name):
    d, meta = util.load_golden(name)
    vn = util.golden_to_volnet(d, meta)
    data = volnet_io.save_volnet(vn)
    net = capi.Network.from_volnet(data)
    assert net.valid()
    assert net.save() == data                      # C++ load -> save is the identity
    back = volnet_io.load_volnet(net.save())        # Python load of C++ bytes
    assert back.num_fourier == vn.num_fourier and back.has_time == vn.has_time and back.output_mode == vn.output_mode
    assert np.array_equal(back.fourier, vn.fourier)
    assert np.array_equal(net.fourier(), vn.fourier)
    assert len(back.layers) == len(vn.layers)
    for i, (a, b) in enumerate(zip(back.layers, vn.layers)):
        co, ci, act, p, w, bias = net.layer(i)
        assert (co, ci) == (b.channels_out, b.channels_in) and volnet_io.ACTIVATIONS[act] == b.activation and p == b.activation_param
        assert np.array_equal(w, b.weights) and np.array_equal(bias, b.bias)
        assert np.array_equal(a.weights, b.weights) and np.array_equal(a.bias, b.bias)
    info = net.info()
    assert info.num_layers == len(vn.layers) and info.hidden_channels == vn.layers[1].channels_in
    assert list(info.box_min) == list(vn.box_min) and list(info.box_size) == list(vn.box_size)
    if vn.has_grid():
        assert info.time_num == len(vn.time_grids or []) and info.ensemble_num == len(vn.ensemble_grids or [])
        for a, b in zip((back.time_grids or []) + (back.ensemble_grids or []), (vn.time_grids or []) + (vn.ensemble_grids or [])):
            assert a.encoding == b.encoding and np.array_equal(a.data, b.data)


def test_first_layer_padding_and_last_layer_transpose():
    """addLayer: zero column inserted at input 3; layers with <16 outputs are stored [in][out] (:806-894)."""
    d, meta = util.load_golden("g1_c32l4_snakealt_rgbo")
    net = build_via_c_abi(d, meta)
    co, ci, _, _, w, _ = net.layer(0)
    assert (co, ci) == (32, 32)
    w = w.reshape(32, 32)
    expect = d["W0"].astype(np.float16).view(np.uint16)
    assert np.all(w[:, 3] == 0) and np.array_equal(w[:, :3], expect[:, :3]) and np.array_equal(w[:, 4:], expect[:, 3:])
    co, ci, act, _, w, _ = net.layer(3)
    assert (co, ci, act) == (4, 32, capi.ACTIVATIONS["None"])
    assert np.array_equal(w.reshape(32, 4), d["W3"].astype(np.float16).view(np.uint16).T)


def test_flops_parameters_and_max_warps_match_the_reference_formulas():
    d, meta = util.load_golden("g1_c64l6_grid16r8_snakealt_density-direct")
    info = build_via_c_abi(d, meta).info()
    assert info.flops_per_sample == 43188
    d, meta = util.load_golden("g1_c32l4_snakealt_density")
    net = build_via_c_abi(d, meta)
    info = net.info()
    assert info.flops_per_sample == 6228
    assert info.num_parameters == 14 * 3 + 32 * 32 * 3 + 32 * 3 + 32 + 1
    # computeMaxWarps (:987-1041): (48K - shared weights) / (C * 2 B * 32 lanes)
    shared = (3 * (32 * 32 + 32)) * 2
    assert info.max_warps_mixed == (48 * 1024 - shared) // (32 * 2 * 32)
    assert info.max_warps_shared == (48 * 1024 - shared - (42 + 33) * 2) // (32 * 2 * 32)


@pytest.mark.parametrize("enc", [volnet_io.ENC_BYTE_LINEAR, volnet_io.ENC_BYTE_GAUSSIAN])
def test_byte_grid_encodings_match_python_writer(enc):
    d, meta = util.load_golden("g1_c32l4_grid16r8_relu_density")
    net = build_via_c_abi(d, meta, encoding=enc)
    vn = util.golden_to_volnet(d, meta, encoding=enc)
    back = volnet_io.load_volnet(net.save())
    g_c, g_py = back.time_grids[0], vn.time_grids[0]
    # quantised bytes may differ by one step where the fp32 pre-image sits on a rounding boundary
    diff = np.abs(g_c.data.astype(np.int32) - g_py.data.astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    assert np.allclose(g_c.offset, g_py.offset, rtol=1e-6, atol=1e-7) and np.allclose(g_c.scale, g_py.scale, rtol=1e-6, atol=1e-7)


def test_invalid_inputs_are_rejected_with_messages():
    d, meta = util.load_golden("g1_c32l4_snakealt_density")
    data = volnet_io.save_volnet(util.golden_to_volnet(d, meta))
    for cut in (0, 3, 10, len(data) // 2, len(data) - 1):
        with pytest.raises(capi.FvsrnError) as e:
            capi.Network.from_volnet(data[:cut] if cut else b"\x00")
        assert e.value.code == -2
    bad_version = struct.pack("<i", 7) + data[4:]
    with pytest.raises(capi.FvsrnError, match="Unknown version"):
        capi.Network.from_volnet(bad_version)
    # hidden layers of different widths: valid() passes, the kernel configuration rejects (getDefines :1177-1179)
    net = capi.Network.create()
    net.set_input(d["B"])
    net.set_output_mode("density")
    rng = np.random.RandomState(0)
    net.add_layer(rng.randn(32, 31), rng.randn(32), "ReLU")
    net.add_layer(rng.randn(48, 32), rng.randn(48), "ReLU")
    net.add_layer(rng.randn(1, 48), rng.randn(1), "None")
    assert net.valid()
    with pytest.raises(capi.FvsrnError, match="same size"):
        net.kernel_name()
    # wrong input width -> valid() is false with the reference's message
    net = capi.Network.create()
    net.set_input(d["B"])
    net.add_layer(rng.randn(32, 30), rng.randn(32), "ReLU")
    net.add_layer(rng.randn(1, 32), rng.randn(1), "None")
    assert not net.valid() and "Invalid input channels" in net.last_error()
    # a latent grid without Fourier features
    net = capi.Network.create()
    net.set_input(None)
    net.set_latent_grid_layout(0, 1, 1, 0, 0)
    net.set_latent_grid(False, 0, np.zeros((16, 4, 4, 4), np.float32), 0)
    assert not net.valid() and "fourier" in net.last_error().lower()


def test_unsupported_variants_fail_loudly():
    rng = np.random.RandomState(0)
    # a hidden width outside the ahead-of-time matrix (16 .. 128 in steps of 16: what the reference's 48 KiB of shared memory can hold)
    net = capi.Network.create()
    net.set_input(rng.randn(70, 3).astype(np.float32))
    net.add_layer(rng.randn(144, 143), rng.randn(144), "ReLU")
    net.add_layer(rng.randn(1, 144), rng.randn(1), "None")
    assert net.valid()
    with pytest.raises(capi.FvsrnError) as e:
        net.kernel_name()
    assert e.value.code == -4 and "not in the compiled variant set" in str(e.value)
    # Sigmoid hidden activations are part of the variant set (Layer::Activation::Sigmoid)
    d, meta = util.load_golden("g1_c32l4_snakealt_density")
    net = capi.Network.create()
    net.set_input(d["B"])
    net.add_layer(rng.randn(32, 31), rng.randn(32), "Sigmoid")
    net.add_layer(rng.randn(1, 32), rng.randn(1), "None")
    assert net.valid() and "evaluate_kernel<2,ACT_SIGMOID" in net.kernel_name(False)
    # r04: 16, 80 and 112 channels are compiled in as well
    for C in (16, 80, 112):
        net = capi.Network.create()
        net.set_input(rng.randn((C - 4) // 2, 3).astype(np.float32))
        net.add_layer(rng.randn(C, C - 1), rng.randn(C), "ReLU")
        net.add_layer(rng.randn(1, C), rng.randn(1), "None")
        assert net.valid() and ("evaluate_kernel<%d," % (C // 16)) in net.kernel_name(False)


def test_curvature_and_no_fourier_networks_select_a_kernel():
    rng = np.random.RandomState(1)
    d, meta = util.load_golden("g1_c32l4_snakealt_density")
    net = capi.Network.create()
    net.set_input(d["B"])
    net.set_output_mode("densitycurvature")
    net.add_layer(rng.randn(32, 31), rng.randn(32), "ReLU")
    net.add_layer(rng.randn(6, 32), rng.randn(6), "None")
    assert net.valid() and "render_kernel<2," in net.kernel_name(True)
    net = capi.Network.create()
    net.set_input(None)
    net.add_layer(rng.randn(64, 3), rng.randn(64), "Sine")
    net.add_layer(rng.randn(1, 64), rng.randn(1), "None")  # first + last layer only
    assert net.valid() and "render_kernel<4," in net.kernel_name(True)


def test_camera_on_a_sphere_all_orientations():
    from oracle import oracle
    for o in capi.ORIENTATIONS:
        for pitch, yaw, dist in [(0.4, 0.7, 1.6), (-0.3, 2.5, 3.0), (0.0, 0.0, 1.0)]:
            a = capi.camera_on_a_sphere(o, (0.1, -0.2, 0.3), pitch, yaw, dist)
            b = oracle.camera_on_a_sphere(o, (0.1, -0.2, 0.3), pitch, yaw, dist)
            for x, y in zip(a, b):
                assert np.allclose(x, y, atol=1e-6)


def test_handle_options_replace_environment_switches():
    """include/fvsrn.h fvsrn_option: per-handle values, validated, readable back; no GPU involved."""
    from fvsrn_amd import capi
    net = capi.Network.from_volnet(volnet_io.save_volnet(util.random_network(seed=3)))
    assert net.get_option("small_kernel") == -1 and net.get_option("relu_clamp") == 1
    net.set_option("relu_clamp", 0)
    assert net.get_option("relu_clamp") == 0
    eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
    scene = capi.Scene(eye=eye, right=right, up=up, fov_y_radians=0.8, stepsize=0.01, tf_kind=capi.TF_IDENTITY)
    assert scene.get_option("fourier_resync") == 0 and scene.get_option("depth_segments") == 0
    scene.set_option("fourier_resync", 1).set_option("depth_segments", 4).set_option("persistent", 0)
    assert (scene.get_option("fourier_resync"), scene.get_option("depth_segments"), scene.get_option("persistent")) == (1, 4, 0)
    for name, bad in (("fourier_resync", 3), ("depth_segments", 65), ("persistent", 2), ("waves_per_block", 3), ("relu_clamp", 2)):
        with pytest.raises(capi.FvsrnError):
            scene.set_option(name, bad)
    with pytest.raises(capi.FvsrnError):
        capi._check(capi.lib().fvsrn_scene_set_option(scene._h, 99, 0))


def test_info_of_a_network_under_construction_does_not_crash():
    """fvsrn_network_get_info between set_latent_grid_layout and the grids (unset grid pointers): an answer, not a segfault."""
    from fvsrn_amd import capi
    net = capi.Network.create()
    net.set_latent_grid_layout(0, 3, 1, 0, 0)
    i = net.info()
    assert i.time_num == 3 and i.grid_channels == 0
    assert not net.valid()


def test_volnet_with_zero_time_step_is_rejected():
    """A latent grid block with timeStep == 0 would make interpolateTime compute 0/0: FormatError at load."""
    from fvsrn_amd import capi
    vn = util.random_network(grid=(16, 4), seed=2)
    data = bytearray(volnet_io.save_volnet(vn))
    good = capi.Network.from_volnet(bytes(data))
    assert good.valid()
    # LatentGridTimeAndEnsemble header: version, timeMin, timeNum, timeStep, ensembleMin, ensembleNum (int32 each); find it by its
    # values (1, 0, 1, 1, 0, 0) and zero the time step
    import struct
    pat = struct.pack("<6i", 1, 0, 1, 1, 0, 0)
    at = bytes(data).find(pat)
    assert at > 0
    data[at + 12:at + 16] = struct.pack("<i", 0)
    with pytest.raises(capi.FvsrnError, match="time step"):
        capi.Network.from_volnet(bytes(data))
