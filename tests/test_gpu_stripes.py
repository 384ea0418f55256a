"""
The frame pipeline of a rank (fv-srn_amd/tiles.py StripeRenderer) and the cross-stream rules of the C ABI it relies on
(include/fvsrn.h: a network may be used from several streams; two working grids) on a real MI355X:  pytest -m gpu
"""
import os

import numpy as np
import pytest

import util
from oracle import oracle
from test_gpu_parity import GAUSS_TF, TOL_IMG, TOL_SAME_MODEL, assert_images_close, make_scene_kwargs

pytestmark = pytest.mark.gpu


def _scene_kw(yaw, **kw):
    d = make_scene_kwargs(yaw=yaw, stepsize=1 / 96, early_out=False, tf_kind=oracle.TF_IDENTITY, tf_scale_absorption=15.0)
    d.update(kw)
    return d


@pytest.mark.parametrize("ahead", [False, True])
@pytest.mark.parametrize("enc,has_time,slots", [(0, False, 0), (0, True, 0), (2, False, 0), (0, False, 2), (1, True, 3)])
def test_two_frames_in_flight_at_different_times(enc, has_time, slots, ahead):
    """BASELINE.json configs[4] in small: a time-dependent network rendered by StripeRenderer with consecutive frames on two
    streams, every frame at another time (and another camera).  The library blends frame i + 1's working grid while frame i
    still renders from the other one, and hands the time input over as a kernel argument; every frame must equal, bit for bit,
    the same frame rendered alone on a second network handle with a device synchronisation around it.
    ahead: the pipeline is told the next frame's time (submit(next_time=)) and enqueues its blend on a side stream beside the current
    render (fvsrn_network_prepare)."""
    import torch
    from fvsrn_amd import capi, tiles, volnet_io
    KEYS = 5
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="density", grid=(16, 8), seed=52, box_min=(-0.5, -0.5, -0.5),
                             fourier_std=0.4, encoding=enc, time_grids=KEYS, has_time=has_time)
    blob = volnet_io.save_volnet(vn)
    net, serial = capi.Network.from_volnet(blob), capi.Network.from_volnet(blob)
    if slots:
        net.set_option("keyframe_slots", slots)
    assert net.get_option("working_grids") == 0  # automatic: two for a network with several key frames
    W, H = 256, 192
    pipe = tiles.StripeRenderer(net, W, H, _scene_kw(0.0), pipelined=True)
    assert pipe.pipelined and len(pipe.scenes) == 2
    times = [0.0, 0.75, 1.5, 3.9, 2.25, 2.25, 4.0, 0.1, 0.1, 3.3]
    yaws = [0.3 + 0.37 * i for i in range(len(times))]
    got = {}
    for i in range(0, len(times), 2):  # two frames in flight, then read both buffers back
        for j in (i, i + 1):
            pipe.submit(j, _scene_kw(yaws[j]), time=times[j], next_time=times[j + 1] if ahead and j + 1 < len(times) else None)
        pipe.finish()
        torch.cuda.synchronize()
        for j in (i, i + 1):
            got[j] = pipe.frame(j & 1).clone()
    assert (pipe.blend_stream is not None) == ahead
    ref_scene = capi.Scene(**_scene_kw(0.0))
    for j, (t, yaw) in enumerate(zip(times, yaws)):
        serial.set_time_and_ensemble(t, 0)
        ref_scene.update(**_scene_kw(yaw))
        ref = ref_scene.render(serial, W, H)
        torch.cuda.synchronize()
        d = (torch.nan_to_num(got[j], nan=-7.0) - torch.nan_to_num(ref, nan=-7.0)).abs()
        assert float(d.max()) == 0.0, "frame %d (time %g): %d values differ, max %.3g, per channel %s" % (
            j, t, int((d > 0).sum()), float(d.max()), [int((d[0, c] > 0).sum()) for c in range(8)])
        assert float(ref[0, 3].max()) > 0.2
    # ... and the values themselves against the oracle at one of the times
    want, _ = oracle.OracleScene(**_scene_kw(yaws[3])).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT, time=times[3]), W, H)
    assert_images_close(got[3].cpu().numpy()[0], want, TOL_IMG)


def test_one_working_grid_serialises_but_stays_correct():
    """working_grids = 1 (the r02 layout): the blend of the next frame waits for the frame that still reads the grid."""
    import torch
    from fvsrn_amd import capi, tiles, volnet_io
    vn = util.random_network(C=32, layers=4, activation="ReLU", output_mode="density:direct", grid=(16, 8), seed=53, box_min=(-0.5, -0.5, -0.5),
                             time_grids=4, grid_scale=0.5)
    blob = volnet_io.save_volnet(vn)
    one, two = capi.Network.from_volnet(blob), capi.Network.from_volnet(blob)
    one.set_option("working_grids", 1)
    frames = {}
    for name, net in (("one", one), ("two", two)):
        pipe = tiles.StripeRenderer(net, 192, 128, _scene_kw(0.0), pipelined=True)
        out = []
        for i in range(0, 6, 2):
            for j in (i, i + 1):
                pipe.submit(j, _scene_kw(0.5 * j), time=0.45 * j)
            pipe.finish()
            torch.cuda.synchronize()
            out += [pipe.frame(0).clone(), pipe.frame(1).clone()]
        frames[name] = out
    for a, b in zip(frames["one"], frames["two"]):
        assert torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))
    with pytest.raises(capi.FvsrnError):
        one.set_option("working_grids", 3)


def test_evaluate_from_two_streams_while_the_time_changes():
    """evaluate_points of one network from two streams, the time changing between the calls, no synchronisation by the caller:
    every result equals the oracle at the time that was current when the call was made."""
    import torch
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=64, layers=3, activation="Sine", output_mode="density", grid=(16, 8), seed=54, fourier_std=0.4, time_grids=4,
                             has_time=True)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    pos = torch.rand(200000, 3, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    times = [0.2, 2.7, 1.1, 1.1, 3.0, 0.6]
    outs = []
    for i, t in enumerate(times):
        net.set_time_and_ensemble(t, 0)
        with torch.cuda.stream(streams[i & 1]):
            outs.append(net.evaluate(pos))
    torch.cuda.synchronize()
    p = pos[:2000].cpu().numpy()
    for t, o in zip(times, outs):
        ref = oracle.OracleNetwork(vn, oracle.ACC_FLOAT, time=t).evaluate(p)
        assert np.abs(o[:2000].cpu().numpy() - ref).max() < TOL_SAME_MODEL, t


def test_curvature_of_a_network_that_takes_the_time_as_input():
    """the curvature weight image has to see the current time like the plain image does (the time is a kernel
    argument since r03, no image is patched).  densitycurvature network with has_time, time changed after first use."""
    import torch
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="densitycurvature", grid=(16, 8), seed=55, fourier_std=0.4,
                             time_grids=3, has_time=True)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    pos = torch.rand(1500, 3, device="cuda", generator=torch.Generator("cuda").manual_seed(6))
    first = None
    for t in (0.0, 1.6, 0.4):
        net.set_time_and_ensemble(t, 0)
        dens, grad, curv = net.evaluate_with_gradients_and_curvature(pos)
        full = oracle.OracleNetwork(vn, oracle.ACC_FLOAT, time=t).evaluate_full(pos.cpu().numpy())  # value[4], normal[3], curvature[2]
        assert np.abs(dens.cpu().numpy()[:, 0] - full[:, 0]).max() < 2e-3
        assert np.abs(grad.cpu().numpy() - full[:, 4:7]).max() < 2e-3 * max(1.0, np.abs(full[:, 4:7]).max())
        assert np.abs(curv.cpu().numpy() - full[:, 7:9]).max() < 2e-3 * max(1.0, np.abs(full[:, 7:9]).max()), t
        if first is None:
            first = curv.clone()
        elif t == 1.6:
            assert float((curv - first).abs().max()) > 1e-3  # the time input matters for this network
    net.clear_gpu_resources()  # releases every device buffer of the handle (also the curvature image); the next call rebuilds them
    again = net.evaluate_with_gradients_and_curvature(pos)[2]
    assert torch.equal(again, curv)


@pytest.mark.parametrize("grid", [None, (16, 8)])
def test_texture_tf_with_negative_opacity_texels(grid):
    """A Texture TF table is not validated: a negative opacity texel gives a negative blend weight unless the sample is skipped
    like the reference does (`if (color1.w > 0)`, renderer_ray_evaluation_stepping_dvr.cuh:137).  The straight-line Texture tail
    leaves that test out, so the host only selects it for tables without negative opacities."""
    import torch
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=32, layers=4, activation="ReLU", output_mode="density", seed=56, box_min=(-0.5, -0.5, -0.5), grid=grid,
                             grid_scale=0.3)
    tab = np.random.RandomState(8).uniform(0.0, 1.0, (48, 4)).astype(np.float32)
    # opacity +, -, -, +, ...: both signs inside any density interval of 0.06 (a moderate amplitude: the slope of this table turns an
    # fp16-level difference of the density into 500 times as much opacity)
    tab[:, 3] = 10.0 * np.cos(np.arange(48) * np.pi / 2 + np.pi / 4)
    kw = make_scene_kwargs(stepsize=1 / 96, early_out=True, tf_kind=oracle.TF_TEXTURE, tf_table=tab)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    for small in (-1, 0):
        scene = capi.Scene(**kw).set_option("small_kernel", small)
        img = scene.render(net, 96, 64)
        torch.cuda.synchronize()
        plan = scene.last_render_info()  # (the steep table amplifies: compare with the kernels' stated arithmetic, test_fuzz_parity.py)
        ref, _ = oracle.OracleScene(rotation_resync=plan["rotation_resync"], segments=plan["segments"], **kw).render(
            oracle.OracleNetwork(vn, oracle.ACC_DEVICE), 96, 64)
        assert_images_close(img.cpu().numpy()[0], ref, TOL_IMG)
    assert ref[3].min() >= 0.0 and ref[3].max() > 0.03


@pytest.mark.parametrize("C,act,enc", [(64, "ReLU", 0), (48, "SnakeAlt", 0), (64, "Sine", 2)])
def test_overlap_kernel_variant_renders_the_same_image(C, act, enc):
    """48 / 64-wide latent-grid networks on the gather path run render_kernel in the fragment-major layer order since r05 (no register spills; r03 - r04: a
    separate render_stripe_kernel on request, compared here with the pipelined order).  Scene option overlap_kernel = 1 takes that kernel also where
    the cell table would be; 0 / -1 leave the choice to the footprint rule: in a 96 x 64 image all three gather -- same kernel, same image -- inside the
    tolerance of the oracle."""
    import torch
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=C, layers=4, activation=act, output_mode="density", grid=(16, 8), seed=61, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4,
                             encoding=enc, grid_scale=0.3)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    kw = make_scene_kwargs(stepsize=1 / 64, early_out=True, tf_kind=oracle.TF_GAUSSIAN, tf_table=GAUSS_TF)
    W, H = 96, 64
    a = capi.Scene(**kw).set_option("overlap_kernel", 0)
    b = capi.Scene(**kw).set_option("overlap_kernel", 1)
    ia, ib = a.render(net, W, H)[0].cpu().numpy(), b.render(net, W, H)[0].cpu().numpy()
    assert a.last_render_info()["resident_kernel"] is False and b.last_render_info()["segments"] == a.last_render_info()["segments"]
    ref, _ = oracle.OracleScene(**kw).render(oracle.OracleNetwork(vn, oracle.ACC_FLOAT), W, H)
    assert_images_close(ia, ref, TOL_IMG)
    assert_images_close(ib, ref, TOL_IMG)
    assert np.abs(ia[:4] - ib[:4]).max() < 1e-3 and ref[3].max() > 0.2
    assert b.last_render_info()["overlap_kernel"] and a.last_render_info()["overlap_kernel"] and np.array_equal(np.nan_to_num(ia), np.nan_to_num(ib))
    auto = capi.Scene(**kw)  # (a 96 x 64 image: the footprint rule gathers; BYTE_GAUSSIAN has no cell table)
    ic = auto.render(net, W, H)[0].cpu().numpy()
    assert auto.last_render_info()["overlap_kernel"] and auto.last_kernel_name().startswith("render_kernel<%d," % (C // 16)) and np.array_equal(np.nan_to_num(ic), np.nan_to_num(ib))
    # the stripes of a multi-GPU frame: rank 1 of 2, compact image against the rows of the whole frame
    stripe = capi.render_stripes(capi.Scene(**kw), net, W, H, 16, 1, 2)
    rows = tiles_rows(H, 16, 1, 2)
    assert np.abs(np.nan_to_num(stripe.cpu().numpy()[:4]) - np.nan_to_num(ib[:4][:, rows])).max() < 1e-3


def tiles_rows(H, stripe, rank, world):
    from fvsrn_amd import tiles
    return tiles.owned_rows(H, stripe, rank, world)


def test_two_rank_processes_share_the_gpu_and_assemble_the_frame():
    """The N > 1 path end to end on one GPU: bench.py --gpus 2 as two rank processes (gloo; RCCL needs one GPU per rank), each
    rendering its round-robin stripes with the HIP kernels, all-gather, frame check against a whole-frame render."""
    import __graft_entry__ as entry
    line = entry.two_rank_smoke(steps=3)
    assert line["gathered_frame_matches_single_gpu_frame"] is True and line["backend"] == "gloo"
    assert line["value"] > 1e7  # samples/s of the whole job (two ranks on one GPU, frames gathered through host memory by gloo)


@pytest.mark.gpu
def test_stripe_launch_shapes_render_the_same_rows():
    """A rank's launch is persistent with a reserve of workgroup slots for the collective of the previous frame (r03; until then: bounded
    waves).  The launch shape only changes which wave renders which tile: persistent with any reserve, bounded waves and one unit per
    wave give the same image (bit for bit: one tile is one wave's work in every shape); the option validates its range."""
    import torch
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=64, layers=3, activation="ReLU", output_mode="density", grid=(16, 8), seed=62, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4,
                             grid_scale=0.3)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    kw = make_scene_kwargs(stepsize=1 / 128, early_out=False, tf_kind=oracle.TF_IDENTITY, tf_scale_absorption=20.0, tf_scale_emission=1.0)
    W, H = 1024, 512  # a rank's share large enough that the launch exceeds what the chip holds at once
    images = []
    for opts in (dict(), dict(persistent_reserve=0), dict(persistent_reserve=100), dict(persistent=0), dict(persistent=0, unit_quota=0)):
        scene = capi.Scene(**kw).set_option("depth_segments", 1)
        for k, v in opts.items():
            scene.set_option(k, v)
        images.append(torch.nan_to_num(capi.render_stripes(scene, net, W, H, 16, 1, 2), nan=-7.0).clone())
    assert float(images[0][3].max()) > 0.2
    for i, img in enumerate(images[1:]):
        assert torch.equal(images[0], img), (i + 1, float((images[0] - img).abs().max()), int((images[0] != img).sum()))
    scene = capi.Scene(**kw)
    assert scene.get_option("persistent_reserve") == -1
    with pytest.raises(capi.FvsrnError):
        scene.set_option("persistent_reserve", 5000)


DETERMINISM_CASES = [
    # (channels, layers, activation, grid encoding, scene options): the configurations of tools/dev/determinism.py that differed most often
    # in r03 (one launch in twenty ... every launch), a 96-wide network and a BYTE_GAUSSIAN grid (decoded in the kernel)
    (64, 3, "ReLU", 0, dict(overlap_kernel=1)), (64, 3, "SnakeAlt", 0, dict(overlap_kernel=1)), (64, 3, "ReLU", 0, dict(overlap_kernel=1, persistent=0)),
    # (cell_table = 0: the gather kernels -- grid_tap is where hipcc emitted the selection; the shaded renderers and evaluate_points run that code)
    (32, 4, "ReLU", 0, dict(cell_table=0)), (32, 4, "SnakeAlt", 0, dict(small_kernel=0, cell_table=0)), (96, 3, "SnakeAlt", 0, dict(cell_table=0)),
    (64, 3, "ReLU", 2, dict()), (48, 3, "Sine", 2, dict()),
    # the cell-table kernels (the default of FLOAT / BYTE_LINEAR grids since r04): register-resident, LDS 32 / 64 / 96 wide
    (32, 4, "ReLU", 0, dict(cell_table=1)), (32, 4, "SnakeAlt", 0, dict(small_kernel=0, cell_table=1)), (64, 3, "ReLU", 0, dict(cell_table=1)),
    (96, 3, "SnakeAlt", 0, dict(cell_table=1))]


def _determinism_scene(C, layers, act, enc, opts):
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=C, layers=layers, activation=act, output_mode="density", grid=(16, 8), seed=62, box_min=(-0.5, -0.5, -0.5),
                             fourier_std=0.4, grid_scale=0.3, encoding=enc)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    kw = make_scene_kwargs(stepsize=1 / 128, early_out=False, tf_kind=oracle.TF_IDENTITY, tf_scale_absorption=20.0, tf_scale_emission=1.0)
    scene = capi.Scene(**kw).set_option("depth_segments", 1)
    for k, v in opts.items():
        scene.set_option(k, v)
    return net, scene


@pytest.mark.gpu
@pytest.mark.parametrize("C,layers,act,enc,opts", DETERMINISM_CASES)
def test_latent_grid_kernels_are_bit_identical_from_launch_to_launch(C, layers, act, enc, opts):
    """The same frame, sixty launches: identical bits (a difference in one launch of twenty -- the rarest case r03 saw -- is found with 95 %).
    Cause, found in r04 (profiles/r04/nondeterminism_r04.md): v_pk_*_f32 with op_sel:[0,1] reads an operand as 0 in lanes 48-63 next to MFMA
    waves; the build rewrites that selection (tools/fix_pk_opsel.py).  The sharper guard is the next test."""
    import torch
    net, scene = _determinism_scene(C, layers, act, enc, opts)
    first = None
    for i in range(60):
        img = torch.nan_to_num(scene.render(net, 1024, 512)[0], nan=-7.0).clone()
        if first is None:
            first = img
            assert float(img[3].max()) > 0.2
        else:
            assert torch.equal(first, img), (i, int((first != img).sum()), float((first - img).abs().max()))


def _aggressor():
    import ctypes
    path = os.path.join(util.ROOT, "tools", "dev", "bin", "libaggressor.so")
    assert os.path.exists(path), "tools/dev/bin/libaggressor.so is missing: __graft_entry__.build() compiles it (tools/dev/bisect/aggressor.hip)"
    lib = ctypes.CDLL(path)
    lib.aggressor.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [1, 2], ids=["mfma32x32x16", "mfma16x16x32"])
@pytest.mark.parametrize("C,layers,act,enc,opts", DETERMINISM_CASES)
def test_latent_grid_kernels_are_bit_identical_next_to_mfma_waves(C, layers, act, enc, opts, kind):
    """The r04 reproduction as a guard: the kernel under test shares the SIMDs with waves of a second kernel on a second stream that do nothing
    but issue MFMAs (tools/dev/bisect/aggressor.hip).  Next to the 16x16x32 aggressor the r03 binary without its s_nops differed in EVERY
    launch (350 000 values per frame, all in lanes 48-63); a packed-fp32 instruction with the bad selection anywhere in these kernels shows
    up here in one launch, not in one of twenty."""
    import ctypes
    import time
    import torch
    agg = _aggressor()
    net, scene = _determinism_scene(C, layers, act, enc, opts)
    ref = torch.nan_to_num(scene.render(net, 1024, 512)[0], nan=-7.0).clone()  # alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.render(net, 1024, 512)
    torch.cuda.synchronize()
    alone_us = (time.perf_counter() - t0) * 1e6
    side = torch.cuda.Stream()
    for i in range(6):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            assert agg.aggressor(kind, 512, int(alone_us * 8) + 3000, ctypes.c_void_p(side.cuda_stream)) == 0
        time.sleep(0.0005)
        img = torch.nan_to_num(scene.render(net, 1024, 512)[0], nan=-7.0)
        torch.cuda.synchronize()
        assert torch.equal(ref, img), (i, int((ref != img).sum()), float((ref - img).abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["c32l4_fourier_512x256", "c64l6_grid16_time16_1024x512"])
def test_rccl_route_runs_on_one_gpu_with_a_one_rank_group(config):
    """The multi-GPU route with its REAL backend on a one-GPU box: a fresh process initialises backend "nccl" (= RCCL) with world_size 1 before
    any GPU call, StripeRenderer(force_collective=True) renders the compact stripe image, runs all_gather_into_tensor on the collective's
    stream and assembles the frame; bench.py compares it BITWISE with scene.render of the same frame and reports the per-rank render / gather
    times.  (N > 1 needs N GPUs: the driver's scaling tier; the gloo tests cover the partition logic.)"""
    import json
    import subprocess
    import sys
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FVSRN_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "4", "--warmup", "1", "--spinup-ms", "0",
                        "--no-twin", "--no-cpu-baseline", "--config", config], capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads(lines[-1])
    assert line["backend"] == "nccl" and line["world_size"] == 1 and line["force_collective"] is True
    assert line["gathered_frame_matches_single_gpu_frame"] is True
    assert len(line["per_rank"]) == 1 and line["per_rank"][0]["gather_ms"] > 0 and line["per_rank"][0]["render_ms"] > 0
    # (one rank, not pipelined: its two streams -- render, collective -- run side by side)
    assert line["stripe_launches"]["hw_streams_concurrent"] >= 1.6 and line["stripe_launches"]["persistent"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--gather", "all"], ["--gather", "root", "--payload", "rgba8", "--frames-per-submit", "4"], ["--gather", "all", "--payload", "rgba8"],
                                   ["--frames-per-submit", "3"]])
def test_rccl_route_gather_to_root_rgba8_payload_and_batches_on_one_gpu(extra):
    """r05: the other forms of the exchange with the REAL backend (RCCL, one-rank group): dist.gather to rank 0, packed RGBA8 words (ExtractColor before
    the collective; with a batch: inside fvsrn_render_stripes_batch), K frames per library call and collective.  bench.py checks the last frame against a
    whole-frame render -- bitwise for the planes, word for word for RGBA8 -- and reports the host time per frame."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29548")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FVSRN_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "1", "--force-collective", "--steps", "8", "--warmup", "1", "--spinup-ms", "0",
                        "--no-twin", "--no-cpu-baseline", "--config", "c32l4_fourier_512x256"] + extra, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads(lines[-1])
    assert line["backend"] == "nccl" and line["gathered_frame_matches_single_gpu_frame"] is True
    # (r06: the default of --gather is north_star's form, the gather to rank 0)
    assert line["gather"] == ("all" if "all" in extra else "root") and line["payload"] == ("rgba8" if "rgba8" in extra else "planes")
    assert line["per_rank"][0]["host_us_per_frame"] > 0 and line["per_rank"][0]["gather_ms"] > 0
    # r06: the record of the one multi-GPU shot -- who took part, the same-run world-1 reference, the three other (gather, payload) combinations
    assert line["rccl"]["world_size_seen"] == 1 and line["rccl"]["nccl_version"] and line["rccl"]["ranks"][0]["device_name"]
    assert line["world1_same_run"]["value"] > 0 and 0.3 < line["efficiency_vs_same_run_world1"] < 1.3
    combos = {(v["gather"], v["payload"]) for v in line["variants"]} | {(line["gather"], line["payload"])}
    assert len(line["variants"]) == 3 and combos == {("root", "planes"), ("all", "planes"), ("root", "rgba8"), ("all", "rgba8")}
    assert all(v["value"] > 0 and v["steps"] == 20 for v in line["variants"])
    assert line["collective_bytes_per_frame_and_rank"] == (4 if "rgba8" in extra else 32) * 512 * 512


@pytest.mark.gpu
@pytest.mark.parametrize("world,rank", [(1, 0), (4, 1)])
@pytest.mark.parametrize("grid,keys", [(None, 1), ((16, 8), 4)])
def test_batch_render_equals_frame_by_frame_renders(world, rank, grid, keys):
    """fvsrn_render_stripes_batch: K camera poses (and times) in one call -- frames that share their time several per LAUNCH (a work unit is (frame, pixel
    tile)), on one lane and on two (two scenes on two streams) -- are bit for bit the frames fvsrn_render / fvsrn_render_stripes give one by one with one
    depth segment per ray (a multi-frame launch does not cut rays; with the automatic segments of a small single launch the sums are re-associated:
    2e-4); the RGBA8 copy of every frame is fvsrn_extract_color_rgba8 of its planes."""
    import torch
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=32, layers=4, activation="SnakeAlt", output_mode="density", grid=grid, seed=61, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4,
                             time_grids=keys)
    blob = volnet_io.save_volnet(vn)
    net, serial = capi.Network.from_volnet(blob), capi.Network.from_volnet(blob)
    W, H, stripe, K = 72, 64, 8, 5
    kws = [_scene_kw(0.4 + 0.9 * i) for i in range(K)]
    times = [0.3 + 0.55 * i for i in range(K)] if keys > 1 else None
    ref, ref8, ref_auto = [], [], []
    scene, scene_auto = capi.Scene(**kws[0]).set_option("depth_segments", 1), capi.Scene(**kws[0])
    for i, kw in enumerate(kws):
        scene.update(**kw)
        scene_auto.update(**kw)
        if times:
            serial.set_time_and_ensemble(times[i], 0)
        img = scene.render(serial, W, H)[0] if world == 1 else capi.render_stripes(scene, serial, W, H, stripe, rank, world)
        ref.append(img.clone())
        ref8.append(capi.extract_color_part(img, capi.CHANNEL_COLOR, True, 1.5))
        auto = scene_auto.render(serial, W, H)[0] if world == 1 else capi.render_stripes(scene_auto, serial, W, H, stripe, rank, world)
        ref_auto.append(auto.clone())
        torch.cuda.synchronize()
    cams = np.stack([np.concatenate([kw["eye"], kw["right"], kw["up"]]) for kw in kws])
    for lanes in (1, 2):
        scenes = [capi.Scene(**kws[0]).set_option("depth_segments", 1) for _ in range(lanes)]
        streams = [torch.cuda.Stream() for _ in range(lanes)]
        rows = H if world == 1 else capi.stripe_rows(H, stripe, rank, world)
        rgba = torch.zeros((K, rows, W), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        out = capi.render_stripes_batch(scenes, [st.cuda_stream for st in streams], net, W, H, stripe, rank, world, cams, times=times, rgba8=rgba,
                                        use_tonemapping=True, max_exposure=1.5)
        torch.cuda.synchronize()
        for i in range(K):
            assert torch.equal(torch.nan_to_num(out[i], nan=-7.0), torch.nan_to_num(ref[i], nan=-7.0)), (lanes, i)
            assert torch.equal(rgba[i], ref8[i]), (lanes, i)
            assert float((torch.nan_to_num(out[i][:7]) - torch.nan_to_num(ref_auto[i][:7])).abs().max()) < 2e-4
        assert float(out[:, 3].max()) > 0.05
        if times is None:  # several poses per launch: ceil(K / lanes) each
            assert scenes[0].last_render_info()["segments"] == 1 and ("frames %d" % -(-K // lanes)) in capi.debug_state()
    with pytest.raises(capi.FvsrnError):  # one scene on two streams
        capi.render_stripes_batch([scenes[0], scenes[0]], [s.cuda_stream for s in (torch.cuda.Stream(), torch.cuda.Stream())], net, W, H, stripe, rank, world, cams)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["shaded_fd_64", "adjoint_48_grid", "rgbo_96", "byte_gaussian_64", "direction_32"])
def test_batch_render_through_the_other_kernel_families(case):
    """Several poses per launch is a property of render_body, which every render kernel family shares: the LDS kernels with finite-difference / adjoint
    shading, a colour network at 96 channels, the BYTE_GAUSSIAN gather kernel (fragment-major), a view-dependent network -- each batch against its frames
    rendered one by one (one depth segment per ray: bit for bit)."""
    import torch
    from fvsrn_amd import capi, volnet_io
    net_kw, scene_extra = {
        "shaded_fd_64": (dict(C=64, layers=3, activation="ReLU", output_mode="density"),
                         dict(gradient_mode=capi.GRADIENT_FINITE_DIFFERENCES, finite_differences_stepsize=0.01, brdf=dict(enable_phong=True, ambient=0.2, specular=0.3))),
        "adjoint_48_grid": (dict(C=48, layers=3, activation="SnakeAlt", output_mode="density", grid=(16, 8)),
                            dict(gradient_mode=capi.GRADIENT_ADJOINT_METHOD, brdf=dict(enable_phong=True, ambient=0.2, specular=0.3))),
        "rgbo_96": (dict(C=96, layers=3, activation="SnakeAlt", output_mode="rgbo"), dict(tf_kind=oracle.TF_NONE)),
        "byte_gaussian_64": (dict(C=64, layers=3, activation="ReLU", output_mode="density", grid=(16, 8), encoding=2), dict()),
        "direction_32": (None, dict(tf_kind=oracle.TF_NONE)),
    }[case]
    if net_kw is None:
        d, meta = util.load_golden("g1_dir1_c32l4_snakealt_rgbo")
        vn = util.golden_to_volnet(d, meta, box_min=(-0.5, -0.5, -0.5))
    else:
        vn = util.random_network(seed=64, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4, **net_kw)
    net, serial = capi.Network.from_volnet(volnet_io.save_volnet(vn)), capi.Network.from_volnet(volnet_io.save_volnet(vn))
    W, H, stripe, K = 56, 48, 8, 3
    base = dict(tf_kind=oracle.TF_GAUSSIAN, tf_table=GAUSS_TF)
    base.update(scene_extra)
    if base["tf_kind"] == oracle.TF_NONE:
        base.pop("tf_table")
    kws = [_scene_kw(0.5 + 1.1 * i, **base) for i in range(K)]
    cams = np.stack([np.concatenate([kw["eye"], kw["right"], kw["up"]]) for kw in kws])
    for world, rank in ((1, 0), (2, 1)):
        scene = capi.Scene(**kws[0]).set_option("depth_segments", 1)
        ref = []
        for kw in kws:
            scene.update(**kw)
            ref.append((scene.render(serial, W, H)[0] if world == 1 else capi.render_stripes(scene, serial, W, H, stripe, rank, world)).clone())
        family = scene.last_kernel_name()
        batch_scene = capi.Scene(**kws[0]).set_option("depth_segments", 1)
        out = capi.render_stripes_batch([batch_scene], [torch.cuda.current_stream().cuda_stream], net, W, H, stripe, rank, world, cams)
        torch.cuda.synchronize()
        assert batch_scene.last_kernel_name() == family and "frames %d" % K in capi.debug_state()
        for i in range(K):
            assert torch.equal(torch.nan_to_num(out[i], nan=-7.0), torch.nan_to_num(ref[i], nan=-7.0)), (case, world, i, family)
        assert float(out[:, 3].max()) > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["depth", "depth_with_empty_pixels", "color", "normal"])
def test_extract_color_of_image_parts_with_a_merged_depth_range(mode):
    """ExtractColor on the stripes of a frame before they travel (payload "rgba8"): the parts of an image, each converted on its own with the depth range
    merged over the parts (fvsrn_depth_range: {-min, max, nan flag}, element-wise maximum = the all-reduce of tiles.StripeRenderer), equal the conversion
    of the whole image -- RGBA8 words and fp32 planes, bitwise; a NaN depth in any part poisons all of them like the reference's min() / max()."""
    import torch
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=32, layers=4, activation="ReLU", output_mode="densitygrad", seed=62, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    W, H = 96, 64
    img = capi.Scene(**_scene_kw(0.9, tf_kind=oracle.TF_GAUSSIAN, tf_table=GAUSS_TF)).render(net, W, H).clone()
    if mode == "depth":
        img[0, 7] = torch.nan_to_num(img[0, 7], nan=5.0)
    ch = {"depth": capi.CHANNEL_DEPTH, "depth_with_empty_pixels": capi.CHANNEL_DEPTH, "color": capi.CHANNEL_COLOR, "normal": capi.CHANNEL_NORMAL}[mode]
    whole8 = capi.extract_color(img, ch, mode == "color", 2.0, rgba8=True)
    whole4 = capi.extract_color(img, ch, mode == "color", 2.0)[0]
    parts = [img[0, :, :24].contiguous(), img[0, :, 24:40].contiguous(), img[0, :, 40:].contiguous()]
    r3 = torch.stack([capi.depth_range(p) for p in parts]).max(dim=0).values
    full_r3 = capi.depth_range(img)
    assert torch.equal(r3, full_r3)
    if mode == "depth_with_empty_pixels":
        assert float(r3[2]) == 1.0  # (rays that miss the box: alpha 0, depth NaN)
    else:
        assert float(r3[2]) == 0.0 or ch != capi.CHANNEL_DEPTH
    got8 = torch.cat([capi.extract_color_part(p, ch, mode == "color", 2.0, rgba8=True, depth_range3=r3) for p in parts], dim=0)
    got4 = torch.cat([capi.extract_color_part(p, ch, mode == "color", 2.0, rgba8=False, depth_range3=r3) for p in parts], dim=1)
    torch.cuda.synchronize()
    assert torch.equal(got8, whole8)
    assert torch.equal(torch.nan_to_num(got4, nan=-7.0), torch.nan_to_num(whole4, nan=-7.0))
    if mode == "depth":
        assert len(torch.unique(whole8)) > 16
    # the restatement of the merged form
    ref_r3 = np.max(np.stack([oracle.depth_range(p.cpu().numpy()) for p in parts]), axis=0)
    assert np.array_equal(ref_r3, r3.cpu().numpy())


@pytest.mark.gpu
def test_cell_tables_are_built_by_the_launches_that_use_them():
    """the cell tables of a latent grid are built lazily.  A time-animated network whose launches take the gathers (small image: footprint
    rule) allocates and builds none; the first launch through the table builds it; from then on a time change rebuilds it with the blend; a shaded launch
    builds the plain-image table on its own; a launch that gathers again stops the rebuilds."""
    import torch
    from fvsrn_amd import capi, volnet_io
    vn = util.random_network(C=32, layers=4, activation="ReLU", output_mode="density", grid=(16, 8), seed=63, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4, time_grids=4)
    net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
    kw = _scene_kw(0.7)
    small, large = capi.Scene(**kw), capi.Scene(**kw)
    t = [0.2]

    def step(scene, size):
        t[0] += 0.3
        net.set_time_and_ensemble(t[0], 0)
        img = scene.render(net, size, size)
        torch.cuda.synchronize()
        return img, scene.last_render_info()["cell_table"], net.cell_table_stats()

    for _ in range(3):
        _, used, st = step(small, 64)
        assert not used and st["table_bytes"] > 0 and st["builds"] == 0 and st["builds_plain"] == 0 and st["resident_bytes"] == 0, st
    img_c, used, st = step(large, 512)
    assert used and st["builds"] == 1 and st["builds_plain"] == 0 and st["resident_bytes"] == st["table_bytes"], st
    for i in range(3):  # (two working grids: the second one's buffer appears with the next blend)
        _, used, st = step(large, 512)
        assert used and st["builds"] == 2 + i and st["builds_plain"] == 0, st
    assert st["resident_bytes"] == 2 * st["table_bytes"]
    before = st["builds"]
    for _ in range(2):
        _, used, st = step(small, 64)
        assert not used
    assert st["builds"] <= before + 1, st  # (the blend of the first gathering frame still saw the flag of the frame before)
    # same picture either way (and against the forced gather path)
    net.set_time_and_ensemble(1.3, 0)
    a = capi.Scene(**kw).render(net, 512, 512).clone()
    b = capi.Scene(**kw).set_option("cell_table", 0).render(net, 512, 512)
    assert float((torch.nan_to_num(a[0, :4]) - torch.nan_to_num(b[0, :4])).abs().max()) < 1e-3
    # a shaded render (the plain weight image): its own table, built by that launch
    shaded = capi.Scene(**dict(kw, gradient_mode=capi.GRADIENT_FINITE_DIFFERENCES, finite_differences_stepsize=0.01, brdf=dict(enable_phong=True, ambient=0.2, specular=0.3)))
    shaded.render(net, 512, 512)
    torch.cuda.synchronize()
    st2 = net.cell_table_stats()
    assert shaded.last_render_info()["cell_table"] and st2["builds_plain"] == 1, st2


@pytest.mark.gpu
def test_stream_concurrency_probe_and_the_shared_queue_warning():
    """fvsrn_probe_stream_concurrency measures what the process got (n fresh streams on q hardware queues run n / ceil(n / q)-wide), and
    StripeRenderer measures ITS streams: in a child limited to two hardware queues the collective's stream shares one with a render stream --
    the pipeline warns and keeps bounded-wave stripes; with GPU_MAX_HW_QUEUES=8 its three streams run side by side and the stripes go persistent."""
    import subprocess
    import sys
    code = ("import sys, warnings; sys.path.insert(0, %r); import torch, numpy as np; from fvsrn_amd import capi, tiles, synthetic, volnet_io\n"
            "c = capi.probe_stream_concurrency(6, 3000)\n"
            "net = capi.Network.from_volnet(volnet_io.save_volnet(synthetic.random_network(C=32, layers=4, activation='ReLU', seed=1)))\n"
            "eye, right, up = capi.camera_on_a_sphere('Ym', (0, 0, 0), 0.4, 0.7, 1.6)\n"
            "kw = dict(eye=eye, right=right, up=up, fov_y_radians=0.8, stepsize=1 / 32, tf_kind=capi.TF_IDENTITY, tf_scale_absorption=20.0, tf_scale_emission=1.0)\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always')\n"
            "    p = tiles.StripeRenderer(net, 64, 64, kw, rank=1, world=2, stripe=16)\n"
            "print('RESULT', c, p.hw_streams_concurrent, int(p.persistent_stripes), len([x for x in w if 'GPU_MAX_HW_QUEUES' in str(x.message)]), p.scenes[0].get_option('persistent'))\n") % util.ROOT
    res = {}
    for q in ("2", "8"):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        out = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
        assert r.returncode == 0 and out, (r.stdout[-1000:], r.stderr[-3000:])
        _, fresh, mine, pers, nwarn, opt = out[-1].split()
        res[q] = (float(fresh), float(mine), int(pers), int(nwarn), int(opt))
    assert 1.0 <= res["2"][0] <= 2.2 and res["2"][1] < 2.4 and res["2"][2:] == (0, 1, -1), res
    assert res["8"][0] >= 5.0 and res["8"][1] >= 2.6 and res["8"][2:] == (1, 0, 1), res
