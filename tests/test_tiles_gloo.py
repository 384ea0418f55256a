"""
The N > 1 path on CPU: two processes (gloo), round-robin row stripes, all-gather, assembly.
The stripe images come from the CPU oracle (this is a test of the partition / exchange logic in
fv-srn_amd/tiles.py, the same code bench.py drives with RCCL): the free functions, and the frame pipeline
``StripeRenderer`` (double-buffered local / gathered images) with the oracle handed in as its render function.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from fvsrn_amd import capi, tiles


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, stripe, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    vn = util.random_network(C=32, layers=4, activation="ReLU", output_mode="density", seed=5, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4)
    eye, right, up = oracle.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.7, 1.6)
    kw = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45)), stepsize=1 / 16, tf_kind=oracle.TF_IDENTITY,
              tf_scale_absorption=20.0)
    net, scene = oracle.OracleNetwork(vn, oracle.ACC_FLOAT), oracle.OracleScene(**kw)
    rows = tiles.owned_rows(H, stripe, rank, world)
    assert len(rows) == capi.stripe_rows(H, stripe, rank, world)
    local = np.zeros((8, len(rows), W), np.float32)
    for i, y in enumerate(rows):  # each rank computes ONLY the rows it owns
        img, _ = scene.render(net, W, H, y, y + 1)
        local[:, i] = img[:, y]
    gathered = tiles.all_gather_frame(torch.from_numpy(local))
    frame = tiles.assemble(gathered, H, stripe)
    full, _ = scene.render(net, W, H)
    ok = np.array_equal(np.nan_to_num(frame[0].numpy(), nan=-1), np.nan_to_num(full, nan=-1))
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(tmp, "w").write(str(int(flag.item())))
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, H, W, stripe, tmp):
    """StripeRenderer on the CPU: three frames (two buffers, so frame 2 reuses frame 0's), every gathered frame against a
    whole-frame oracle render of the same camera."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    vn = util.random_network(C=32, layers=4, activation="ReLU", output_mode="density", seed=5, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4)
    net = oracle.OracleNetwork(vn, oracle.ACC_FLOAT)

    def scene_kw(yaw):
        eye, right, up = oracle.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, yaw, 1.6)
        return dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45)), stepsize=1 / 16, tf_kind=oracle.TF_IDENTITY,
                    tf_scale_absorption=20.0)

    calls = []

    def render(kw, out, r, w, st):  # the rows this rank owns, in compact order
        assert (r, w, st) == (rank, world, stripe)
        scene = oracle.OracleScene(**kw)
        rows = tiles.owned_rows(H, st, r, w)
        assert tuple(out.shape) == (8, len(rows), W)
        for i, y in enumerate(rows):
            img, _ = scene.render(net, W, H, y, y + 1)
            out[:, i] = torch.from_numpy(img[:, y])
        calls.append(len(rows))

    pipe = tiles.StripeRenderer(net, W, H, scene_kw(0.0), rank=rank, world=world, stripe=stripe, device="cpu", render=render)
    assert pipe.pipelined and pipe.rows * world == H
    ok = True
    yaws = [0.3, 1.1, 2.0]
    for i, yaw in enumerate(yaws):
        b = pipe.submit(i, scene_kw(yaw))
        assert b == i % pipe.buffers and pipe.buffers == 3
        pipe.finish()
        full, _ = oracle.OracleScene(**scene_kw(yaw)).render(net, W, H)
        frame = pipe.frame(b)
        assert tuple(frame.shape) == (1, 8, H, W)
        ok = ok and np.array_equal(np.nan_to_num(frame[0].numpy(), nan=-1), np.nan_to_num(full, nan=-1))
        ok = ok and tiles.frames_match(torch.from_numpy(full)[None], frame)
    ok = ok and len(calls) == len(yaws) and pipe.frames_submitted == len(yaws)
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(tmp, "w").write(str(int(flag.item())))
    dist.destroy_process_group()


def _payload_worker(rank, world, port, H, W, stripe, tmp):
    """r05: what travels.  gather to rank 0 instead of to everyone, packed RGBA8 words instead of eight fp32 planes (ExtractColor on the rank's own rows; the
    DEPTH channel mode needs the depth range of the whole frame: one all-reduce of three floats), several frames per submit -- every combination against
    the single-process result, BITWISE (the partition and the collectives move values, they compute nothing)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    vn = util.random_network(C=32, layers=4, activation="ReLU", output_mode="density", seed=5, box_min=(-0.5, -0.5, -0.5), fourier_std=0.4)
    net = oracle.OracleNetwork(vn, oracle.ACC_FLOAT)

    def scene_kw(yaw):
        eye, right, up = oracle.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, yaw, 1.6)
        return dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45)), stepsize=1 / 16, tf_kind=oracle.TF_IDENTITY,
                    tf_scale_absorption=20.0)

    full_cache = {}

    def full_frame(yaw, finite_depth):
        key = (yaw, finite_depth)
        if key not in full_cache:
            img, _ = oracle.OracleScene(**scene_kw(yaw)).render(net, W, H)
            if finite_depth:
                img[7] = np.nan_to_num(img[7], nan=9.0)
            full_cache[key] = img
        return full_cache[key]

    def make_render(finite_depth):
        def render(kw, out, r, w, st):  # this rank's rows of the frame of that camera, compact order
            yaw = [y for y in YAWS if np.allclose(scene_kw(y)["eye"], kw["eye"])][0]
            rows = tiles.owned_rows(H, st, r, w)
            out[:] = torch.from_numpy(full_frame(yaw, finite_depth)[:, rows])
        return render

    def extract(local, mode, tonemap, max_exposure, range3):
        return torch.from_numpy(oracle.rgba_to_int(oracle.extract_color(local.numpy(), mode, tonemap, max_exposure,
                                                                        depth_range3=None if range3 is None else range3.numpy())).astype(np.int64)).to(torch.int32)

    YAWS = [0.3, 1.1, 2.0, 2.6, 3.3, 4.1]
    ok = True
    cases = [("root", "planes", capi.CHANNEL_COLOR, 1, False), ("all", "rgba8", capi.CHANNEL_COLOR, 1, False), ("root", "rgba8", capi.CHANNEL_NORMAL, 1, False),
             ("root", "rgba8", capi.CHANNEL_DEPTH, 1, False), ("all", "rgba8", capi.CHANNEL_DEPTH, 2, True), ("root", "rgba8", capi.CHANNEL_DEPTH, 1, True),
             ("root", "planes", capi.CHANNEL_COLOR, 3, False), ("all", "planes", capi.CHANNEL_COLOR, 2, False), ("all", "rgba8", capi.CHANNEL_COLOR, 3, False)]
    for gather, payload, mode, K, finite in cases:
        tone = mode == capi.CHANNEL_COLOR and payload == "rgba8"
        pipe = tiles.StripeRenderer(net, W, H, scene_kw(YAWS[0]), rank=rank, world=world, stripe=stripe, device="cpu", render=make_render(finite), gather=gather,
                                    payload=payload, channel_mode=mode, use_tonemapping=tone, max_exposure=1.5, extract=extract,
                                    depth_range=lambda local: torch.from_numpy(oracle.depth_range(local.numpy())), frames_per_submit=K)
        batches = [YAWS[i:i + K] for i in range(0, len(YAWS), K)]
        for bi, yaws in enumerate(batches):
            b = pipe.submit_batch(bi, [scene_kw(y) for y in yaws]) if K > 1 else pipe.submit(bi, scene_kw(yaws[0]))
            pipe.finish()
            for k, yaw in enumerate(yaws):
                frame = pipe.frame(b, k)
                if gather == "root" and rank != 0:
                    ok = ok and frame is None
                    continue
                full = full_frame(yaw, finite)
                if payload == "planes":
                    ok = ok and tuple(frame.shape) == (1, 8, H, W) and np.array_equal(np.nan_to_num(frame[0].numpy(), nan=-1), np.nan_to_num(full, nan=-1))
                else:
                    want = oracle.rgba_to_int(oracle.extract_color(full, mode, tone, 1.5)).astype(np.int64).astype(np.int32)
                    ok = ok and tuple(frame.shape) == (H, W) and frame.dtype == torch.int32 and np.array_equal(frame.numpy(), want)
                    if mode == capi.CHANNEL_DEPTH and finite:  # (not the all-NaN image a depth plane with empty pixels gives)
                        ok = ok and len(np.unique(want)) > 4
        ok = ok and pipe.frames_submitted == len(YAWS) and pipe.host_us_per_frame > 0
        if not ok:
            print("payload case failed on rank %d: %r" % (rank, (gather, payload, mode, K, finite)), flush=True)
            break
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(tmp, "w").write(str(int(flag.item())))
    dist.destroy_process_group()


def test_stripe_renderer_gather_to_root_rgba8_payload_and_batches_two_ranks(tmp_path):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_payload_worker, args=(2, _free_port(), 32, 24, 8, out), nprocs=2, join=True)
    assert open(out).read() == "1"


def test_stripe_renderer_pipeline_two_ranks(tmp_path):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_pipeline_worker, args=(2, _free_port(), 32, 24, 8, out), nprocs=2, join=True)
    assert open(out).read() == "1"


def test_stripe_renderer_refuses_the_cpu_without_a_render_function():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tiles.StripeRenderer(None, 16, 16, {}, device="cpu")
    with pytest.raises(ValueError):  # ranks would own different row counts
        tiles.StripeRenderer(None, 16, 40, {}, world=3, stripe=8, device="cpu", render=lambda *a: None)


@pytest.mark.parametrize("world,stripe,H", [(2, 8, 32), (2, 16, 64)])
def test_two_rank_stripe_render_and_gather(tmp_path, world, stripe, H):
    out = str(tmp_path / "ok.txt")
    mp.spawn(_worker, args=(world, _free_port(), H, 24, stripe, out), nprocs=world, join=True)
    assert open(out).read() == "1"


def test_partition_covers_every_row_exactly_once():
    for H, stripe, world in [(1024, 16, 8), (1024, 16, 4), (64, 8, 2), (40, 8, 3), (8, 8, 1)]:
        seen = []
        for r in range(world):
            rows = tiles.owned_rows(H, stripe, r, world)
            assert len(rows) == capi.stripe_rows(H, stripe, r, world)
            seen += rows
        assert sorted(seen) == list(range(H))
    with pytest.raises(ValueError):
        tiles.check_even_partition(40, 8, 3)
    with pytest.raises(ValueError):
        tiles.check_even_partition(64, 12, 2)
