"""bench.py's multi-GPU record on the CPU: two gloo ranks run the SAME functions `bench.py --gpus N` runs on RCCL -- the process group
with its first collective (init_group_or_exit), who took part (collective_identity), one timed region per (gather, payload) combination on the real frame
pipeline (fv-srn_amd/tiles.py StripeRenderer, a host-side stand-in as the renderer) -- and the test asserts the shape of what rank 0 would print.  And the
failure side: a process group that cannot come up ends the run with exit code 3 and the reason on stderr, never with a one-GPU number."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

W, H, STEPS = 16, 32, 16
CFG = (32, 4, None, W, H, STEPS)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net:
    def set_time_and_ensemble(self, *a):
        pass


def _pixel(kw, c, y, x):
    return float(c) + 0.01 * y + 0.0001 * x + float(np.asarray(kw["eye"]).reshape(3)[0])


def _render(kw, out, rank, world, stripe):
    from fvsrn_amd import tiles
    if out.dim() == 4:  # the one-GPU route: (1, 8, H, W)
        for c in range(8):
            for y in range(H):
                out[0, c, y] = torch.tensor([_pixel(kw, c, y, x) for x in range(W)])
        return
    for i, y in enumerate(tiles.owned_rows(H, stripe, rank, world)):
        for c in range(8):
            out[c, i] = torch.tensor([_pixel(kw, c, y, x) for x in range(W)])


def _extract(local, channel_mode, use_tonemapping, max_exposure, range3):
    return (local[0] * 1000).to(torch.int32)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import bench
    from fvsrn_amd import capi
    bench.init_group_or_exit("gloo")
    ident = bench.collective_identity("gloo", rank, None, on_gpu=False)

    def make_runner(g, p):
        return bench.Runner(capi, _Net(), CFG, rank, world, False, 1, gather=g, payload=p, frames_per_submit=2, device="cpu", render=_render, extract=_extract)

    recs = [bench.timed_combo(make_runner, g, p, 4, 2, True) for g, p in bench.COMBOS]
    # one more pipeline: what arrives on rank 0 is the frame (gather to rank 0 of the eight planes: the primary region's form)
    r = make_runner(bench.PRIMARY_GATHER, bench.PRIMARY_PAYLOAD)
    r.frames(0, 2)
    r.finish()
    frame = r.assemble(*r.where(1))
    ok = True
    if rank == 0:
        kw = bench.build_scene_kwargs(capi, 2 * np.pi * 1 / 64, 1.0 / STEPS, False)
        want = torch.tensor([[[_pixel(kw, c, y, x) for x in range(W)] for y in range(H)] for c in range(8)])
        ok = frame is not None and bool(torch.equal(frame[0], want))
    else:
        ok = frame is None
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        with open(tmp, "w") as f:
            json.dump({"identity": ident, "variants": recs, "frame_ok": int(flag.item())}, f)
    dist.destroy_process_group()


def test_multi_gpu_record_two_gloo_ranks(tmp_path):
    import bench
    tmp = str(tmp_path / "rec.json")
    mp.spawn(_worker, args=(2, _free_port(), tmp), nprocs=2, join=True)
    d = json.load(open(tmp))
    assert d["frame_ok"] == 1
    ident = d["identity"]
    assert ident["world_size_seen"] == 2 and ident["backend"] == "gloo" and ident["nccl_version"] is None
    assert [r["rank"] for r in ident["ranks"]] == [0, 1] and len({r["pid"] for r in ident["ranks"]}) == 2
    assert all(set(r) >= {"rank", "pid", "device_index", "device_name", "pci_bus_id"} for r in ident["ranks"])
    assert [(v["gather"], v["payload"]) for v in d["variants"]] == [tuple(c) for c in bench.COMBOS]
    assert (d["variants"][0]["gather"], d["variants"][0]["payload"]) == (bench.PRIMARY_GATHER, bench.PRIMARY_PAYLOAD) == ("root", "planes")
    for v in d["variants"]:
        assert set(v) >= {"gather", "payload", "steps", "value", "unit", "ms_per_step", "frames_per_s", "collective_bytes_per_frame_and_rank", "rank0_render_ms", "rank0_gather_ms"}
        assert v["steps"] == 4 and v["ms_per_step"] > 0 and v["frames_per_s"] > 0
        assert v["collective_bytes_per_frame_and_rank"] == (32 if v["payload"] == "planes" else 4) * W * H // 2


def test_prediction_lookup_reads_the_committed_emulation():
    import bench
    p = bench.predicted_efficiency("c32l4_fourier_1024x512", 8)
    assert p is not None and 0.5 < p["efficiency_vs_world1"] <= 1.0 + 1e-9 and p["from"].startswith("profiles/")
    assert bench.predicted_efficiency("no_such_workload", 8) is None


@pytest.mark.parametrize("port", ["free-port-nobody-joins", "not-a-port"])
def test_process_group_failure_ends_the_run_with_a_reason(port):
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", FVSRN_BENCH_INIT_TIMEOUT_S="3",
               MASTER_PORT=str(_free_port()) if port.startswith("free") else "not-a-port")
    r = subprocess.run([sys.executable, "-c", "import bench; bench.init_group_or_exit('gloo'); print('UP')"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3, (r.returncode, r.stderr[-400:])
    assert "did not come up" in r.stderr and "UP" not in r.stdout
