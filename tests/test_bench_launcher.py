"""bench.py --gpus N starts its own rank processes when no launcher did (it used to render on one GPU and print
n_gpus 1).  CPU-only: the rank processes are stand-ins that report the environment they were given."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture
def stub(tmp_path):
    p = tmp_path / "rank_stub.py"
    p.write_text(textwrap.dedent("""
        import json, os, sys
        rank = int(os.environ["RANK"])
        if "--fail-rank1" in sys.argv and rank == 1:
            sys.exit(7)
        if rank == 0:
            print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} | {"argv": sys.argv[1:]}))
        else:
            print("rank %d speaking" % rank)   # must not reach the launcher's stdout
    """))
    return [sys.executable, str(p)]


def test_launcher_starts_one_process_per_rank_and_relays_rank0(stub, monkeypatch, capfd):
    import bench
    monkeypatch.setenv("FVSRN_BENCH_BACKEND", "gloo")  # all ranks share the visible devices: no device-count requirement
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.launch_ranks(3, ["--gpus", "3", "--steps", "5"], child=stub)
    out = capfd.readouterr().out
    assert rc == 0
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["RANK"] == "0" and d["WORLD_SIZE"] == "3" and d["MASTER_ADDR"] == "127.0.0.1" and int(d["MASTER_PORT"]) > 0
    assert d["argv"] == ["--gpus", "3", "--steps", "5"]
    assert "speaking" not in out


def test_launcher_reports_a_failed_rank(stub, monkeypatch):
    import bench
    monkeypatch.setenv("FVSRN_BENCH_BACKEND", "gloo")
    assert bench.launch_ranks(2, ["--fail-rank1"], child=stub) == 7


def test_launcher_refuses_fewer_gpus_than_ranks(stub, monkeypatch):
    """RCCL needs one GPU per rank: with fewer visible devices the run fails instead of measuring one GPU."""
    import bench
    import torch
    monkeypatch.delenv("FVSRN_BENCH_BACKEND", raising=False)
    n = torch.cuda.device_count() + 1
    assert bench.launch_ranks(n, [], child=stub) == 2


def test_gpus_flag_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE is 2" in (r.stderr + r.stdout)


def test_plain_multi_gpu_invocation_does_not_fall_back_to_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FVSRN_BENCH_BACKEND")}
    import torch
    if torch.cuda.device_count() >= 64:
        pytest.skip("enough GPUs for the request")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
    assert '"n_gpus"' not in r.stdout
