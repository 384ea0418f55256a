#!/usr/bin/env python3
"""How long does ONE rank's share of a multi-GPU frame take on its GPU, against frame_time / world?  One GPU, no collective: the
rank's frame pipeline (fv-srn_amd/tiles.py StripeRenderer: two frames in flight on two streams, for time-dependent networks the
blend of frame i + 1 into the second working grid) is run with gather=False, for every rank of world = 2 / 4 / 8 in turn.
Prints one JSON line per configuration: the frame period of the slowest rank and the efficiency against full_frame / world.
FVSRN_STRIPE_BATCH=K: K frames per call into the library and per (stand-in) collective (StripeRenderer.submit_batch, fvsrn_render_stripes_batch) -- the
per-frame host work of the Python loop is paid once per K; the line carries host_us_per_frame either way.
usage: tools/stripe_efficiency.py [config ...]      default: the headline workload, BASELINE.json configs[3] and configs[4]"""
import importlib.util
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
import torch  # noqa: E402
from fvsrn_amd import capi, tiles, volnet_io  # noqa: E402


_OCCUPY = None
_STREAMS = None


def _streams():
    """one set of streams for every pipeline this process builds (a rank has one pipeline; this tool builds 14 per configuration)"""
    global _STREAMS
    if _STREAMS is None:
        _STREAMS = (torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream())
    return _STREAMS


def emulate_gather(pipe, b, frames=1):
    """FVSRN_STRIPE_EMULATE_GATHER=blocks,threads,microseconds: a kernel of that shape on the comm stream behind this frame's render,
    in the place of the all-gather (tools/dev/occupy.hip); the next use of buffer b waits for it like for the collective."""
    global _OCCUPY
    spec = os.environ.get("FVSRN_STRIPE_EMULATE_GATHER")
    if not spec or pipe.world == 1:
        return
    import ctypes
    if _OCCUPY is None:
        _OCCUPY = ctypes.CDLL(os.path.join(ROOT, "tools", "dev", "bin", "liboccupy.so"))
        _OCCUPY.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    blocks, threads, us = [int(v) for v in spec.split(",")]
    us *= frames  # (a batch moves `frames` times the bytes)
    with torch.cuda.stream(pipe.comm_stream):
        pipe.comm_stream.wait_event(pipe.render_done[b])
        assert _OCCUPY.occupy(blocks, threads, us, ctypes.c_void_p(pipe.comm_stream.cuda_stream)) == 0
        pipe.gather_done[b].record()


HOST_US = {}


def frame_period(net, cfg, time_keys, rank, world, frames=24, warm=8, pipelined=None, ahead=True, batch=1):
    _, _, _, W, H, steps = cfg
    kw = lambda i: b.build_scene_kwargs(capi, 2 * math.pi * (i % 64) / 64, 1.0 / steps, False)  # noqa: E731
    pipe = tiles.StripeRenderer(net, W, H, kw(0), rank=rank, world=world, stripe=int(os.environ.get("FVSRN_STRIPE_ROWS", b.STRIPE)), pipelined=pipelined, streams=_streams(), frames_per_submit=batch)
    t = lambda i: (0.25 * i) % (time_keys - 1) if time_keys > 1 else None  # noqa: E731
    tn = (lambda i: t(i + 1)) if (ahead and pipe.pipelined) else (lambda i: None)

    def run(first, count):
        if batch == 1:
            for i in range(first, first + count):
                emulate_gather(pipe, pipe.submit(i, kw(i), time=t(i), next_time=tn(i), gather=False))
            return
        for j in range(first, first + count, batch):
            idx = list(range(j, min(j + batch, first + count)))
            emulate_gather(pipe, pipe.submit_batch(j // batch, [kw(i) for i in idx], times=[t(i) for i in idx] if time_keys > 1 else None, gather=False), len(idx))

    warm, frames = -(-warm // batch) * batch, -(-frames // batch) * batch
    run(0, warm)
    pipe.finish()
    torch.cuda.synchronize()
    # r05: every pipeline spins up for FVSRN_STRIPE_SPINUP_MS (150) of wall time before it is timed, like bench.py.  Without it the pipeline measured FIRST after a
    # change of launch shape read 3 - 4 % slow whichever rank it was (rank 0 forward, rank 7 with FVSRN_STRIPE_RANK_ORDER=reverse: 0.303 / 0.312 ms against 0.282 - 0.299 for
    # the others, headline at world 8) -- and the slowest rank is the figure of this tool.  A rank process of a real run renders nothing else: it is in that steady state.
    import time
    t_end = time.perf_counter() + 1e-3 * float(os.environ.get("FVSRN_STRIPE_SPINUP_MS", "150"))
    first = warm
    while time.perf_counter() < t_end:
        run(first, warm)
        first += warm
        pipe.finish()
        torch.cuda.synchronize()
    pipe.host_seconds, pipe.frames_submitted = 0.0, 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run(first, frames)
    pipe.finish()
    e1.record()
    torch.cuda.synchronize()
    HOST_US[(rank, world)] = pipe.host_us_per_frame
    return e0.elapsed_time(e1) / frames


def main():
    names = sys.argv[1:] or ["c32l4_fourier_1024x512", "c64l6_grid16_1024x512", "c64l6_grid16_time16_1024x512"]
    batch = int(os.environ.get("FVSRN_STRIPE_BATCH", "1"))
    for name in names:
        cfg = b.CONFIGS[name]
        keys = b.TIME_KEYS.get(name, 1)
        _, net = b.make_network(volnet_io, capi, cfg, "ReLU", keys)
        # Two world-1 baselines: the whole frame (a) frame by frame on one stream -- what r05 divided by -- and (b) in the SAME launch mode as the
        # ranks' shares (same frames per submit, two lanes).  The efficiency is against (b): like for like; (a) rides along as "..._vs_frame_by_frame".  An
        # efficiency above 1 against (b) would be a methodology error.
        full_fbf = frame_period(net, cfg, keys, 0, 1, pipelined=False)
        full = frame_period(net, cfg, keys, 0, 1, pipelined=True, batch=batch, frames=max(24, 6 * batch)) if batch > 1 else frame_period(net, cfg, keys, 0, 1, pipelined=True)
        full = min(full, full_fbf)  # (the better of the two is what one GPU can do with the frame)
        row = {"workload": name, "full_frame_ms": full, "full_frame_ms_frame_by_frame_one_stream": full_fbf, "time_keys": keys, "frames_per_submit": batch, "world": {}}
        ahead = os.environ.get("FVSRN_BENCH_BLEND_AHEAD", "0") == "1"
        row["blend_ahead"] = bool(ahead and keys > 1)
        row["working_grids"] = net.get_option("working_grids")
        row["emulated_gather"] = os.environ.get("FVSRN_STRIPE_EMULATE_GATHER")
        for world in [int(w) for w in os.environ.get("FVSRN_STRIPE_WORLDS", "2,4,8").split(",")]:
            # (FVSRN_STRIPE_RANK_ORDER=reverse: the ranks measured last to first -- is the first pipeline of a configuration slow, or rank 0's rows?)
            order = list(range(world))[::-1] if os.environ.get("FVSRN_STRIPE_RANK_ORDER") == "reverse" else list(range(world))
            measured = {r: frame_period(net, cfg, keys, r, world, ahead=ahead, batch=batch, frames=max(24, 6 * batch)) for r in order}
            periods = [measured[r] for r in range(world)]
            worst = max(periods)
            row["world"][str(world)] = {"slowest_rank_frame_period_ms": worst, "ideal_ms": full / world, "render_only_efficiency": full / world / worst,
                                        "render_only_efficiency_vs_frame_by_frame": full_fbf / world / worst,
                                        "rank_frame_period_ms": [round(p, 4) for p in periods], "stripe_rows": int(os.environ.get("FVSRN_STRIPE_ROWS", b.STRIPE)),
                                        "host_us_per_frame": max(HOST_US[(r, world)] for r in range(world))}
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
