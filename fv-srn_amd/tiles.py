"""
Multi-GPU image-tile partition: one process per GPU, round-robin row stripes, one RCCL all-gather per frame.

The reference has no distributed code at all; its entry this sits behind is
ImageEvaluatorSimple::render(W, H) (renderer/image_evaluator_simple.cpp:198-361), which renders a whole frame on one
device.  Pixels are independent, so the only exchange step of a frame is assembling it:

  rank r renders the rows {y : (y // stripe) % world == r}  (fvsrn_render_stripes, include/fvsrn.h) into a
  compact (8, rows_r, W) image; all compact images have the same size when H % (stripe*world) == 0, so ONE
  ``all_gather_into_tensor`` (RCCL over xGMI with backend "nccl", gloo on CPU) moves 8*rows*W*4 bytes per
  rank and a view/permute puts the stripes back in image order.

Round-robin stripes (instead of H/world contiguous blocks) balance empty-space rows against dense rows.

Hardware queues.  The pipeline of a rank runs on five streams (two render streams, the collective's, the library's copy stream for key
frames, the caller's) and ROCm maps all streams of a process onto GPU_MAX_HW_QUEUES hardware queues, FOUR by default: two streams that
share a queue run in submission order, and the collective of frame i ends up behind the render of frame i + 1 (r03: a rank's share at
world 8 at 80 % of frame / world instead of 98 %).  The variable is read when the HIP runtime starts, so a library cannot set it: export
GPU_MAX_HW_QUEUES=8 before the process starts (bench.py and launch_ranks() do that for the processes THEY start).  StripeRenderer measures
what it got -- one spinning wave on each of ITS render / collective streams (``hw_streams_concurrent``, fvsrn_probe_stream_concurrency) --,
warns when they do not all run side by side, and takes the persistent stripe launches that need them only then.

What travels (r05).  ``gather="all"`` puts the frame on every rank (one all-gather), ``gather="root"`` on rank 0 only (``dist.gather``: grouped
send / receive with RCCL, the form SURVEY 8(e) names for callers that need the frame once -- a viewer, a file writer).  ``payload="planes"`` moves the
reference's eight fp32 planes (32 B per pixel: what ImageEvaluatorSimple::render returns), ``payload="rgba8"`` runs IImageEvaluator::ExtractColor on
the rank's own rows first and moves packed RGBA8 words (4 B per pixel: what a viewer shows); only its DEPTH channel mode needs the other ranks -- the
depth range of the whole frame, one all-reduce of three floats (fvsrn_depth_range).  ``frames_per_submit=K`` renders K camera poses with ONE call into
the C library (fvsrn_render_stripes_batch) and moves them with ONE collective: the per-frame host work of the Python loop (scene update, two context
managers, events, the collective's enqueue -- tens of microseconds against a rank's 0.26 ms share of the headline frame at world 8) is paid once per K.

``StripeRenderer`` is the frame pipeline of one rank: double-buffered local / gathered images, the gather
of frame i on a communication stream while frame i + 1 renders, consecutive frames on two render streams so that the
tail of one launch overlaps the head of the next, time-dependent networks included (the C library keeps two working
grids, FVSRN_OPT_WORKING_GRIDS).  bench.py, __graft_entry__.smoke() and the tests drive this class; ``launch_ranks``
starts one process per rank when no launcher (torch.distributed.run) did.
"""
from __future__ import annotations

import contextlib
import os
import socket
import subprocess
import sys
import time
import time as _time
import warnings
from typing import Callable, List, Optional, Sequence

import torch

STRIPE = 16  # default stripe height: two 8-row pixel tiles of a wave


def owned_rows(height: int, stripe: int, rank: int, world: int) -> List[int]:
    """Image rows of `rank`, in the order they appear in its compact image."""
    rows = []
    for y0 in range(rank * stripe, height, stripe * world):
        rows.extend(range(y0, min(y0 + stripe, height)))
    return rows


def check_even_partition(height: int, stripe: int, world: int) -> None:
    if stripe <= 0 or stripe % 8 != 0:
        raise ValueError("stripe must be a positive multiple of 8 (the pixel tile of one wave)")
    if height % (stripe * world) != 0:
        raise ValueError("height %d is not a multiple of stripe*world = %d: ranks would own different row counts"
                         % (height, stripe * world))


def all_gather_frame(local: torch.Tensor, gathered: torch.Tensor = None, group=None) -> torch.Tensor:
    """local (8, rows, W) of every rank -> (world, 8, rows, W) on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    # concatenation along dim 0 is the layout every backend (RCCL, gloo) accepts
    dist.all_gather_into_tensor(gathered.view((world * local.shape[0],) + tuple(local.shape[1:])), local.contiguous(), group=group)
    return gathered


def assemble_rgba8(gathered: torch.Tensor, height: int, stripe: int) -> torch.Tensor:
    """(world, rows, W) packed RGBA8 stripe images -> (H, W) frame."""
    world, rows, width = gathered.shape
    check_even_partition(height, stripe, world)
    assert rows * world == height
    return gathered.view(world, rows // stripe, stripe, width).permute(1, 0, 2, 3).reshape(height, width)


def assemble(gathered: torch.Tensor, height: int, stripe: int) -> torch.Tensor:
    """(world, 8, rows, W) compact stripe images -> (1, 8, H, W) frame."""
    world, ch, rows, width = gathered.shape
    check_even_partition(height, stripe, world)
    assert rows * world == height
    return gathered.view(world, ch, rows // stripe, stripe, width).permute(1, 2, 0, 3, 4).reshape(1, ch, height, width)


class _HostEvent:
    """Stand-in for torch.cuda.Event on the CPU (gloo tests): host code is already ordered."""

    def record(self, stream=None):
        pass

    def elapsed_time(self, other):
        return 0.0


class _HostStream:
    def wait_event(self, e):
        pass

    def wait_stream(self, s):
        pass


class StripeRenderer:
    """Frame pipeline of one rank.

    net        capi.Network (device renders), or anything ``render`` understands
    scene_kw   keyword arguments of capi.Scene for the first frame (camera, step size, TF ...)
    render     None: the HIP path (capi.Scene.render for world == 1, capi.render_stripes otherwise).  A callable
               ``render(scene_kwargs, out, rank, world, stripe)`` that fills ``out`` ((1,8,H,W) for world == 1, compact
               (8, rows, W) otherwise) replaces it -- the CPU tests hand in the oracle so that the partition, the buffers and
               the collective are exercised with gloo.
    streams    optional (render stream 0, render stream 1, collective's stream) to use instead of new ones
    pipelined  consecutive frames alternate between two scenes on two render streams (default: world > 1).  At world == 1
               the same trick is worth +4 % (r01), but one launch at a time keeps the HIP-event duration of the kernel, the
               rocprofv3 trace and the frame period the same number, which is what bench.py reports there.
    gather     "all" (default): every rank ends up with the frame; "root": rank 0 only (``frame()`` is None elsewhere)
    payload    "planes" (default): the eight fp32 planes; "rgba8": ExtractColor on the rank's rows, packed RGBA8 words travel (``frame()`` is an (H, W)
               int32 image); channel_mode / use_tonemapping / max_exposure as in capi.extract_color; extract: the host-side stand-in of the CPU tests,
               ``extract(local (8,rows,W), channel_mode, use_tonemapping, max_exposure, depth_range3 or None) -> (rows, W) int32``, and
               depth_range: ``depth_range(local) -> (3,) float32 {-min, max, nan flag}``
    frames_per_submit   K >= 1: ``submit_batch`` renders K poses with one library call and one collective
    force_collective   world == 1 only: take the multi-GPU route anyway -- compact stripe image, ``all_gather_into_tensor`` on the
               collective's stream, ``assemble`` -- so that the RCCL path executes on a one-GPU box (tests, bench.py --force-collective)

    Attributes after construction: ``hw_streams_concurrent`` (measured, None on the CPU), ``persistent_stripes`` (whether the stripe
    launches are persistent: only when the process's streams run side by side), ``timings`` (per submitted frame with record=True:
    (render events, gather events))."""

    def __init__(self, net, width: int, height: int, scene_kw: dict, *, rank: int = 0, world: int = 1, stripe: int = STRIPE,
                 group=None, pipelined: Optional[bool] = None, device: str = "cuda", render: Optional[Callable] = None, streams=None,
                 force_collective: bool = False, gather: str = "all", payload: str = "planes", channel_mode: int = 3, use_tonemapping: bool = False,
                 max_exposure: float = 1.0, extract: Optional[Callable] = None, depth_range: Optional[Callable] = None, frames_per_submit: int = 1):
        self.net, self.W, self.H = net, int(width), int(height)
        self.rank, self.world, self.stripe, self.group = int(rank), int(world), int(stripe), group
        if force_collective and world != 1:
            raise ValueError("force_collective is the one-GPU test mode of the multi-GPU route (world == 1)")
        self.collective = world > 1 or bool(force_collective)  # compact stripes + all-gather + assemble
        if gather not in ("all", "root") or payload not in ("planes", "rgba8"):
            raise ValueError('gather is "all" or "root", payload "planes" or "rgba8"')
        self.gather, self.payload = gather, payload
        self.channel_mode, self.use_tonemapping, self.max_exposure = int(channel_mode), bool(use_tonemapping), float(max_exposure)
        self._extract_fn, self._depth_range_fn = extract, depth_range
        self.K = int(frames_per_submit)
        if self.K < 1:
            raise ValueError("frames_per_submit >= 1")
        self.host_seconds = 0.0  # wall time this rank's host spent inside submit / submit_batch (host_us_per_frame)
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        if not self.on_gpu and render is None:
            raise RuntimeError("StripeRenderer: the HIP path needs a GPU (there is no CPU fallback); pass render= for host-side tests")
        self._render_fn = render
        self.pipelined = (world > 1) if pipelined is None else bool(pipelined)
        self.kernel_events = []
        self.gather_events = []
        self.gather_frames = []
        self.frames_submitted = 0
        self.hw_streams_concurrent = None
        self.persistent_stripes = False
        if self.collective:
            check_even_partition(self.H, self.stripe, self.world)
        self.rows = self.H // self.world
        # Buffers: two render targets at world 1.  A rank of a multi-GPU frame rotates THREE local / gathered pairs: frame i + 2 then does not wait
        # for the collective of frame i, which may only get onto the chip once the render of frame i + 1 has ended (a collective whose
        # workgroups need more registers than a persistent render launch leaves free) -- with two pairs that wait is a bubble of one
        # collective per frame, with three the collective runs beside the start of the next render.
        nbuf = self.buffers = 3 if self.collective else 2
        K = self.K
        if not self.collective:
            if payload != "planes" or gather != "all":
                raise ValueError("gather / payload are options of the multi-GPU route (world > 1 or force_collective)")
            self.outs_k = [torch.zeros((K, 8, self.H, self.W), dtype=torch.float32, device=self.device) for _ in range(nbuf)]
            self.outs = [o[0:1] for o in self.outs_k]
        else:
            # [buffer][frame of the batch]: local = this rank's rows, gathered = every rank's (only rank 0 holds it with gather="root")
            self.local_k = [torch.zeros((K, 8, self.rows, self.W), dtype=torch.float32, device=self.device) for _ in range(nbuf)]
            self.local = [l[0] for l in self.local_k]
            holds = gather == "all" or self.rank == 0
            if payload == "planes":
                self.gathered_k = [torch.zeros((self.world, K, 8, self.rows, self.W), dtype=torch.float32, device=self.device) if holds else None for _ in range(nbuf)]
            else:
                self.send_k = [torch.zeros((K, self.rows, self.W), dtype=torch.int32, device=self.device) for _ in range(nbuf)]
                self.gathered_k = [torch.zeros((self.world, K, self.rows, self.W), dtype=torch.int32, device=self.device) if holds else None for _ in range(nbuf)]
                self.range3 = [torch.zeros(3, dtype=torch.float32, device=self.device) for _ in range(nbuf)]
            self.gathered = [g[:, 0] if g is not None else None for g in self.gathered_k]
        if self.on_gpu:
            from . import capi
            self._capi = capi
            self.scenes = [capi.Scene(**scene_kw) for _ in range(2 if self.pipelined else 1)]
            # streams = (render stream 0, render stream 1, collective's stream): a caller that builds several pipelines in one process hands
            # the same three to all of them (torch hands out streams from a pool, ROCm maps them onto its hardware queues by creation order)
            self.comm_stream = (streams[2] if streams else torch.cuda.Stream()) if self.collective else None

            self.blend_stream = None  # created with the first next_time: key-frame blends of the NEXT frame, beside the current render
            self.render_done = [torch.cuda.Event() for _ in range(nbuf)]
            self.gather_done = [torch.cuda.Event() for _ in range(nbuf)]
            if self.pipelined:
                # The first use of a network handle uploads its weight images and key frames on the stream of that call; the
                # library orders other streams behind it by events (include/fvsrn.h), this untimed frame only keeps the
                # uploads out of the first timed one.
                self._render(self.scenes[0], 0, None)
                torch.cuda.synchronize()
                self.render_streams = [streams[0], streams[1]] if streams else [torch.cuda.Stream(), torch.cuda.Stream()]
                for st in self.render_streams:
                    st.wait_stream(torch.cuda.current_stream())
            if self.collective and render is None:
                # Persistent stripe launches (97 - 100 % of frame / world at world 8 in the one-GPU emulation of r03, against 82 - 90 % for
                # bounded waves) need the collective's stream and the render streams on hardware queues of their own: measured on THESE
                # streams (fvsrn_probe_stream_concurrency), not assumed from the environment.
                mine = ([st.cuda_stream for st in self.render_streams] if self.pipelined else [torch.cuda.current_stream().cuda_stream]) + [self.comm_stream.cuda_stream]
                self.hw_streams_concurrent = capi.probe_stream_concurrency(mine, 2000)
                self.persistent_stripes = self.hw_streams_concurrent >= len(mine) - 0.4
                if not self.persistent_stripes:
                    warnings.warn("fvsrn StripeRenderer: its %d streams run only %.1f-wide (ROCm maps the streams of a process onto GPU_MAX_HW_QUEUES "
                                  "hardware queues, default 4): export GPU_MAX_HW_QUEUES=8 before the process starts; stripe launches stay in "
                                  "bounded waves" % (len(mine), self.hw_streams_concurrent), RuntimeWarning, stacklevel=2)
                else:
                    for sc in self.scenes:
                        if sc.get_option("persistent") < 0:  # (an explicit setting of the caller / the environment stays)
                            sc.set_option("persistent", 1)
                        # Batches (several poses per launch, consecutive batches on alternating lanes): the slots a persistent launch would leave free for
                        # the collective of the previous frame (1/16 of the chip) are taken by the other lane's launch at once, and a collective that waits
                        # for the end of a 2 ms batch costs latency (three buffer pairs), not throughput -- no reserve (r05: 88 -> 9x % of frame / world)
                        if self.K > 1 and sc.get_option("persistent_reserve") < 0:
                            sc.set_option("persistent_reserve", 0)
        else:
            self.scenes = [dict(scene_kw) for _ in range(2 if self.pipelined else 1)]
            self.comm_stream = _HostStream() if self.collective else None
            self.render_done = [_HostEvent() for _ in range(nbuf)]
            self.gather_done = [_HostEvent() for _ in range(nbuf)]
            self.render_streams = [_HostStream(), _HostStream()]

    # ---------------------------------------------------------------------------------------------------------------
    def _render(self, scene, b: int, stats, k: int = 0):
        out = self.local_k[b][k] if self.collective else self.outs_k[b][k:k + 1]
        if self._render_fn is not None:
            self._render_fn(scene, out, self.rank, self.world, self.stripe)
        elif not self.collective:
            scene.render(self.net, self.W, self.H, out=out, stats=stats)
        else:
            self._capi.render_stripes(scene, self.net, self.W, self.H, self.stripe, self.rank, self.world, out=out, stats=stats)

    def _stream_ctx(self, stream):
        return torch.cuda.stream(stream) if self.on_gpu else contextlib.nullcontext()

    def _extract(self, b: int, k: int, range3):
        """ExtractColor of frame k of buffer b's local rows -> send_k[b][k] (packed RGBA8), on the current stream."""
        if self._extract_fn is not None:
            self.send_k[b][k].copy_(torch.as_tensor(self._extract_fn(self.local_k[b][k], self.channel_mode, self.use_tonemapping, self.max_exposure, range3)))
        else:
            self._capi.extract_color_part(self.local_k[b][k], self.channel_mode, self.use_tonemapping, self.max_exposure, rgba8=True, depth_range3=range3,
                                          out=self.send_k[b][k])

    def _collect(self, b: int, n: int, record: bool, extracted: bool = False):
        """What follows the render of buffer b's n frames on the collective's stream: payload conversion, the collective, gather_done."""
        import torch.distributed as dist
        from .capi import CHANNEL_DEPTH
        with self._stream_ctx(self.comm_stream):  # gather(frame i) overlaps render(frame i + 1)
            self.comm_stream.wait_event(self.render_done[b])
            if record and self.on_gpu:
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
            if self.payload == "rgba8":
                if self.channel_mode == CHANNEL_DEPTH:
                    # the depth range of the WHOLE frame (iimage_evaluator.cpp:60-77): {-min, max, nan flag} of this rank's rows, merged by ONE all-reduce
                    for k in range(n):
                        r3 = self.range3[b]
                        if self._depth_range_fn is not None:
                            r3.copy_(torch.as_tensor(self._depth_range_fn(self.local_k[b][k])))
                        else:
                            r3.copy_(self._capi.depth_range(self.local_k[b][k]))
                        dist.all_reduce(r3, op=dist.ReduceOp.MAX, group=self.group)
                        self._extract(b, k, r3)
                elif not extracted:
                    for k in range(n):
                        self._extract(b, k, None)
                send, recv = self.send_k[b], self.gathered_k[b]
            else:
                send, recv = self.local_k[b], self.gathered_k[b]
            if self.gather == "all":
                dist.all_gather_into_tensor(recv.view((self.world * self.K,) + tuple(recv.shape[2:])), send, group=self.group)
            else:
                self._gather_to_root(send, recv)
            if record and self.on_gpu:
                g1.record()
                self.gather_events.append((g0, g1))
                self.gather_frames.append(n)  # frames this collective moved (a short last batch moves fewer than frames_per_submit)
            self.gather_done[b].record()

    def _gather_to_root(self, send, recv):
        """dist.gather to rank 0 (RCCL: one grouped send / receive; gloo: its gather on host tensors -- gloo has none for device tensors, the two-ranks-on-
        one-GPU test mode stages through the host)."""
        import torch.distributed as dist
        if self.on_gpu and dist.get_backend(self.group) == "gloo":
            host = send.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, parts, dst=0, group=self.group)
            if self.rank == 0:
                for r in range(self.world):
                    recv[r].copy_(parts[r])
            return
        dist.gather(send, [recv[r] for r in range(self.world)] if self.rank == 0 else None, dst=0, group=self.group)

    def submit(self, index: int, scene_kw: dict, *, time: Optional[float] = None, ensemble: int = 0, next_time: Optional[float] = None,
               stats=None, gather: bool = True, record: bool = False) -> int:
        """Enqueues frame `index`: scene update, optional time change, render of this rank's share, gather.  Returns the
        buffer index b (= index % self.buffers): ``frame(b)`` is this frame once ``finish()`` (or a later wait) has passed.
        next_time: the time of the frame that will be submitted next -- its key-frame blend is enqueued right behind this frame's
        render call on a high-priority side stream (fvsrn_network_prepare), so that it runs beside this render (into the working
        grid this render does not read) instead of between the two renders."""
        t_host = _time.perf_counter()
        b = index % self.buffers
        r = index & 1  # scene / render stream: consecutive frames alternate
        s = r if self.pipelined else 0
        if self.on_gpu:
            scene = self.scenes[s]
            scene.update(**scene_kw)
            stream = self.render_streams[r] if self.pipelined else torch.cuda.current_stream()
        else:
            scene = self.scenes[s]
            scene.clear()
            scene.update(scene_kw)
            stream = self.render_streams[r]
        prepared = getattr(self, "_prepared", None)
        if time is not None and not (prepared is not None and abs(prepared[0] - time) <= 1e-6 * max(1.0, abs(time)) and prepared[1] == ensemble):
            # key frames are resident (or streamed by the library's copy stream); this only marks the working grid dirty, the
            # blend into the grid the other frame in flight does NOT read is enqueued by the render call below
            self.net.set_time_and_ensemble(time, ensemble)
        self._prepared = None
        with self._stream_ctx(stream):
            if self.collective:
                stream.wait_event(self.gather_done[b])  # buffer b is free again
            if record and self.on_gpu:  # (behind the wait: the render's duration does not include the collective that frees buffer b)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self._render(scene, b, stats)
            if self.collective:
                self.render_done[b].record()
            if record and self.on_gpu:
                e1.record()
                self.kernel_events.append((e0, e1))
        if next_time is not None and self.on_gpu and self._render_fn is None:
            if self.blend_stream is None:
                self.blend_stream = torch.cuda.Stream(priority=-1)
                self.blend_stream.wait_stream(torch.cuda.current_stream())
            self.net.set_time_and_ensemble(next_time, ensemble)
            self.net.prepare(stream=self.blend_stream.cuda_stream)
            self._prepared = (next_time, ensemble)
        if self.collective and gather:
            self._collect(b, 1, record)
        self.frames_submitted += 1
        self.host_seconds += _time.perf_counter() - t_host
        return b

    def submit_batch(self, batch_index: int, scene_kws: Sequence[dict], *, times: Optional[Sequence[float]] = None, stats=None, gather: bool = True,
                     record: bool = False) -> int:
        """Enqueues len(scene_kws) <= frames_per_submit frames that differ in their camera (eye / right / up; everything else is taken from the
        first dict) and, optionally, their time: ONE call into the library renders them back to back (fvsrn_render_stripes_batch; the frames alternate
        between the two scenes / render streams of a pipelined renderer), ONE collective moves them.  Returns the buffer index b; ``frame(b, k)`` is
        frame k of the batch after ``finish()``."""
        t_host = _time.perf_counter()
        n = len(scene_kws)
        if not 1 <= n <= self.K:
            raise ValueError("a batch holds 1 .. frames_per_submit frames")
        b = batch_index % self.buffers
        if self._render_fn is not None or not self.on_gpu:  # host-side stand-in (CPU tests): frame by frame
            for k, kw in enumerate(scene_kws):
                scene = self.scenes[0]
                if self.on_gpu:
                    scene.update(**kw)
                else:
                    scene.clear()
                    scene.update(kw)
                if times is not None:
                    self.net.set_time_and_ensemble(times[k], 0)
                self._render(scene, b, stats, k)
            if self.collective:
                self.render_done[b].record()
        else:
            import numpy as np
            # one lane per batch, consecutive batches on alternating lanes (scene + render stream): the library renders the batch as launches of up to
            # eight poses each (a work unit is (frame, tile)), and the tail of one batch's last launch overlaps the head of the next batch's first
            # (frames with their own times are one launch each -- every frame blends its own working grid --: those batches take BOTH lanes, frame f on
            # lane f % 2, like the frame-by-frame pipeline: r05, configs[4] at world 8: 91.7 % on one lane, 98 % frame by frame on two)
            r = batch_index & 1 if self.pipelined else 0
            if times is not None and self.pipelined:
                lanes, streams = list(self.scenes), list(self.render_streams)
            else:
                lanes = [self.scenes[r]]
                streams = [self.render_streams[r]] if self.pipelined else [torch.cuda.current_stream()]
            for sc in lanes:
                sc.update(**scene_kws[0])
            cams = np.stack([np.concatenate([np.asarray(kw[key], np.float32).reshape(3) for key in ("eye", "right", "up")]) for kw in scene_kws])
            if self.collective:
                for st in streams:
                    st.wait_event(self.gather_done[b])  # buffer b is free again
            if record:  # (behind the wait: the render's duration does not include the collective that frees buffer b)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(streams[0])
            in_call = self.collective and self.payload == "rgba8" and self.channel_mode == self._capi.CHANNEL_COLOR
            # (world == 1 -- one GPU, or force_collective's one-rank group: a whole frame has the layout of a one-rank stripe image)
            self._capi.render_stripes_batch(lanes, [st.cuda_stream for st in streams], self.net, self.W, self.H, self.stripe, self.rank, self.world, cams,
                                            times=times, out=self.local_k[b] if self.collective else self.outs_k[b], rgba8=self.send_k[b] if in_call else None,
                                            use_tonemapping=self.use_tonemapping, max_exposure=self.max_exposure, stats=stats)
            for st in streams[1:]:  # the batch is done when every lane is
                streams[0].wait_stream(st)
            if self.collective:
                self.render_done[b].record(streams[0])
            if record:
                e1.record(streams[0])
                self.kernel_events.append((e0, e1))
        if self.collective and gather:
            self._collect(b, n, record, extracted=self.on_gpu and self._render_fn is None and self.payload == "rgba8" and self.channel_mode == 3)
        self.frames_submitted += n
        self.host_seconds += _time.perf_counter() - t_host
        return b

    @property
    def host_us_per_frame(self) -> float:
        return 1e6 * self.host_seconds / max(1, self.frames_submitted)

    def finish(self) -> None:
        """The current stream waits for everything submitted (renders on both streams, gathers)."""
        if not self.on_gpu:
            return
        cur = torch.cuda.current_stream()
        if self.pipelined:
            for st in self.render_streams:
                cur.wait_stream(st)
        if self.collective:
            cur.wait_stream(self.comm_stream)
        if self.blend_stream is not None:
            cur.wait_stream(self.blend_stream)

    def frame(self, b: int = 0, k: int = 0):
        """Frame k of buffer b.  payload "planes": the (1, 8, H, W) image -- the render target itself on one GPU, the gathered stripes put back in image
        order (a view + one permuting copy) otherwise; payload "rgba8": the (H, W) int32 image of packed RGBA8 words.  None on the ranks that do not hold
        the frame (gather="root", rank > 0)."""
        if not self.collective:
            return self.outs_k[b][k:k + 1]
        g = self.gathered_k[b]
        if g is None:
            return None
        if self.payload == "rgba8":
            return assemble_rgba8(g[:, k], self.H, self.stripe)
        return assemble(g[:, k], self.H, self.stripe)


def frames_match(full: torch.Tensor, gathered: torch.Tensor, tol_image: float = 2e-4, tol_depth: float = 3e-2) -> bool:
    """A frame assembled from stripes against the single-GPU render of the same scene.  Same samples, but a rank's stripes
    are a small launch and may be rendered in depth segments (re-associated sums, other restart points of the feature
    rotation).  Since r04 a restart derives the features from the fp32 position (srn_device.hpp, fourier_features hilo), so
    where the rotation restarts no longer shows beyond fp32 rounding: measured <= 4.3e-5 between 1 / 2 / 4 / 8 segments and
    stripes on the bench networks (tools/dev/segment_diff.py; r03: up to the 3e-3 of the parity tests) -- the tolerance is
    2e-4.  Depth (NaN where alpha == 0) only on pixels that are not within rounding of empty."""
    solid = (full[0, 3] > 1e-4) | (gathered[0, 3] > 1e-4)
    if float((full[0, :7] - gathered[0, :7]).abs().max()) >= tol_image:
        return False
    if not torch.equal(torch.isnan(full[0, 7])[solid], torch.isnan(gathered[0, 7])[solid]):
        return False
    if not bool(solid.any()):
        return True
    d = (torch.nan_to_num(full[0, 7], nan=0.0) - torch.nan_to_num(gathered[0, 7], nan=0.0))[solid]
    return float(d.abs().max()) < tol_depth


def launch_ranks(n: int, argv: Sequence[str], script: Optional[str] = None, child: Optional[Sequence[str]] = None,
                 capture: bool = False, timeout: float = 1800.0):
    """Starts `n` rank processes of `script` (fresh interpreters, one per GPU, rendezvous on 127.0.0.1), relays rank 0's
    stdout and returns the worst exit code (with capture=True: (code, rank 0's stdout) and nothing is printed).  This process
    never initialises a GPU (torch.cuda.device_count() does not, on this image).  Fewer than `n` visible GPUs is an error, not
    a silent 1-GPU run -- unless FVSRN_BENCH_BACKEND=gloo, the test mode in which all ranks share the visible GPU(s).
    child: command prefix of a rank process (tests substitute a stand-in); default = `script` under this interpreter."""
    backend = os.environ.get("FVSRN_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print("--gpus %d: only %d GPU(s) visible, one per rank is required" % (n, have), file=sys.stderr)
        return (2, "") if capture else 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = list(child) if child else [sys.executable, os.path.abspath(script)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "8"))  # (module docstring: read when HIP starts in the child)
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    try:
        out0, _ = procs[0].communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        for p in procs:  # exactly the processes started here
            p.kill()
        out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:  # rank 0 is gone and this one hangs in a collective: end exactly this process
            p.kill()
            rcs.append(p.wait())
    bad = [rc for rc in rcs if rc != 0]
    code = 0 if not bad else (bad[0] if bad[0] > 0 else 1)
    if capture:
        return code, out0.decode()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return code
