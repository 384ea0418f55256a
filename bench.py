#!/usr/bin/env python3
"""
bench.py -- headline benchmark of the MI355X fV-SRN renderer.

A "step" = one frame of the fused SRN-MLP + DVR ray-stepping hot path (fvsrn_render[_stripes]) on
synthetic inputs already resident in HBM:
  network  32-wide x 4-layer fp16 SRN, Fourier-only encoding (F=14, NeRF block-identity matrix),
           seeded nn.Linear-style init, output density:direct              (BASELINE.json configs[1] net)
  frame    1024 x 1024, world step 1/512 (512 steps across the unit box), box [-0.5,0.5]^3,
           CameraOnASphere(Ym, pitch 0.4, distance 1.6, fovY 45 deg), yaw advancing 2*pi/64 per frame,
           Identity TF (absorption 10, emission 1), Beer-Lambert, early-out OFF (every sample evaluated)
  activation  ReLU by default: the activation the reference's own timing harness runs
           (applications/volnet/eval_NetworkConfigsGrid.py:35, activationX = ["ReLU"]); the paper's SnakeAlt
           twin is timed beside it and reported in "snakealt_twin".
metric: SRN samples/s, counting lane-exact EVALUATED samples (the loop bound of
renderer_ray_evaluation_stepping_dvr.cuh:84-90 summed over all rays), read from the kernel's own counter.

N > 1 (`python bench.py --gpus N` starts the N rank processes itself; under torch.distributed.run it takes the launcher's
RANK / WORLD_SIZE instead): the SAME frame is split into round-robin 16-row stripes, one
process per GPU, each rank renders its stripes and one RCCL all-gather assembles the frame ("strong" scaling;
the frame pipeline is fv-srn_amd/tiles.py StripeRenderer: the gather of frame i overlaps the render of frame i+1 on a side stream,
consecutive frames alternate between two render streams so that the tail of one launch overlaps the start of the next --
time-dependent networks included, the library blends into the working grid the frame in flight does not read).

Prints ONE JSON line (see the contract in the task description) with "roofline" and "cpu_baseline".
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# before the HIP runtime starts (fv-srn_amd/__init__.py): the frame pipeline's streams need more than ROCm's default four hardware queues
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
NOMINAL_CLOCK_HZ = 2.4e9
NUM_SIMDS = 256 * 4
# transcendental instructions (v_cos_f32 / v_sin_f32 / v_exp_f32 ...): 8 issue cycles per wave64 instruction = 8 lanes per clock and SIMD
# (MI355X_MICROARCH.md, constants table "vector-instruction ISSUE cost"; measured here: 8.3 - 8.4, profiles/r01/microbench_issue_model.md).
# SURVEY 8(d) prices them at a quarter of the plain VALU rate (4 lanes per clock); "frac_at_quarter_rate" restates the line for that figure.
TRANS_LANES_PER_CLOCK = 8.0
# rows per stripe of a multi-GPU frame (round-robin over the ranks).  r05: 8 (one row of pixel tiles) instead of 16 -- the ranks' shares of the headline frame at world 8 differ by
# 2.7 % instead of 5.5 % (tools/stripe_efficiency.py, profiles/r05/stripe_efficiency_8_rows_r05.jsonl: 98.6 against 96.6 % of frame / 8; configs[2] 96.6 against 94.9 %)
STRIPE = 8

CONFIGS = {
    # name: (C, layers, grid(ch,res)|None, width, height, steps)
    "c32l4_fourier_512x256": (32, 4, None, 512, 512, 256),       # BASELINE.json configs[1]
    "c32l4_fourier_1024x512": (32, 4, None, 1024, 1024, 512),    # the size the metric is quoted on
    "c32l4_grid16_1024x512": (32, 4, (16, 16), 1024, 1024, 512),  # configs[2]
    "c64l6_grid16_1024x512": (64, 6, (16, 32), 1024, 1024, 512),  # configs[3]
    "c32l4_grid16r32_1024x512": (32, 4, (16, 32), 1024, 1024, 512),  # the paper's fV-SRN size: 32x4 network, 32^3 x 16 grid
    # configs[4]: time-dependent latent grids, 16 key frames, the time advances 0.25 key frames per rendered frame
    "c64l6_grid16_time16_1024x512": (64, 6, (16, 32), 1024, 1024, 512),
}
TIME_KEYS = {"c64l6_grid16_time16_1024x512": 16}


# Variants of a workload off the headline's happy path (every kernel family gets a bench line and a profile), set by main() from the command
# line: camera distance (a close-up trips the footprint rule of the cell table), gradient mode + Phong shading (render_shaded_kernel / render_adjoint_kernel).
VARIANT = {"distance": 1.6, "gradient_mode": 0}


def build_scene_kwargs(capi, yaw, stepsize, early_out):
    eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, yaw, VARIANT["distance"])
    kw = dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=stepsize,
              early_out=early_out, tf_kind=capi.TF_IDENTITY, tf_scale_absorption=10.0, tf_scale_emission=1.0)
    if VARIANT["gradient_mode"]:  # the shading of tools/bench_shaded.py: Phong, point light at the camera, finite-difference step 1 / 256
        kw.update(gradient_mode=VARIANT["gradient_mode"], finite_differences_stepsize=1 / 256,
                  brdf=dict(enable_phong=True, ambient=0.2, specular=0.4, magnitude_center=0.6, magnitude_radius=0.5, specular_exponent=8, light_type=0,
                            light=tuple(float(v) for v in eye)))
    return kw


GRID_ENCODINGS = {"float": 0, "byte_linear": 1, "byte_gaussian": 2}  # volnet_io.ENC_*


def bench_network(C, layers, grid, activation, time_keys=1, encoding="float"):
    """The synthetic network of a workload: seed 1234, nn.Linear-style init, NeRF ladder, density:direct,
    latent grid randn * 0.01.  tests/test_gpu_parity.py renders the same networks against the oracle."""
    from fvsrn_amd import synthetic
    return synthetic.random_network(C=C, layers=layers, activation=activation, param=1.0, output_mode="density:direct",
                                    grid=grid, seed=1234, box_min=(-0.5, -0.5, -0.5), grid_scale=0.01, time_grids=time_keys,
                                    encoding=GRID_ENCODINGS[encoding])


def make_network(volnet_io, capi, cfg, activation, time_keys=1):
    C, layers, grid, *_ = cfg
    vn = bench_network(C, layers, grid, activation, time_keys, VARIANT.get("encoding", "float"))
    return vn, capi.Network.from_volnet(volnet_io.save_volnet(vn))


def bench_arrays(C, layers, grid, time_keys=1):
    """The fp32 arrays bench_network() is built from (the CPU baseline times exactly this network)."""
    from fvsrn_amd import synthetic
    return synthetic.random_arrays(C=C, layers=layers, output_mode="density:direct", grid=grid, seed=1234, grid_scale=0.01, time_grids=time_keys)


class Runner:
    """The benchmark's frame sequence (rotating camera, advancing time) on fv-srn_amd/tiles.py StripeRenderer: whole frames on
    one GPU, this rank's stripes + all-gather on several.  All multi-GPU logic lives in the package."""

    def __init__(self, capi, net, cfg, rank, world, early_out, time_keys=1, scene_options=None, force_collective=False, gather="all", payload="planes",
                 frames_per_submit=1, device="cuda", render=None, extract=None):
        from fvsrn_amd import tiles
        self.capi, self.net, self.rank, self.world = capi, net, rank, world
        self.time_keys = time_keys
        _, _, _, self.W, self.H, steps = cfg
        self.stepsize = 1.0 / steps
        self.early_out = early_out
        # N > 1: two frames in flight on two streams (time-dependent networks too: FVSRN_OPT_WORKING_GRIDS).  At N = 1 the same
        # trick is worth +4 % (r01: 152.7 -> 159.0 Gsamples/s, FVSRN_BENCH_PIPELINE=1), but the default keeps one launch at a time
        # there so that the HIP-event duration of the kernel, the rocprofv3 kernel trace and the frame period are the same number.
        pipe = os.environ.get("FVSRN_BENCH_PIPELINE")
        pipelined = pipe == "1" if pipe is not None else world > 1
        self.pipeline = tiles.StripeRenderer(net, self.W, self.H, build_scene_kwargs(capi, 0.0, self.stepsize, early_out), rank=rank,
                                             world=world, stripe=STRIPE, pipelined=pipelined, force_collective=force_collective,
                                             frames_per_submit=frames_per_submit, device=device, render=render, extract=extract,
                                             **(dict(gather=gather, payload=payload) if world > 1 or force_collective else {}))
        self.K = frames_per_submit
        for sc in self.pipeline.scenes:
            for k, v in (scene_options or {}).items():
                sc.set_option(k, v)
        self.pipelined = self.pipeline.pipelined
        # (device / render / extract: the host-side stand-in of tests/test_bench_record.py -- the partition, the buffers, the collectives and this
        # file's record with gloo on the CPU; the product path is device="cuda", render=None)
        self.stats = torch.zeros(2, dtype=torch.int64, device=device)

    @property
    def kernel_events(self):
        return self.pipeline.kernel_events

    def frame(self, index, record=False, gather=True):
        yaw = 2 * math.pi * (index % 64) / 64
        t = (0.25 * index) % (self.time_keys - 1) if self.time_keys > 1 else None
        # (the next frame's time is known: its key-frame blend is enqueued beside this frame's render, tiles.StripeRenderer.submit)
        tn = (0.25 * (index + 1)) % (self.time_keys - 1) if self.time_keys > 1 and self.pipelined and os.environ.get("FVSRN_BENCH_BLEND_AHEAD", "0") == "1" else None
        return self.pipeline.submit(index, build_scene_kwargs(self.capi, yaw, self.stepsize, self.early_out), time=t, next_time=tn, stats=self.stats,
                                    gather=gather, record=record)

    def frames(self, first, count, record=False, gather=True):
        """frames first .. first + count - 1: one by one, or (--frames-per-submit K) K per call into the library and per collective"""
        if self.K == 1:
            for i in range(first, first + count):
                self.frame(i, record=record, gather=gather)
            return
        for j in range(first, first + count, self.K):
            idx = list(range(j, min(j + self.K, first + count)))
            kws = [build_scene_kwargs(self.capi, 2 * math.pi * (i % 64) / 64, self.stepsize, self.early_out) for i in idx]
            times = [(0.25 * i) % (self.time_keys - 1) for i in idx] if self.time_keys > 1 else None
            self.pipeline.submit_batch(j // self.K, kws, times=times, stats=self.stats, gather=gather, record=record)

    def where(self, index):
        """(buffer, frame of the batch) that holds frame `index`"""
        return ((index // self.K) % self.pipeline.buffers, index % self.K) if self.K > 1 else (index % self.pipeline.buffers, 0)

    def finish(self):
        self.pipeline.finish()

    def assemble(self, b=0, k=0):
        return self.pipeline.frame(b, k)


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def timed_run(runner, steps, warmup, distributed, spinup_ms=0.0):
    import torch.distributed as dist
    # Clock spin-up (untimed, before the W warm-up steps): the shader clock of an idle MI355X needs a few hundred ms of load to
    # settle (r01: the same 32 timed frames read 1.5 % lower after 4 warm-up frames than after 150, 8 timed frames 5 % lower), and
    # at N GPUs the timed region shrinks to 32 x 0.3 ms.  Every rank spins for the same wall time, rendering only (the number of
    # frames differs between ranks, so no collective is called here).
    if distributed and spinup_ms > 0:  # one untimed frame with its gather on every rank: RCCL sets its communicator up lazily
        runner.frames(0, 1)
        runner.finish()
        _sync()
        dist.barrier()
    t_end = time.perf_counter() + 1e-3 * spinup_ms
    i = 0
    while time.perf_counter() < t_end:
        runner.frames(i, 8, gather=False)
        i += 8
        runner.finish()
        _sync()
    # (batches are aligned to frame indices: the warm-up is rounded up to whole batches so that the timed frames start on one)
    warmup = -(-warmup // runner.K) * runner.K
    runner.frames(0, warmup)
    runner.finish()
    _sync()
    runner.stats.zero_()
    runner.kernel_events.clear()
    runner.pipeline.gather_events.clear()
    runner.pipeline.gather_frames.clear()
    runner.pipeline.host_seconds, runner.pipeline.frames_submitted = 0.0, 0
    if distributed:
        dist.barrier()
    _sync()
    t0 = time.perf_counter()
    runner.frames(warmup, steps, record=True)
    runner.finish()
    _sync()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    # (one event pair per submit = per launch: a batch of K frames is K frames' worth of work)
    per = [min(runner.K, steps - j) for j in range(0, steps, runner.K)] if runner.K > 1 else [1] * len(runner.kernel_events)
    runner.launch_ms = [a.elapsed_time(b) for a, b in runner.kernel_events]
    runner.frames_per_launch = per
    kernel_ms = [t / n for t, n in zip(runner.launch_ms, per)]
    st = runner.stats.cpu().numpy().astype(np.int64)
    runner.last_frame = warmup + steps - 1
    return dt, kernel_ms, int(st[0]), int(st[1])


def cpu_baseline(cfg, activation):
    """The reference's PyTorch path (port: oracle/torch_port.py) on the host cores, bounded sample; the network is
    bench_network()'s (same seed, same arrays, fp32 like the reference's PyTorch model)."""
    from oracle import torch_port
    from fvsrn_amd import capi
    C, layers, grid, *_ = cfg
    a = bench_arrays(C, layers, grid)
    net = torch_port.TorchSRN(a["B"], a["weights"], a["biases"], activation, 1.0, "density:direct", a["grids"][0][None] if a["grids"] else None)
    eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.0, 1.6)
    # torch's intra-op pool degrades badly beyond a few dozen threads on these small GEMMs (256 threads: 1.3e4
    # samples/s, measured r01) -- use at most 32 host threads and say so in "cores"
    cores = min(os.cpu_count() or 1, 32)
    W = H = 512
    steps = 256
    r = torch_port.time_cpu_baseline(net, eye, right, up, float(np.deg2rad(45.0)), (-0.5, -0.5, -0.5), (1, 1, 1), width=W,
                                     height=H, stepsize=1.0 / steps, tf_identity=(10.0, 1.0), threads=cores, budget_s=15.0)
    return {"value": r["value"], "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "PyTorch-CPU fp32 port of SceneRepresentationNetwork.forward + Raytracing._full_trace_forward, "
                      "the bench network (seed 1234) and camera, %dx%d rays at step 1/%d, the first %d network samples (%.1f s of the step loop)"
                      % (W, H, steps, r["samples"], r["seconds"])}


def grid_volume_bench(res, size, interpolation, frames, with_cpu):
    """BASELINE.json configs[0] on the GPU: DVR of a dense grid volume (VolumeInterpolationGrid) next to the C restatement on the
    host cores (the cpu_baseline leg).  The reference's example-volume.cvol is not in the repository snapshot
    (.MISSING_LARGE_BLOBS): a synthetic res^3 float volume stands in.  Prints one JSON line."""
    import fvsrn_amd  # noqa: F401
    from fvsrn_amd import capi
    n = res
    ax = np.linspace(-1, 1, n, dtype=np.float32)
    x, y, z = np.meshgrid(ax, ax, ax, indexing="ij")
    data = np.clip(np.exp(-3 * (x * x + y * y + z * z)) + 0.1 * np.sin(9 * x) * np.cos(7 * y) * np.sin(5 * z), 0, 1).astype(np.float32)
    vol = capi.Volume.from_array(data, (-0.5, -0.5, -0.5), (1, 1, 1))
    stepsize = 1.0 / n

    def kwargs(yaw):
        eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, yaw, 1.6)
        return dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=stepsize, early_out=False,
                    tf_kind=capi.TF_IDENTITY, tf_scale_absorption=10.0, tf_scale_emission=1.0)

    scene = capi.Scene(**kwargs(0.0))
    out = torch.zeros((1, 8, size, size), dtype=torch.float32, device="cuda")
    stats = torch.zeros(2, dtype=torch.int64, device="cuda")
    t_end = time.perf_counter() + 0.25  # clock spin-up, see timed_run
    while time.perf_counter() < t_end:
        vol.render(scene, size, size, interpolation, out=out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(frames):
        scene.update(**kwargs(2 * np.pi * i / 64))
        vol.render(scene, size, size, interpolation, out=out, stats=stats)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / frames
    samples = int(stats.cpu()[0]) / frames
    taps = {0: 1, 1: 8, 2: 64}[interpolation]
    res_ = {"workload": "grid_dvr: %d^3 fp32 volume, %dx%d, step 1/%d, interpolation %d, Identity TF, early-out off" % (n, size, size, n, interpolation),
            "ms_per_frame": ms, "frames_per_s": 1e3 / ms, "samples_per_s": samples / (ms * 1e-3), "samples_per_frame": samples,
            "voxel_reads_per_s": taps * samples / (ms * 1e-3), "voxel_read_GBps_algorithmic": 4 * taps * samples / (ms * 1e-3) / 1e9,
            "volume_MB": data.nbytes / 1e6}
    if with_cpu:
        from oracle import oracle
        ov = oracle.OracleVolume(data, (-0.5, -0.5, -0.5), (1, 1, 1), interpolation, oracle.VOLUME_SOURCE_TEXTURE)
        kw = kwargs(0.0)
        kw.pop("tf_kind")
        sc = oracle.OracleScene(tf_kind=oracle.TF_IDENTITY, **kw)
        t0 = time.perf_counter()
        ref, cnt = ov.render(sc, size, size)
        dt = time.perf_counter() - t0
        res_["cpu_baseline"] = {"value": cnt / dt, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                                "sample": "oracle_render_volume (C restatement, OpenMP) of the same frame, %.2f s" % dt}
        scene.update(**kwargs(0.0))
        img = vol.render(scene, size, size, interpolation)[0].cpu().numpy()
        res_["max_abs_diff_vs_cpu"] = float(np.abs(img[:4] - ref[:4]).max())
    print(json.dumps(res_))


def pmc_counters(workload):
    """Per-launch averages of the committed rocprofv3 PMC summary of this workload (profiles/r*/<workload>_*_pmc.csv, the newest round), {} if
    there is none.  The counters cannot be read inside the timed run; tools/pmc_profile.sh collects them in separate --pmc passes."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*", workload + "_r[0-9][0-9]_pmc.csv")))
    if not files:
        return {}
    per_kernel = {}
    with open(files[-1]) as f:
        for row in csv.reader(l for l in f if not l.startswith("#")):
            if len(row) == 5 and row[0].startswith("pmc"):
                per_kernel.setdefault(row[1], {})[row[2]] = float(row[4])
    if not per_kernel:
        return {}
    # (a profile may hold several kernels -- evaluate_points' deferred pass, the composite of depth segments: the one the time went to)
    kernel = max(per_kernel, key=lambda k: per_kernel[k].get("kernel_ms_avg_under_pmc", 0.0) * 1.0)
    vals = dict(per_kernel[kernel])
    vals["file"] = os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__)))
    vals["kernel"] = kernel
    return vals


def transcendentals_per_sample(info, activation, rotation_resync):
    """SURVEY 8(d) "secondary bounds": 2F for the Fourier features + (L - 1) C for a periodic activation (Sine / Snake / SnakeAlt: one v_sin / v_cos
    per channel and hidden layer; Sigmoid: exp + rcp), minus what the feature rotation of the 32-wide Fourier-only kernels removes (exact features
    only every `rotation_resync` steps, fvsrn_scene_last_render_info)."""
    per_act = {"ReLU": 0, "Sine": 1, "Snake": 1, "SnakeAlt": 1, "Sigmoid": 2}[activation]
    fourier = 2.0 * info.num_fourier / (rotation_resync if rotation_resync > 0 else 1)
    return fourier + per_act * (info.num_layers - 1) * info.hidden_channels


# Issue cost of a wave instruction on a SIMD's single vector issue port, in cycles (measured on MI355X: profiles/r01/microbench_issue_model.md,
# profiles/r02/microbench_issue_model_r02.md; MI355X_MICROARCH.md constants table): an MFMA holds the port for 8 of its 32 matrix-pipe cycles, a
# transcendental for 8.4, a convert / packed instruction for 4.4, any other vector instruction for 2.6 - 2.9.
ISSUE_CYCLES = {"mfma": 8.0, "trans": 8.4, "cvt": 4.4, "other": 2.75}


def issue_roofline(pmc):
    """The vector issue port as a roofline: sum over instruction classes of (wave instructions per launch x issue cycles) against the SIMD cycles of the
    launch (SIMDs x shader clock x duration, clock and duration from the same PMC profile).  Counters: SQ_INSTS_VALU (all vector instructions, MFMA
    included), SQ_INSTS_MFMA, SQ_INSTS_VALU_TRANS_F32, SQ_INSTS_VALU_CVT; "other" is the rest priced as plain fp32 -- packed fp32 instructions (4.4
    cycles) are in it, so the fraction is a LOWER bound.  None without a committed PMC profile of the workload."""
    need = ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "GRBM_GUI_ACTIVE", "kernel_ms_avg_under_pmc")
    if not all(k in pmc for k in need):
        return None
    n = {"mfma": pmc["SQ_INSTS_MFMA"], "trans": pmc["SQ_INSTS_VALU_TRANS_F32"], "cvt": pmc["SQ_INSTS_VALU_CVT"]}
    n["other"] = max(0.0, pmc["SQ_INSTS_VALU"] - sum(n.values()))
    used = {k: n[k] * ISSUE_CYCLES[k] for k in n}
    simd_cycles = NUM_SIMDS * pmc["GRBM_GUI_ACTIVE"] / 8.0  # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = shader cycles of the launch
    return {"bound": "issue", "frac": sum(used.values()) / simd_cycles, "issue_cycles_per_launch": used, "insts_per_launch": n,
            "simd_cycles_per_launch": simd_cycles, "cycles_per_inst": ISSUE_CYCLES, "note": "packed fp32 counted as plain fp32: a lower bound", "from": pmc["file"]}


def roofline(info, activation, evaluated_per_launch, kernel_s, workload_tag, rotation_resync, world=1, algorithmic_bytes=None):
    """The roofline object of one bench line.  ReLU: the MFMA roofline (algorithmic FLOP).  Periodic activations: the
    transcendental unit -- achieved = evaluated samples x transcendentals per sample / kernel time, peak = SIMDs x lanes per clock x
    the shader clock, which is taken from GRBM_GUI_ACTIVE of the committed PMC profile of this workload (summed over the 8 XCDs) where
    there is one, else the nominal 2.4 GHz; the MFMA fraction rides along as "mfma_frac"."""
    pmc = pmc_counters(workload_tag) if world == 1 else {}
    flops = info.flops_per_sample * evaluated_per_launch
    mfma_achieved = flops / kernel_s / 1e12
    traffic = 1024.0 * (pmc["WRITE_SIZE"] + 2.0 * pmc["FETCH_SIZE"]) if "WRITE_SIZE" in pmc and "FETCH_SIZE" in pmc else None
    common = {"traffic": traffic, "flops_per_sample": info.flops_per_sample, "mfma_flops_per_sample": info.mfma_flops_per_sample,
              "kernel_ms_avg": 1e3 * kernel_s, "mfma_frac": mfma_achieved / MFMA_F16_PEAK_TFLOPS}
    if traffic is not None and algorithmic_bytes:
        # HBM bytes the launch must move by construction (the eight output planes, written once) against what the counters saw; a ratio well above 1 is
        # re-read data -- for the 64-wide latent-grid frames the 30 MB cell table, which misses the 4 MiB L2 of an XCD on every pass
        common["algorithmic_bytes"] = algorithmic_bytes
        common["traffic_ratio"] = traffic / algorithmic_bytes
    issue = issue_roofline(pmc)
    if issue:
        common["issue"] = issue
    if activation == "ReLU":
        return dict({"bound": "mfma", "achieved": mfma_achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mfma_achieved / MFMA_F16_PEAK_TFLOPS}, **common)
    tps = transcendentals_per_sample(info, activation, rotation_resync)
    clock, clock_from = NOMINAL_CLOCK_HZ, "nominal"
    if "GRBM_GUI_ACTIVE" in pmc and "kernel_ms_avg_under_pmc" in pmc:
        clock, clock_from = pmc["GRBM_GUI_ACTIVE"] / 8.0 / (1e-3 * pmc["kernel_ms_avg_under_pmc"]), pmc["file"]
    achieved = tps * evaluated_per_launch / kernel_s / 1e12               # T lane-operations / s
    peak = NUM_SIMDS * TRANS_LANES_PER_CLOCK * clock / 1e12
    r = dict({"bound": "valu_trans", "achieved": achieved, "peak": peak, "unit": "Ttrans/s", "frac": achieved / peak,
              "frac_at_quarter_rate": achieved / (peak * 4.0 / TRANS_LANES_PER_CLOCK), "transcendentals_per_sample": tps,
              "lanes_per_clock_and_simd": TRANS_LANES_PER_CLOCK, "clock_ghz": clock / 1e9, "clock_from": clock_from}, **common)
    if "SQ_INSTS_VALU_TRANS_F32" in pmc:
        r["pmc_trans_insts_per_launch"] = pmc["SQ_INSTS_VALU_TRANS_F32"]   # wave instructions; x 64 lanes / evaluated samples ~ tps
    return r


# ---- the multi-GPU record (one invocation, every piece of evidence) -----------------------------------------------------------------
# north_star: "frames are partitioned by image tile across the 8 GPUs of one node with RCCL gather over xGMI" -- the PRIMARY timed region of --gpus N moves the
# eight fp32 planes of ImageEvaluatorSimple::render to rank 0; the other three (gather, payload) combinations are timed behind it in the same process group, on
# the same frames, and ride along as "variants".
PRIMARY_GATHER, PRIMARY_PAYLOAD = "root", "planes"
COMBOS = [("root", "planes"), ("all", "planes"), ("root", "rgba8"), ("all", "rgba8")]


def init_group_or_exit(backend, device_index=None):
    """The process group of this run from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT), and a first collective on it (RCCL brings
    its communicator up lazily: a dead peer or an unreachable port shows there).  Any failure: the reason on stderr and exit code 3 -- never a silent
    one-GPU measurement.  FVSRN_BENCH_INIT_TIMEOUT_S bounds the wait (default: torch's)."""
    import datetime
    import torch.distributed as dist
    kw = {}
    if os.environ.get("FVSRN_BENCH_INIT_TIMEOUT_S"):
        kw["timeout"] = datetime.timedelta(seconds=float(os.environ["FVSRN_BENCH_INIT_TIMEOUT_S"]))
    try:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index), **kw)
        else:
            dist.init_process_group(backend=backend, **kw)
        t = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        if int(t.item()) != dist.get_world_size():
            raise RuntimeError("the first all-reduce over %d ranks summed to %d" % (dist.get_world_size(), int(t.item())))
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 (c10d raises its own exception types; whatever it is, the run is over)
        print("bench.py: the %s process group did not come up (RANK=%s WORLD_SIZE=%s MASTER_ADDR=%s MASTER_PORT=%s): %s: %s"
              % ("RCCL" if backend == "nccl" else backend, os.environ.get("RANK"), os.environ.get("WORLD_SIZE"), os.environ.get("MASTER_ADDR"),
                 os.environ.get("MASTER_PORT"), type(e).__name__, e), file=sys.stderr)
        sys.stderr.flush()
        os._exit(3)  # (a half-initialised c10d store may hang interpreter shutdown)


def collective_identity(backend, rank, device_index, on_gpu=True):
    """Who took part: what the group itself reports (world size), the RCCL version torch was built against / loaded, and every rank's device."""
    import torch.distributed as dist
    me = {"rank": rank, "pid": os.getpid(), "device_index": device_index, "device_name": None, "pci_bus_id": None}
    if on_gpu:
        pr = torch.cuda.get_device_properties(device_index)
        me["device_name"] = pr.name
        if hasattr(pr, "pci_bus_id"):
            me["pci_bus_id"] = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
        if hasattr(pr, "uuid"):
            me["uuid"] = str(pr.uuid)
    ranks = [None] * dist.get_world_size()
    dist.all_gather_object(ranks, me)
    version = None
    if backend == "nccl":
        try:
            version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            version = "unavailable: %s" % e
    return {"world_size_seen": dist.get_world_size(), "backend": backend, "nccl_version": version, "ranks": ranks,
            "distinct_devices": len({(r["pci_bus_id"], r.get("uuid"), r["device_index"]) for r in ranks}) if on_gpu else None}


def predicted_efficiency(config, world):
    """What the one-GPU emulation says about this world size (tools/stripe_efficiency.py: every rank's share rendered in turn on ONE GPU, no collective,
    against the whole frame in the same launch mode): the newest committed profiles/r*/stripe_efficiency*_r*.jsonl that holds the workload.  None if none."""
    import glob
    root = os.path.dirname(os.path.abspath(__file__))
    best = None
    # (exactly stripe_efficiency_rNN.jsonl: the round's default run -- not its frame-by-frame / stand-in / one-working-grid siblings)
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r*", "stripe_efficiency_r[0-9][0-9].jsonl"))):
        with open(f) as fh:
            for line in fh:
                try:
                    row = json.loads(line)
                except ValueError:
                    continue
                w = (row.get("world") or {}).get(str(world))
                if row.get("workload") == config and w and "render_only_efficiency" in w:
                    best = {"efficiency_vs_world1": w["render_only_efficiency"], "kind": "one-GPU emulation of every rank's share, render only, no collective",
                            "frames_per_submit": row.get("frames_per_submit"), "emulated_gather": row.get("emulated_gather"),
                            "from": os.path.relpath(f, root)}
    return best


def timed_combo(make_runner, gather, payload, steps, warmup, distributed):
    """One (gather, payload) combination on the group: its own pipeline, `steps` timed frames bracketed like the primary region; MAX over ranks."""
    import torch.distributed as dist
    r = make_runner(gather, payload)
    dt, kms, ev, _ = timed_run(r, steps, warmup, True, 0.0)
    t = torch.tensor([dt, float(ev)], dtype=torch.float64, device=r.stats.device)
    if distributed:
        tm = t[:1].clone()
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        te = t[1:].clone()
        dist.all_reduce(te, op=dist.ReduceOp.SUM)
        dt, ev = float(tm.item()), float(te.item())
    gms = [a.elapsed_time(b) / max(1, n) for (a, b), n in zip(r.pipeline.gather_events, r.pipeline.gather_frames)]
    world = r.world
    rec = {"gather": gather, "payload": payload, "steps": steps, "value": ev / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt / steps, "frames_per_s": steps / dt,
           "collective_bytes_per_frame_and_rank": (32 if payload == "planes" else 4) * r.W * r.H // world,
           "rank0_render_ms": sum(kms) / len(kms) if kms else None, "rank0_gather_ms": sum(gms) / len(gms) if gms else None}
    del r
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return rec


def dist_world_size(distributed):
    if not distributed:
        return 1
    import torch.distributed as dist
    return dist.get_world_size()


def launch_ranks(n, argv, child=None):
    """`python bench.py --gpus N` without a launcher around it: N rank processes of this script (fv-srn_amd/tiles.py)."""
    from fvsrn_amd import tiles
    return tiles.launch_ranks(n, argv, script=os.path.abspath(__file__), child=child)


def only_the_json_line_on_stdout():
    """Everything this process and its libraries write to file descriptor 1 goes to stderr from here on -- RCCL prints a version banner with
    C stdio when a communicator comes up (measured r04: five lines behind the JSON line) -- and the returned stream is the real stdout, for the
    ONE JSON line of the contract."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--spinup-ms", type=float, default=250.0, help="untimed clock spin-up before the warm-up steps (0 = off)")
    ap.add_argument("--config", default="c32l4_fourier_1024x512", choices=sorted(CONFIGS))
    ap.add_argument("--activation", default="ReLU", choices=["ReLU", "SnakeAlt", "Snake", "Sine"])
    ap.add_argument("--early-out", action="store_true", help="as-shipped DVR with alpha early-out")
    ap.add_argument("--no-twin", action="store_true", help="skip the second-activation twin run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1 only: take the multi-GPU route (compact stripes, all_gather_into_tensor on the collective's stream, assemble) "
                         "with a one-rank process group, so that the RCCL path runs on a one-GPU box")
    ap.add_argument("--gather", default=PRIMARY_GATHER, choices=["all", "root"],
                    help="N > 1 / --force-collective, the PRIMARY timed region: the frame on rank 0 only (north_star's gather; default) or on every rank (all-gather)")
    ap.add_argument("--payload", default=PRIMARY_PAYLOAD, choices=["planes", "rgba8"],
                    help="N > 1 / --force-collective: what travels -- the eight fp32 planes of ImageEvaluatorSimple::render, or ExtractColor'ed RGBA8 words (4 B / pixel)")
    ap.add_argument("--no-variants", action="store_true",
                    help="N > 1 / --force-collective: skip the same-run extras (the whole frame on rank 0 alone before the group run, the three other "
                         "(gather, payload) combinations behind it)")
    ap.add_argument("--variant-steps", type=int, default=20, help="timed frames of each same-run extra")
    ap.add_argument("--frames-per-submit", type=int, default=8,
                    help="K camera poses of the rotation per call into the library, per LAUNCH (a work unit is (frame, pixel tile)) and per collective "
                         "(fvsrn_render_stripes_batch); 1 = one launch per frame, what a caller of render(width, height) gets frame by frame")
    ap.add_argument("--grid-encoding", default="float", choices=sorted(GRID_ENCODINGS), help="latent-grid encoding of the synthetic network (LatentGrid::Encoding)")
    ap.add_argument("--camera-distance", type=float, default=1.6, help="CameraOnASphere distance (1.6: the headline; 0.8: a close-up that trips the cell table's footprint rule)")
    ap.add_argument("--gradient-mode", default="off", choices=["off", "finite_differences", "adjoint"], help="shaded render: Phong BRDF with normals by this mode")
    ap.add_argument("--grid-volume", action="store_true",
                    help="side benchmark (not the headline metric): DVR of a dense grid volume, BASELINE.json configs[0]; one JSON line")
    ap.add_argument("--grid-res", type=int, default=256)
    ap.add_argument("--grid-size", type=int, default=256)
    ap.add_argument("--grid-interpolation", type=int, default=1, help="0 nearest, 1 trilinear, 2 tricubic")
    args = ap.parse_args()
    VARIANT.update(distance=args.camera_distance, gradient_mode={"off": 0, "finite_differences": 1, "adjoint": 2}[args.gradient_mode], encoding=args.grid_encoding)
    variant_tag = "".join(["_" + args.grid_encoding if args.grid_encoding != "float" else "", "_closeup" if args.camera_distance != 1.6 else "",
                           "_" + args.gradient_mode if args.gradient_mode != "off" else ""])
    if variant_tag:
        args.no_twin = True  # (the twin / exact-features lines belong to the headline shape)
    if args.grid_volume:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP kernels have no CPU fallback")
        return grid_volume_bench(args.grid_res, args.grid_size, args.grid_interpolation, 64, not args.no_cpu_baseline)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly with --gpus N: this process becomes the launcher of N rank processes and never touches a GPU itself
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    json_out = only_the_json_line_on_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    collective = distributed or args.force_collective
    if args.force_collective and world != 1:
        raise SystemExit("bench.py: --force-collective is the one-GPU mode of the multi-GPU route")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernels have no CPU fallback")
    # FVSRN_BENCH_BACKEND=gloo lets the N>1 orchestration be exercised on a box with ONE GPU (all ranks share it);
    # the driver's multi-GPU runs use the default "nccl" (= RCCL over xGMI), one rank per GPU.
    backend = os.environ.get("FVSRN_BENCH_BACKEND", "nccl")
    if backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible, one per rank is required" % (world, torch.cuda.device_count()))
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not distributed:  # a one-rank group of our own
            os.environ.setdefault("MASTER_PORT", "29541")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        init_group_or_exit(backend, device_index)
        if dist.get_world_size() != args.gpus:
            print("bench.py: --gpus %d but the process group holds %d ranks" % (args.gpus, dist.get_world_size()), file=sys.stderr)
            raise SystemExit(3)

    import fvsrn_amd  # noqa: F401
    from fvsrn_amd import capi, volnet_io

    cfg = CONFIGS[args.config]
    time_keys = TIME_KEYS.get(args.config, 1)
    vn, net = make_network(volnet_io, capi, cfg, args.activation, time_keys)
    info = net.info()
    identity = collective_identity(backend, rank, device_index) if collective else None
    # Same-run reference (before the group run): rank 0 renders the whole frames alone on the plain one-GPU route, the other ranks wait at the barrier
    world1 = None
    if collective and not args.no_variants:
        if rank == 0:
            r1 = Runner(capi, net, cfg, 0, 1, args.early_out, time_keys, frames_per_submit=args.frames_per_submit)
            dt1, k1, ev1, _ = timed_run(r1, args.variant_steps, args.warmup, False, args.spinup_ms)
            world1 = {"value": ev1 / dt1, "unit": "samples/s", "ms_per_step": 1e3 * dt1 / args.variant_steps, "frames_per_s": args.variant_steps / dt1,
                      "steps": args.variant_steps, "what": "rank 0 alone, whole frames, no collective, before the group run"}
            del r1
            torch.cuda.empty_cache()
        dist.barrier()
    runner = Runner(capi, net, cfg, rank, world, args.early_out, time_keys, force_collective=args.force_collective, gather=args.gather, payload=args.payload,
                    frames_per_submit=args.frames_per_submit)
    dt, kernel_ms, evaluated, executed = timed_run(runner, args.steps, args.warmup, collective, args.spinup_ms)
    plan = runner.pipeline.scenes[0].last_render_info()
    launched_kernel = runner.pipeline.scenes[0].last_kernel_name()  # the launch's own account (fvsrn_scene_last_kernel_name), not a forecast from the network
    host_us = runner.pipeline.host_us_per_frame
    gather_ms = [a.elapsed_time(b) / max(1, n) for (a, b), n in zip(runner.pipeline.gather_events, runner.pipeline.gather_frames)]
    # per-rank decomposition of a frame (HIP events on the render / collective streams of every rank): rank 0 prints all of them
    per_rank = None
    if collective:
        mine = torch.tensor([sum(kernel_ms) / max(1, len(kernel_ms)), sum(gather_ms) / max(1, len(gather_ms)), 1e3 * dt / args.steps, host_us], dtype=torch.float64, device="cuda")
        allr = torch.zeros((world, 4), dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(allr.view(-1), mine)
        per_rank = [{"rank": r, "render_ms": float(allr[r, 0]), "gather_ms": float(allr[r, 1]), "frame_period_ms": float(allr[r, 2]),
                     "host_us_per_frame": float(allr[r, 3])} for r in range(world)]
    if distributed:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([evaluated, executed], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        evaluated, executed = int(c[0]), int(c[1])

    frame_check = None
    if collective:  # untimed: the gathered stripes of the last frame equal a whole-frame render on this rank
        last = runner.last_frame
        gathered = runner.assemble(*runner.where(last))   # None on the ranks that do not hold the frame (--gather root)
        yaw = 2 * math.pi * (last % 64) / 64
        scene = capi.Scene(**build_scene_kwargs(capi, yaw, runner.stepsize, args.early_out))
        if time_keys > 1:
            net.set_time_and_ensemble((0.25 * last) % (time_keys - 1), 0)
        full = scene.render(net, runner.W, runner.H)
        torch.cuda.synchronize()
        from fvsrn_amd import tiles
        if gathered is None:
            frame_check = True
        elif args.payload == "rgba8":  # packed words: every 8-bit channel within one step of the whole-frame render's (depth segments re-associate sums)
            want = capi.extract_color(full, capi.CHANNEL_COLOR, False, 1.0, rgba8=True)
            d = torch.stack([((gathered >> s) & 255) - ((want >> s) & 255) for s in (0, 8, 16, 24)]).abs().max()
            # (bitwise where the launch shapes agree; a batch renders several poses per launch without depth segments: re-associated sums)
            frame_check = bool(int(d) <= (0 if args.force_collective and args.frames_per_submit == 1 else 1))
        else:
            frame_check = tiles.frames_match(full, gathered)
            if args.force_collective and args.frames_per_submit == 1:  # one rank, one launch shape: the assembled frame is the same render -- bitwise
                frame_check = frame_check and bool(torch.equal(torch.nan_to_num(full, nan=-7.0), torch.nan_to_num(gathered, nan=-7.0)))
        ok = torch.tensor([1 if frame_check else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        frame_check = bool(ok.item())

    variants = None
    if collective and not args.no_variants:
        def make_runner(g, p):
            return Runner(capi, net, cfg, rank, world, args.early_out, time_keys, force_collective=args.force_collective, gather=g, payload=p,
                          frames_per_submit=args.frames_per_submit)
        variants = [timed_combo(make_runner, g, p, args.variant_steps, args.frames_per_submit, distributed)
                    for g, p in COMBOS if (g, p) != (args.gather, args.payload)]

    def tag_of(activation):  # the name tools/pmc_profile.sh files the PMC summary of (config, activation) under
        base = args.config if activation == "ReLU" else args.config.replace("_1024x512", "_%s_1024x512" % activation.lower())
        return base + variant_tag

    single = None
    if not args.no_twin and not collective and args.frames_per_submit > 1:
        # the same frames one launch per frame: what a caller of the reference's render(width, height) gets when it renders a sequence frame by frame
        r1 = Runner(capi, net, cfg, rank, world, args.early_out, time_keys, frames_per_submit=1)
        dt1, k1, ev1, _ = timed_run(r1, args.steps, args.warmup, False)
        single = {"frames_per_launch": 1, "value": ev1 / dt1, "unit": "samples/s", "ms_per_step": 1e3 * dt1 / args.steps, "steps": args.steps,
                  "kernel_ms_avg": sum(k1) / len(k1), "mfma_frac": info.flops_per_sample * (ev1 / args.steps) / (sum(k1) * 1e-3 / len(k1)) / 1e12 / MFMA_F16_PEAK_TFLOPS}
    twin = None
    if not args.no_twin and not collective:
        other = "SnakeAlt" if args.activation == "ReLU" else "ReLU"
        _, net2 = make_network(volnet_io, capi, cfg, other, time_keys)
        r2 = Runner(capi, net2, cfg, rank, world, args.early_out, time_keys, frames_per_submit=args.frames_per_submit)
        dt2, k2, ev2, ex2 = timed_run(r2, args.steps, args.warmup, False)  # the same step counts as the primary
        plan2 = r2.pipeline.scenes[0].last_render_info()
        twin = {"activation": other, "value": ev2 / dt2, "unit": "samples/s", "ms_per_step": 1e3 * dt2 / args.steps, "steps": args.steps,
                "kernel": r2.pipeline.scenes[0].last_kernel_name(),
                "roofline": roofline(net2.info(), other, ev2 / args.steps, sum(k2) * 1e-3 / len(k2), tag_of(other), plan2["rotation_resync"],
                                     algorithmic_bytes=32.0 * cfg[3] * cfg[4])}
        twin["mfma_frac"] = twin["roofline"]["mfma_frac"]

    exact = None
    if not args.no_twin and not collective:
        # the same frames with exact Fourier features at every step (FVSRN_OPT_FOURIER_RESYNC = 1: the reference's per-sample
        # arithmetic, positions rounded to fp16 at every sample, no feature rotation)
        r3 = Runner(capi, net, cfg, rank, world, args.early_out, time_keys, scene_options={"fourier_resync": 1}, frames_per_submit=args.frames_per_submit)
        dt3, k3, ev3, ex3 = timed_run(r3, args.steps, args.warmup, False)
        plan3 = r3.pipeline.scenes[0].last_render_info()
        exact = {"option": "FVSRN_OPT_FOURIER_RESYNC=1", "activation": args.activation, "value": ev3 / dt3, "unit": "samples/s",
                 "ms_per_step": 1e3 * dt3 / args.steps, "steps": args.steps,
                 "roofline": roofline(info, args.activation, ev3 / args.steps, sum(k3) * 1e-3 / len(k3), "no_profile_of_this_option", plan3["rotation_resync"])}
        exact["mfma_frac"] = exact["roofline"]["mfma_frac"]

    spread = None
    if not collective and not args.no_twin and args.steps >= 8:
        # How far the headline fraction moves on THIS box within one process (the chip holds a lower clock the denser and the longer the load): three
        # more rounds of 20 frames, interleaved with 20 frames of the other activation so that the rounds do not all sample the state the timed
        # region left behind; frac = algorithmic FLOP / HIP-event kernel time / peak per round.
        other_net = make_network(volnet_io, capi, cfg, "SnakeAlt" if args.activation == "ReLU" else "ReLU", time_keys)[1]
        ra = Runner(capi, net, cfg, rank, world, args.early_out, time_keys, frames_per_submit=args.frames_per_submit)
        rb = Runner(capi, other_net, cfg, rank, world, args.early_out, time_keys, frames_per_submit=args.frames_per_submit)
        fracs = []
        for _ in range(3):
            _, kms, ev, _ = timed_run(ra, 20, 2, False)
            fracs.append(info.flops_per_sample * (ev / 20) / (sum(kms) * 1e-3 / len(kms)) / 1e12 / MFMA_F16_PEAK_TFLOPS)
            timed_run(rb, 20, 2, False)
        fracs.sort()
        spread = {"mfma_frac_min": fracs[0], "mfma_frac_median": fracs[1], "mfma_frac_max": fracs[2], "mfma_frac_rounds": "3 x 20 frames, interleaved with the other activation"}
    if rank == 0:
        _, _, _, W, H, steps = cfg
        kernel_s = sum(kernel_ms) * 1e-3 / len(kernel_ms)             # average launch duration on this rank
        if runner.pipelined:  # consecutive frames overlap on two streams: the event pairs overlap too, use the frame period
            kernel_s = dt / args.steps
        rl = roofline(info, args.activation, (evaluated / world) / args.steps, kernel_s, tag_of(args.activation), plan["rotation_resync"], world,
                      algorithmic_bytes=32.0 * W * H)
        if spread:
            rl.update(spread)
        # roofline.kernel_ms_avg is per FRAME; one launch renders frames_per_launch frames and lasts launch_ms_avg -- the duration rocprofv3 lists for the kernel
        full_launches = [t for t, n in zip(runner.launch_ms, runner.frames_per_launch) if n == max(runner.frames_per_launch)]
        rl["frames_per_launch"] = max(runner.frames_per_launch)
        rl["launch_ms_avg"] = sum(full_launches) / len(full_launches)
        out = {
            "metric": "srn_samples_per_s", "value": evaluated / dt, "unit": "samples/s", "n_gpus": world,
            "world_size": dist_world_size(collective), "backend": backend if collective else None,
            "steps": args.steps, "warmup": args.warmup, "spinup_ms": args.spinup_ms, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d steps/ray, %d-wide x %d-layer fp16 SRN%s, %s, density:direct + Identity TF, "
                                   "early-out %s" % (args.config, W, H, steps, cfg[0], cfg[1],
                                                     (" + %d-ch %d^3 latent grid" % cfg[2]) if cfg[2] else " (Fourier-only)",
                                                     args.activation, "on" if args.early_out else "off"),
                       "parallelism": "1 GPU" if world == 1 else "%d GPUs, round-robin %d-row stripes + RCCL %s of %s" % (
                           world, STRIPE, "all-gather" if args.gather == "all" else "gather to rank 0", "8 fp32 planes" if args.payload == "planes" else "RGBA8 words")},
            "variant": {"grid_encoding": args.grid_encoding, "camera_distance": args.camera_distance, "gradient_mode": args.gradient_mode,
                        # network evaluations behind one counted sample: 7 with finite differences, value + 3 tangent tiles in the adjoint pass (roofline.achieved
                        # stays the algorithmic FLOP of ONE evaluation per sample: the line's frac is comparable with the unshaded one, not a utilisation)
                        "network_evaluations_per_sample": {"off": 1, "finite_differences": 7, "adjoint": 4}[args.gradient_mode]} if variant_tag else None,
            "frames_per_s": args.steps / dt,
            "evaluated_samples_per_frame": evaluated / args.steps,
            "nominal_samples_per_frame": W * H * steps,
            "wave_executed_samples_per_frame": executed / args.steps,
            "kernel": launched_kernel,
            "frames_per_submit": args.frames_per_submit,  # camera poses per library call / launch / collective (1: frame by frame)
            "host_us_per_frame": host_us,  # wall time of this rank's host inside submit per frame (scene update, launch, events, the collective's enqueue)
            # what the launches of the timed region did (fvsrn_scene_last_render_info): depth segments per ray, period of the exact re-derivation of rotated
            # Fourier features (0: derived at every step), latent grid through the cell table or by gathers (None: no latent grid)
            "launch": {"depth_segments": plan["segments"], "rotation_resync": plan["rotation_resync"],
                       "latent_grid": ("cell_table" if plan["cell_table"] else "gather") if info.grid_channels > 0 else None},
            "roofline": rl,
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
            out["gather"], out["payload"] = args.gather, args.payload
            out["rccl"] = identity
            if world1:
                out["world1_same_run"] = world1
                out["efficiency_vs_same_run_world1"] = (evaluated / dt) / (world * world1["value"])
            pred = predicted_efficiency(args.config, world) if world > 1 else None
            if pred and world1:
                pred["value"] = pred["efficiency_vs_world1"] * world * world1["value"]
            out["predicted"] = pred
            if variants is not None:
                out["variants"] = variants
            out["collective_bytes_per_frame_and_rank"] = (32 if args.payload == "planes" else 4) * runner.W * runner.H // world
            out["stripe_launches"] = {"persistent": bool(runner.pipeline.persistent_stripes), "hw_streams_concurrent": runner.pipeline.hw_streams_concurrent,
                                      "needs": "GPU_MAX_HW_QUEUES >= 8 in the environment before the process starts (set by bench.py / launch_ranks)",
                                      "persistent_reserve": "1/16 of the workgroup slots (FVSRN_OPT_PERSISTENT_RESERVE, automatic)"}
        if args.force_collective:
            out["force_collective"] = True
        if frame_check is not None:
            out["gathered_frame_matches_single_gpu_frame"] = frame_check
        if single:
            out["single_frame_launches"] = single
        if twin:
            out["twin"] = twin
        if exact:
            out["exact_features"] = exact
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args.activation)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if collective:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
