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
           (applications/volnet/eval_NetworkConfigsGrid.py:31, activationX = ["ReLU"]); the paper's SnakeAlt
           twin is timed beside it and reported in "snakealt_twin".
metric: SRN samples/s, counting lane-exact EVALUATED samples (the loop bound of
renderer_ray_evaluation_stepping_dvr.cuh:84-90 summed over all rays), read from the kernel's own counter.

N > 1 (`python bench.py --gpus N` starts the N rank processes itself; under torch.distributed.run it takes the launcher's
RANK / WORLD_SIZE instead): the SAME frame is split into round-robin 16-row stripes, one
process per GPU, each rank renders its stripes and one RCCL all-gather assembles the frame ("strong" scaling;
the gather of frame i overlaps the render of frame i+1 on a side stream, and consecutive frames alternate between two render
streams so that the tail of one launch overlaps the start of the next).

Prints ONE JSON line (see the contract in the task description) with "roofline" and "cpu_baseline".
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
STRIPE = 16

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


def build_scene_kwargs(capi, yaw, stepsize, early_out):
    eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, yaw, 1.6)
    return dict(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=stepsize,
                early_out=early_out, tf_kind=capi.TF_IDENTITY, tf_scale_absorption=10.0, tf_scale_emission=1.0)


def bench_network(C, layers, grid, activation, time_keys=1):
    """The synthetic network of a workload: seed 1234, nn.Linear-style init, NeRF ladder, density:direct,
    latent grid randn * 0.01.  tests/test_gpu_parity.py renders the same networks against the oracle."""
    from fvsrn_amd import synthetic
    return synthetic.random_network(C=C, layers=layers, activation=activation, param=1.0, output_mode="density:direct",
                                    grid=grid, seed=1234, box_min=(-0.5, -0.5, -0.5), grid_scale=0.01, time_grids=time_keys)


def make_network(volnet_io, capi, cfg, activation, time_keys=1):
    C, layers, grid, *_ = cfg
    vn = bench_network(C, layers, grid, activation, time_keys)
    return vn, capi.Network.from_volnet(volnet_io.save_volnet(vn))


class Runner:
    """Renders frames of one network on this rank (whole frame, or this rank's stripes + all-gather)."""

    def __init__(self, capi, net, cfg, rank, world, early_out, time_keys=1):
        self.capi, self.net, self.rank, self.world = capi, net, rank, world
        self.time_keys = time_keys
        _, _, _, self.W, self.H, steps = cfg
        self.stepsize = 1.0 / steps
        self.early_out = early_out
        self.scene = capi.Scene(**build_scene_kwargs(capi, 0.0, self.stepsize, early_out))
        self.stats = torch.zeros(2, dtype=torch.int64, device="cuda")
        # N > 1: frames alternate between two scenes on two streams, so that the launch tail of frame i (its last waves, the
        # composite of its depth segments) overlaps the start of frame i + 1.  Not with time-dependent grids: a time change
        # rewrites the working grid the previous frame may still read.  At N = 1 the same trick is worth +4 % (r01: 152.7 ->
        # 159.0 Gsamples/s, FVSRN_BENCH_PIPELINE=1), but the default keeps one launch at a time there so that the HIP-event
        # duration of the kernel, the rocprofv3 kernel trace and the frame period are the same number.
        pipe = os.environ.get("FVSRN_BENCH_PIPELINE")
        self.pipelined = time_keys == 1 and (pipe == "1" if pipe is not None else world > 1)
        if world == 1:
            self.outs = [torch.zeros((1, 8, self.H, self.W), dtype=torch.float32, device="cuda") for _ in range(2)]
            self.out = self.outs[0]
        else:
            assert self.H % (STRIPE * world) == 0, "image height must be a multiple of stripe*world"
            rows = capi.stripe_rows(self.H, STRIPE, rank, world)
            self.local = [torch.zeros((8, rows, self.W), dtype=torch.float32, device="cuda") for _ in range(2)]
            self.gathered = [torch.zeros((world, 8, rows, self.W), dtype=torch.float32, device="cuda") for _ in range(2)]
            self.comm_stream = torch.cuda.Stream()
            self.render_done = [torch.cuda.Event() for _ in range(2)]
            self.gather_done = [torch.cuda.Event() for _ in range(2)]
        if self.pipelined:
            self.scenes = [self.scene, capi.Scene(**build_scene_kwargs(capi, 0.0, self.stepsize, early_out))]
            # First use of a network handle (upload of the LDS image and the key frames, include/fvsrn.h) is ordered on the
            # stream of the call that triggers it only: do it once here, untimed, and let both render streams start behind it.
            if world == 1:
                self.scene.render(net, self.W, self.H, out=self.outs[0])
            else:
                capi.render_stripes(self.scene, net, self.W, self.H, STRIPE, rank, world, out=self.local[0])
            torch.cuda.synchronize()
            self.render_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            for st in self.render_streams:
                st.wait_stream(torch.cuda.current_stream())
        self.kernel_events = []

    def frame(self, index, record=False, gather=True):
        import torch.distributed as dist
        yaw = 2 * math.pi * (index % 64) / 64
        b = index & 1
        scene = self.scenes[b] if self.pipelined else self.scene
        stream = self.render_streams[b] if self.pipelined else torch.cuda.current_stream()
        scene.update(**build_scene_kwargs(self.capi, yaw, self.stepsize, self.early_out))
        if self.time_keys > 1:  # key frames are resident in HBM; this only schedules the device-side blend
            self.net.set_time_and_ensemble((0.25 * index) % (self.time_keys - 1), 0)
        with torch.cuda.stream(stream):
            if record:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.world == 1:
                scene.render(self.net, self.W, self.H, out=self.outs[b], stats=self.stats)
            else:
                stream.wait_event(self.gather_done[b])  # buffer b free again
                self.capi.render_stripes(scene, self.net, self.W, self.H, STRIPE, self.rank, self.world,
                                         out=self.local[b], stats=self.stats)
                self.render_done[b].record()
            if record:
                e1.record()
                self.kernel_events.append((e0, e1))
        if self.world > 1 and gather:
            with torch.cuda.stream(self.comm_stream):  # gather(frame i) overlaps render(frame i+1)
                self.comm_stream.wait_event(self.render_done[b])
                dist.all_gather_into_tensor(self.gathered[b].view(self.world * 8, -1, self.W), self.local[b])
                self.gather_done[b].record()

    def finish(self):
        if self.pipelined:
            for st in self.render_streams:
                torch.cuda.current_stream().wait_stream(st)
        if self.world > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def assemble(self, b=0):
        """(world, 8, rows, W) -> (1, 8, H, W): undo the round-robin stripe order."""
        if self.world == 1:
            return self.out
        g = self.gathered[b]
        R, _, rows, W = g.shape
        return g.view(R, 8, rows // STRIPE, STRIPE, W).permute(1, 2, 0, 3, 4).reshape(1, 8, self.H, W)


def timed_run(runner, steps, warmup, distributed, spinup_ms=0.0):
    import torch.distributed as dist
    # Clock spin-up (untimed, before the W warm-up steps): the shader clock of an idle MI355X needs a few hundred ms of load to
    # settle (r01: the same 32 timed frames read 1.5 % lower after 4 warm-up frames than after 150, 8 timed frames 5 % lower), and
    # at N GPUs the timed region shrinks to 32 x 0.3 ms.  Every rank spins for the same wall time, rendering only (the number of
    # frames differs between ranks, so no collective is called here).
    if distributed and spinup_ms > 0:  # one untimed frame with its gather on every rank: RCCL sets its communicator up lazily
        runner.frame(0)
        runner.finish()
        torch.cuda.synchronize()
        dist.barrier()
    t_end = time.perf_counter() + 1e-3 * spinup_ms
    i = 0
    while time.perf_counter() < t_end:
        runner.frame(i, gather=False)
        i += 1
        if i % 8 == 0:
            runner.finish()
            torch.cuda.synchronize()
    for i in range(warmup):
        runner.frame(i)
    runner.finish()
    torch.cuda.synchronize()
    runner.stats.zero_()
    runner.kernel_events.clear()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        runner.frame(warmup + i, record=True)
    runner.finish()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    dt = time.perf_counter() - t0
    kernel_ms = [a.elapsed_time(b) for a, b in runner.kernel_events]
    st = runner.stats.cpu().numpy().astype(np.int64)
    return dt, kernel_ms, int(st[0]), int(st[1])


def cpu_baseline(cfg, activation):
    """The reference's PyTorch path (port: oracle/torch_port.py) on the host cores, bounded sample."""
    from oracle import torch_port
    from fvsrn_amd import capi
    C, layers, grid, *_ = cfg
    rng = np.random.RandomState(1234)
    F = (C - 4) // 2
    blocks = [(2.0 ** i) * np.eye(3) for i in range((F + 2) // 3)]
    B = (np.concatenate(blocks, axis=0)[:F] * 2 * np.pi).astype(np.float32)
    G = grid[0] if grid else 0
    dims = [3 + 2 * F + G] + [C] * (layers - 1) + [1]
    ws = [rng.uniform(-1, 1, (dims[i + 1], dims[i])).astype(np.float32) / np.sqrt(dims[i]) for i in range(layers)]
    bs = [rng.uniform(-1, 1, dims[i + 1]).astype(np.float32) / np.sqrt(dims[i]) for i in range(layers)]
    g = (rng.randn(1, G, grid[1], grid[1], grid[1]) * 0.01).astype(np.float32) if grid else None
    net = torch_port.TorchSRN(B, ws, bs, activation, 1.0, "density:direct", g)
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
                      "same network/camera, %dx%d rays at step 1/%d, the first %d network samples (%.1f s of the step loop)"
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


def pmc_traffic(workload):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC summary of this workload
    (profiles/r*/<workload>_*_pmc.csv, written by tools/pmc_profile.sh: separate --pmc passes; WRITE_SIZE is exact,
    FETCH_SIZE counts 128-B requests as 64 B on gfx950 and is doubled, MI355X_MICROARCH.md).  None if there is no
    profile of this workload: the counters cannot be read inside the timed run.  Unit: bytes per launch."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*", workload + "_*pmc.csv")))
    if not files:
        return None
    vals = {}
    with open(files[-1]) as f:
        for row in csv.reader(l for l in f if not l.startswith("#")):
            if len(row) == 5 and row[2] in ("FETCH_SIZE", "WRITE_SIZE"):
                vals[row[2]] = float(row[4])
    if len(vals) != 2:
        return None
    return 1024.0 * (vals["WRITE_SIZE"] + 2.0 * vals["FETCH_SIZE"])  # bytes per launch


def dist_world_size(distributed):
    if not distributed:
        return 1
    import torch.distributed as dist
    return dist.get_world_size()


def launch_ranks(n, argv, child=None):
    """`python bench.py --gpus N` without a launcher around it: start N rank processes (fresh interpreters, one per GPU, RCCL
    rendezvous on 127.0.0.1), relay rank 0's JSON line and return the worst exit code.  This process never initialises a GPU
    (torch.cuda.device_count() does not, on this image).  Fewer than N visible GPUs is an error, not a silent 1-GPU run --
    unless FVSRN_BENCH_BACKEND=gloo, the test mode in which all ranks share the visible GPU(s).
    child: command prefix of a rank process (tests substitute a stand-in); default = this script."""
    backend = os.environ.get("FVSRN_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < n:
        print("bench.py --gpus %d: only %d GPU(s) visible, one per rank is required" % (n, have), file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = list(child) if child else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:  # rank 0 is gone and this one hangs in a collective: end exactly this process
            p.kill()
            rcs.append(p.wait())
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    return 0 if not bad else (bad[0] if bad[0] > 0 else 1)


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
    ap.add_argument("--grid-volume", action="store_true",
                    help="side benchmark (not the headline metric): DVR of a dense grid volume, BASELINE.json configs[0]; one JSON line")
    ap.add_argument("--grid-res", type=int, default=256)
    ap.add_argument("--grid-size", type=int, default=256)
    ap.add_argument("--grid-interpolation", type=int, default=1, help="0 nearest, 1 trilinear, 2 tricubic")
    args = ap.parse_args()
    if args.grid_volume:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP kernels have no CPU fallback")
        return grid_volume_bench(args.grid_res, args.grid_size, args.grid_interpolation, 64, not args.no_cpu_baseline)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly with --gpus N: this process becomes the launcher of N rank processes and never touches a GPU itself
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
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
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
        assert dist.get_world_size() == args.gpus

    import fvsrn_amd  # noqa: F401
    from fvsrn_amd import capi, volnet_io

    cfg = CONFIGS[args.config]
    time_keys = TIME_KEYS.get(args.config, 1)
    vn, net = make_network(volnet_io, capi, cfg, args.activation, time_keys)
    info = net.info()
    runner = Runner(capi, net, cfg, rank, world, args.early_out, time_keys)
    dt, kernel_ms, evaluated, executed = timed_run(runner, args.steps, args.warmup, distributed, args.spinup_ms)
    if distributed:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([evaluated, executed], dtype=torch.int64, device="cuda")
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        evaluated, executed = int(c[0]), int(c[1])

    frame_check = None
    if distributed:  # untimed: the gathered stripes of the last frame equal a whole-frame render on this rank
        last = args.warmup + args.steps - 1
        gathered = runner.assemble(last & 1)
        yaw = 2 * math.pi * (last % 64) / 64
        scene = capi.Scene(**build_scene_kwargs(capi, yaw, runner.stepsize, args.early_out))
        full = scene.render(net, runner.W, runner.H)
        torch.cuda.synchronize()
        # Same samples, but a rank's stripes are a small launch and may be rendered in depth segments (re-associated sums,
        # other restart points of the feature rotation): compare within the image tolerance of the parity tests, depth
        # (NaN where alpha == 0) only on pixels that are not within rounding of empty.
        solid = (full[0, 3] > 1e-4) | (gathered[0, 3] > 1e-4)
        frame_check = bool(
            float((full[0, :7] - gathered[0, :7]).abs().max()) < 3e-3
            and torch.equal(torch.isnan(full[0, 7])[solid], torch.isnan(gathered[0, 7])[solid])
            and float((torch.nan_to_num(full[0, 7], nan=0.0) - torch.nan_to_num(gathered[0, 7], nan=0.0))[solid].abs().max()) < 3e-2)
        ok = torch.tensor([1 if frame_check else 0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        frame_check = bool(ok.item())

    twin = None
    if not args.no_twin and not distributed:
        other = "SnakeAlt" if args.activation == "ReLU" else "ReLU"
        _, net2 = make_network(volnet_io, capi, cfg, other, time_keys)
        r2 = Runner(capi, net2, cfg, rank, world, args.early_out, time_keys)
        dt2, k2, ev2, ex2 = timed_run(r2, args.steps, args.warmup, False)  # the same step counts as the primary
        twin = {"activation": other, "value": ev2 / dt2, "unit": "samples/s", "ms_per_step": 1e3 * dt2 / args.steps, "steps": args.steps,
                "kernel": net2.kernel_name(True),
                "mfma_frac": (ev2 * info.flops_per_sample / (sum(k2) * 1e-3)) / (MFMA_F16_PEAK_TFLOPS * 1e12)}

    if rank == 0:
        _, _, _, W, H, steps = cfg
        kernel_s = sum(kernel_ms) * 1e-3 / len(kernel_ms)             # average launch duration on this rank
        if runner.pipelined:  # consecutive frames overlap on two streams: the event pairs overlap too, use the frame period
            kernel_s = dt / args.steps
        flops_per_launch = info.flops_per_sample * (evaluated / world) / args.steps  # algorithmic, SURVEY 8(d)
        achieved = flops_per_launch / kernel_s / 1e12
        out = {
            "metric": "srn_samples_per_s", "value": evaluated / dt, "unit": "samples/s", "n_gpus": world,
            "world_size": dist_world_size(distributed), "backend": backend if distributed else None,
            "steps": args.steps, "warmup": args.warmup, "spinup_ms": args.spinup_ms, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "%s: %dx%d, %d steps/ray, %d-wide x %d-layer fp16 SRN%s, %s, density:direct + Identity TF, "
                                   "early-out %s" % (args.config, W, H, steps, cfg[0], cfg[1],
                                                     (" + %d-ch %d^3 latent grid" % cfg[2]) if cfg[2] else " (Fourier-only)",
                                                     args.activation, "on" if args.early_out else "off"),
                       "parallelism": "1 GPU" if world == 1 else "%d GPUs, round-robin %d-row stripes + RCCL all-gather" % (world, STRIPE)},
            "frames_per_s": args.steps / dt,
            "evaluated_samples_per_frame": evaluated / args.steps,
            "nominal_samples_per_frame": W * H * steps,
            "wave_executed_samples_per_frame": executed / args.steps,
            "kernel": net.kernel_name(True),
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_F16_PEAK_TFLOPS, "traffic": pmc_traffic(args.config) if world == 1 else None,
                         "flops_per_sample": info.flops_per_sample, "mfma_flops_per_sample": info.mfma_flops_per_sample,
                         "kernel_ms_avg": 1e3 * kernel_s},
        }
        if frame_check is not None:
            out["gathered_frame_matches_single_gpu_frame"] = frame_check
        if twin:
            out["twin"] = twin
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, args.activation)
        print(json.dumps(out))
    if distributed:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
