#!/usr/bin/env python3
"""BASELINE.json configs[0] on the GPU: DVR of a dense grid volume (VolumeInterpolationGrid, 256^2 image) next to the CPU
restatement on the host cores.  The reference's example-volume.cvol is not in the repository snapshot (.MISSING_LARGE_BLOBS):
a synthetic 256^3 float volume stands in.
usage: tools/bench_grid_volume.py [--res 256] [--size 256] [--interpolation 1] [--no-cpu]   -> one JSON line"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--interpolation", type=int, default=1, help="0 nearest, 1 trilinear, 2 tricubic")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    import torch
    import fvsrn_amd  # noqa: F401
    from fvsrn_amd import capi
    n = a.res
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
    out = torch.zeros((1, 8, a.size, a.size), dtype=torch.float32, device="cuda")
    stats = torch.zeros(2, dtype=torch.int64, device="cuda")
    for i in range(4):
        vol.render(scene, a.size, a.size, a.interpolation, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(a.frames):
        scene.update(**kwargs(2 * np.pi * i / 64))
        vol.render(scene, a.size, a.size, a.interpolation, out=out, stats=stats)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.frames
    samples = int(stats.cpu()[0]) / a.frames
    taps = {0: 1, 1: 8, 2: 64}[a.interpolation]
    res = {"workload": "grid_dvr: %d^3 fp32 volume, %dx%d, step 1/%d, interpolation %d, Identity TF, early-out off" % (n, a.size, a.size, n, a.interpolation),
           "ms_per_frame": ms, "frames_per_s": 1e3 / ms, "samples_per_s": samples / (ms * 1e-3), "samples_per_frame": samples,
           "voxel_reads_per_s": taps * samples / (ms * 1e-3), "voxel_read_GBps_algorithmic": 4 * taps * samples / (ms * 1e-3) / 1e9,
           "volume_MB": data.nbytes / 1e6}
    if not a.no_cpu:
        from oracle import oracle
        ov = oracle.OracleVolume(data, (-0.5, -0.5, -0.5), (1, 1, 1), a.interpolation, oracle.VOLUME_SOURCE_TEXTURE)
        kw = kwargs(0.0)
        kw.pop("tf_kind")
        sc = oracle.OracleScene(tf_kind=oracle.TF_IDENTITY, **kw)
        t0 = time.perf_counter()
        ref, cnt = ov.render(sc, a.size, a.size)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": cnt / dt, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "oracle_render_volume (C restatement, OpenMP) of the same frame, %.2f s" % dt}
        scene.update(**kwargs(0.0))
        img = vol.render(scene, a.size, a.size, a.interpolation)[0].cpu().numpy()
        res["max_abs_diff_vs_cpu"] = float(np.abs(img[:4] - ref[:4]).max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
