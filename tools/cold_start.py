#!/usr/bin/env python3
"""Cold-start figures of the AOT library, from a FRESH process on the GPU box:

  dlopen_ms            ctypes.CDLL(libfvsrn.so) -- the loader maps the file and registers its code objects with the HIP runtime
  first_render_ms      network from .volnet bytes + scene + the first fvsrn_render of the headline network (32x4 Fourier, 512^2, 256 steps), synchronised
  second_render_ms     the same frame again (what a warm process pays)
  other_family_ms      first render of a different kernel family (64x6 + 32^3 latent grid, 256^2): only that family's code object is loaded lazily
  hip_init_ms          torch.cuda.init() + a first synchronise, measured before the library is touched

The reference compiles one kernel per configuration at first use with NVRTC and caches it (kernel_loader.cpp:285-366); the figure to hold this
against is its first-frame compile time, which the reference does not publish.  Usage: python tools/cold_start.py [--out file.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    rec = {}
    t0 = time.perf_counter()
    import numpy as np
    import torch
    rec["import_torch_ms"] = 1e3 * (time.perf_counter() - t0)
    t0 = time.perf_counter()
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    rec["hip_init_ms"] = 1e3 * (time.perf_counter() - t0)

    import fvsrn_amd  # noqa: F401
    from fvsrn_amd import capi, synthetic, volnet_io
    rec["library_bytes"] = os.path.getsize(capi.LIB_PATH)
    t0 = time.perf_counter()
    ctypes.CDLL(capi.LIB_PATH)
    rec["dlopen_ms"] = 1e3 * (time.perf_counter() - t0)
    capi.lib()

    def first_frames(C, layers, grid, size, steps, n=3):
        vn = synthetic.random_network(C=C, layers=layers, activation="ReLU", param=1.0, output_mode="density:direct", grid=grid, seed=1234,
                                      box_min=(-0.5, -0.5, -0.5), grid_scale=0.01)
        data = volnet_io.save_volnet(vn)
        eye, right, up = capi.camera_on_a_sphere("Ym", (0, 0, 0), 0.4, 0.0, 1.6)
        out = []
        t0 = time.perf_counter()
        net = capi.Network.from_volnet(data)
        sc = capi.Scene(eye=eye, right=right, up=up, fov_y_radians=float(np.deg2rad(45.0)), stepsize=1.0 / steps, early_out=False,
                        tf_kind=capi.TF_IDENTITY, tf_scale_absorption=10.0, tf_scale_emission=1.0)
        for _ in range(n):
            sc.render(net, size, size)
            torch.cuda.synchronize()
            out.append(1e3 * (time.perf_counter() - t0))
            t0 = time.perf_counter()
        return out, sc.last_kernel_name()

    f, k = first_frames(32, 4, None, 512, 256)
    rec["first_render_ms"], rec["second_render_ms"], rec["third_render_ms"], rec["first_kernel"] = f[0], f[1], f[2], k
    f, k = first_frames(64, 6, (16, 32), 256, 256)
    rec["other_family_ms"], rec["other_family_second_ms"], rec["other_family_kernel"] = f[0], f[1], k
    f, k = first_frames(32, 4, (16, 16), 256, 256)
    rec["third_family_ms"], rec["third_family_second_ms"], rec["third_family_kernel"] = f[0], f[1], k
    line = json.dumps(rec)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            fh.write(line + "\n")


if __name__ == "__main__":
    main()
