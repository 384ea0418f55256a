#!/usr/bin/env python3
"""Randomised parity of fvsrn_evaluate_points against the oracle over the variant space of the evaluate kernels (r05: one launch, batches with a point
outside the unit box take the scaled ReLU image unclamped inside the kernel; fp16 I/O): widths 16 .. 128, 2 .. 6 layers, every activation, with and
without a latent grid in every encoding, scalar and colour outputs, positions inside / straddling / outside / far from the box, fp32 and fp16 tensors.
usage (GPU box): python tools/dev/fuzz_evaluate.py [cases]      one line per case, a summary line at the end; exit code 1 on a miss"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from fvsrn_amd import capi, volnet_io  # noqa: E402
from oracle import oracle  # noqa: E402

TOL = 2e-3  # TOL_SAME_MODEL of tests/test_gpu_parity.py (network outputs against the fp32-accumulate oracle), times max(1, |value|)


def draw(seed):
    rng = np.random.RandomState(7000 + seed)
    C = int(rng.choice([16, 32, 32, 32, 48, 64, 64, 80, 96, 112, 128]))
    layers = int(rng.randint(2, 7)) if C <= 64 else int(rng.randint(2, 4))
    act = str(rng.choice(["ReLU", "ReLU", "ReLU", "SnakeAlt", "Sine", "Snake", "Sigmoid"]))
    out = str(rng.choice(["density", "density:direct", "density:direct", "rgbo", "rgbo:direct"]))
    grid = (16, int(rng.choice([4, 8, 16]))) if rng.rand() < 0.4 else None
    enc = int(rng.choice([0, 0, 1, 2])) if grid else 0
    ladder_fits = (C - 4) // 2 <= 30  # (a NeRF ladder of more features leaves the half range)
    std = None if (ladder_fits and rng.rand() < 0.6) else 0.4
    net = dict(C=C, layers=layers, activation=act, output_mode=out, grid=grid, encoding=enc, fourier_std=std, seed=int(rng.randint(1 << 20)), grid_scale=0.3)
    where = str(rng.choice(["inside", "inside", "mixed", "outside", "one_point", "far"]))
    n = int(rng.choice([1, 63, 64, 65, 1000, 4097, 20011]))
    pos = rng.uniform(0.0, 1.0, (n, 3)).astype(np.float32)
    if where == "mixed":
        m = rng.rand(n) < 0.3
        pos[m] = rng.uniform(-0.7, 1.8, (int(m.sum()), 3)).astype(np.float32)
    elif where == "outside":
        pos = rng.uniform(1.01, 1.6, (n, 3)).astype(np.float32)
    elif where == "one_point":
        pos[n // 2, int(rng.randint(3))] = -0.01
    elif where == "far":
        m = rng.rand(n) < 0.5
        pos[m] = rng.uniform(-3.0, 4.0, (int(m.sum()), 3)).astype(np.float32)
    return net, where, pos, bool(rng.rand() < 0.35)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    worst, misses = 0.0, 0
    for seed in range(cases):
        net_kw, where, pos, half = draw(seed)
        vn = util.random_network(**net_kw)
        net = capi.Network.from_volnet(volnet_io.save_volnet(vn))
        if half:
            pos = pos.astype(np.float16).astype(np.float32)  # the positions the fp16 call sees
        ref = oracle.OracleNetwork(vn, oracle.ACC_FLOAT).evaluate(pos)
        p = torch.from_numpy(pos).cuda()
        out = net.evaluate(p.half() if half else p).float().cpu().numpy()
        scale = np.maximum(1.0, np.abs(ref))
        err = np.abs(out - ref) / scale
        if half:
            err = np.maximum(err - 2.0 ** -11, 0.0)  # the rounding of the value to half
        e = float(err.max()) if err.size else 0.0
        ok = e < TOL and not np.isnan(out).any()
        worst = max(worst, e)
        misses += 0 if ok else 1
        print("%4d %-4s err %.2e  %-9s n %-6d %s  C %d layers %d %s %s grid %s enc %d fourier %s  kernel %s" % (
            seed, "OK" if ok else "MISS", e, where, pos.shape[0], "fp16" if half else "fp32", net_kw["C"], net_kw["layers"], net_kw["activation"], net_kw["output_mode"],
            net_kw["grid"], net_kw["encoding"], "ladder" if net_kw["fourier_std"] is None else "gaussian", net.kernel_name(False)), flush=True)
    print("cases %d, misses %d, worst relative-to-max(1,|v|) error %.2e (tolerance %.0e)" % (cases, misses, worst, TOL))
    sys.exit(1 if misses else 0)


if __name__ == "__main__":
    main()
