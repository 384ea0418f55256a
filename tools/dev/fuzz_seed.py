"""Box script: one fuzz seed with the latent grid through the cell table, on the gather path, and with exact features -- which part of the error is whose?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import test_fuzz_parity as f
for seed in [int(a) for a in sys.argv[1:]]:
    for opts in ({"cell_table": 1}, {"cell_table": 0}, {"cell_table": 1, "fourier_resync": 1}, {"cell_table": 0, "small_kernel": 0}):
        r = f.compare_case(seed, dict(opts))
        d = np.abs(r["img"][:7] - r["dev"][:7])
        ch = [float(d[c].max()) for c in range(7)]
        print(seed, opts, "gpu-DEVICE %.2e gpu-FLOAT %.2e spread %.2e per channel %s plan %s" % (r["err_device"], r["err_float"], r["spread"], ["%.1e" % v for v in ch], r["plan"]))
