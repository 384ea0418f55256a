#!/usr/bin/env python3
"""BASELINE.json configs[0] (DVR of a dense grid volume) = `bench.py --grid-volume`; kept as a shortcut.
usage: tools/bench_grid_volume.py [--res 256] [--size 256] [--interpolation 1] [--no-cpu]"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=256)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--interpolation", type=int, default=1)
ap.add_argument("--no-cpu", action="store_true")
a = ap.parse_args()
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--grid-volume", "--grid-res", str(a.res), "--grid-size", str(a.size),
       "--grid-interpolation", str(a.interpolation)] + (["--no-cpu-baseline"] if a.no_cpu else [])
sys.exit(subprocess.call(cmd))
