#!/usr/bin/env python3
"""The reference's network-configuration study (applications/volnet/eval_NetworkConfigsGrid.py) as a timing run on the MI355X build: its fifteen
(channels, layers) networks -- (32,2) (32,4) (32,10) (32,16) (32,22) (48,2) (48,4) (48,6) (48,8) (48,10) (64,2) (64,4) (64,6) (96,3) (128,2), :37 -- each with
its 16-channel 32^3 latent grid (:22-23), (C - 4) / 2 Fourier features of std 1 (:62) and ReLU (what the study times, :35; --activation SnakeAlt: its
BEST_ACTIVATION), written as .volnet files in the reference's export format and timed with ITS protocol through the `pyrenderer` module
(tools/render_protocol.py: 512 x 512, world step 1 / 256, 64 rotation cameras, GPUTimer around render + extract_color, first frame dropped, :100-140).
Random weights (no checkpoints exist offline; a weight gain keeps the deep stacks alive, fvsrn_amd.synthetic): the time of a frame does not depend on them
beyond the opacity the rays meet, the images are no reproduction of the paper's.  One JSON line per network.
usage: python tools/bench_study_grid.py [--activation ReLU|SnakeAlt] [--width 512 --height 512 --stepsize 0.00390625]"""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fvsrn_amd import synthetic, volnet_io  # noqa: E402
import render_protocol  # noqa: E402

NETWORKS = [(32, 2), (32, 4), (32, 10), (32, 16), (32, 22), (48, 2), (48, 4), (48, 6), (48, 8), (48, 10), (64, 2), (64, 4), (64, 6), (96, 3), (128, 2)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--activation", default="ReLU", choices=["ReLU", "SnakeAlt"])
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--stepsize", type=float, default=1.0 / 256)
    ap.add_argument("--cameras", type=int, default=64)
    ap.add_argument("--only", default=None, help="comma-separated networks, e.g. 128x2,96x3")
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        for C, L in NETWORKS:
            if a.only and "%dx%d" % (C, L) not in a.only.split(","):
                continue
            gain = 1.0 if L < 6 else (2.3 if a.activation == "ReLU" else 2.0)
            vn = synthetic.random_network(C=C, layers=L, activation=a.activation, param=1.0, output_mode="density:direct", grid=(16, 32), seed=1234,
                                          box_min=(-0.5, -0.5, -0.5), fourier_std=1.0, grid_scale=0.01, weight_gain=gain)
            path = os.path.join(tmp, "run_%s_%d_%d.volnet" % (a.activation, C, L))
            open(path, "wb").write(volnet_io.save_volnet(vn))
            args = argparse.Namespace(volnet=path, scene=None, out=None, frames=False, width=a.width, height=a.height, cameras=a.cameras, stepsize=a.stepsize,
                                      timestep=0.0, ensemble=0, texture_tf=False)
            st = render_protocol.run(args)
            print(json.dumps({"network": "%dx%d" % (C, L), "activation": a.activation, "grid": "16 ch x 32^3", "num_parameters": st["num_parameters"],
                              "ms_mean": st["ms_mean"], "ms_std": st["ms_std"], "fps": st["fps"], "width": a.width, "height": a.height, "stepsize": a.stepsize,
                              "protocol": "eval_NetworkConfigsGrid.py:100-140 through pyrenderer (tools/render_protocol.py)"}), flush=True)


if __name__ == "__main__":
    main()
