#!/usr/bin/env python3
"""
The reference's inference-timing protocol on the MI355X build, through the `pyrenderer` drop-in module.

What it reproduces (reference files):
  * applications/volnet/eval_NetworkConfigsGrid.py:100-140 -- 512x512, world step size 1/256, 64 cameras rotating
    around the object, `pyrenderer.GPUTimer` around render + extract_color, first frame discarded, mean +- std in ms
  * applications/volnet/inference.py:383-401 (`get_rotation_cameras`: yaw offsets linspace(0, 2 pi, N, endpoint=False)
    on the default pitch / yaw / distance, cameras handed over as `camera.get_parameters()` tensors) and :589-619
    (`render_network`, TensorCores branch: set volume / step size / time+ensemble / camera parameters, `render`,
    `extract_color`)
  * LoadedModel.convert_image (:627-631) + imageio.imwrite -> PNG frames (written here with zlib, imageio is not installed)

Inputs: a `.volnet` file (the reference's own export format, loaded unchanged) -- or an uncompressed `.cvol` grid volume for the ground truth -- and optionally a scene JSON of the
reference (`pyrenderer.load_from_json`); without one a default DVR scene (Identity TF) is used.

  python tools/render_protocol.py net.volnet [--scene scene.json] [--out outdir] [--width 512 --height 512]
         [--cameras 64] [--stepsize 0.00390625] [--timestep T --ensemble E] [--frames]
Prints one JSON object with the statistics; --out also writes stats.json and (with --frames) frameNNN.png.
"""
import argparse
import json
import math
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fv-srn_amd", "pyrenderer"))


def write_png(path: str, rgba8: np.ndarray) -> None:
    """(H,W,4) uint8 -> PNG (8-bit RGBA, no interlace)."""
    h, w, c = rgba8.shape
    assert c == 4 and rgba8.dtype == np.uint8
    raw = b"".join(b"\x00" + rgba8[y].tobytes() for y in range(h))

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def convert_image(img) -> np.ndarray:
    """LoadedModel.convert_image (inference.py:627-631): (1,4,H,W) float -> (H,W,4) uint8."""
    a = img[0].detach().cpu().numpy()
    a = np.nan_to_num(a, nan=0.0)
    return (np.clip(a.transpose(1, 2, 0), 0, 1) * 255).astype(np.uint8)


def default_evaluator(pr):
    ev = pr.ImageEvaluatorSimple()
    ev.camera.orientation = pr.CameraOnASphere.Ym
    ev.camera.pitchYawDistance.value = pr.double3(0.4, 0.0, 1.6)
    ev.camera.fov_y_radians = math.radians(45.0)
    tf = pr.TransferFunctionIdentity()
    tf.absorption_emission.value = pr.double2(10.0, 1.0)
    ev.ray_evaluator.tf = tf
    ev.ray_evaluator.early_out = True
    return ev


def run(args) -> dict:
    import torch
    import pyrenderer as pr
    if not torch.cuda.is_available():
        raise SystemExit("no GPU: the renderer has no CPU path")
    ev = pr.load_from_json(args.scene) if args.scene else default_evaluator(pr)
    if args.volnet.endswith(".cvol"):  # ground truth: the grid volume itself (the scene files' own "Grid" volume)
        net = None
        vol = pr.VolumeInterpolationGrid()
        vol.setSource(pr.Volume(args.volnet))
    else:
        net = pr.SceneNetwork.load(args.volnet)
        vol = pr.VolumeInterpolationNetwork()
        vol.set_network(net)
    ev.volume = vol
    ev.ray_evaluator.stepsize = args.stepsize  # world step size (render_network :603-606)
    if getattr(args, "texture_tf", False):  # LoadedModel.enable_preintegration(..., convert_to_texture=True), inference.py:332-336
        ev.ray_evaluator.convert_to_texture_tf()
    if net is not None:
        net.set_time_and_ensemble(args.timestep, args.ensemble)

    pyd = ev.camera.pitchYawDistance.value
    pitch, yaw, dist = pyd.x, pyd.y, pyd.z
    cameras = []
    for off in np.linspace(0, 2 * np.pi, args.cameras, endpoint=False):  # get_rotation_cameras :383-401
        ev.camera.pitchYawDistance.value = pr.double3(pitch, yaw + float(off), dist)
        cameras.append(ev.camera.get_parameters().clone())

    timer = pr.GPUTimer()
    times = []
    if args.out:
        os.makedirs(args.out, exist_ok=True)
    for i, cam in enumerate(cameras):
        timer.start()
        ev.camera.set_parameters(cam)
        img = ev.render(args.width, args.height)
        rgba = ev.extract_color(img)
        timer.stop()
        if i > 0:  # eval_NetworkConfigsGrid.py:131-133: the first frame is discarded
            times.append(timer.elapsed_milliseconds())
        if args.out and args.frames:
            write_png(os.path.join(args.out, "frame%03d.png" % i), convert_image(rgba))
    t = np.asarray(times, np.float64)
    stats = {"protocol": "eval_NetworkConfigsGrid.py:100-140 (rotation cameras, GPUTimer around render + extract_color, "
                         "first frame discarded)",
             "volnet": os.path.basename(args.volnet), "width": args.width, "height": args.height, "stepsize": args.stepsize,
             "num_cameras": args.cameras, "num_parameters": net.num_parameters() if net is not None else None,
             "ms_mean": float(t.mean()) if len(t) else None, "ms_std": float(t.std()) if len(t) else None,
             "fps": float(1000.0 / t.mean()) if len(t) else None}
    if args.out:
        with open(os.path.join(args.out, "stats.json"), "w") as f:
            json.dump(stats, f, indent=1)
    return stats


def main(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("volnet")
    p.add_argument("--scene", default=None, help="scene JSON of the reference (config-files/*.json)")
    p.add_argument("--out", default=None)
    p.add_argument("--frames", action="store_true", help="write every frame as PNG into --out")
    p.add_argument("--width", type=int, default=512)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--cameras", type=int, default=64)
    p.add_argument("--stepsize", type=float, default=1.0 / 256)
    p.add_argument("--timestep", type=float, default=0.0)
    p.add_argument("--ensemble", type=int, default=0)
    p.add_argument("--texture-tf", action="store_true", help="convert the scene's TF to a 256-texel texture TF first (convert_to_texture_tf)")
    args = p.parse_args(argv)
    print(json.dumps(run(args)))


if __name__ == "__main__":
    main()
