#!/usr/bin/env python3
"""Scene fixtures: the reference's own DVR scene files (applications/config-files/*.json), trimmed to what the hot path reads -- camera, ray
evaluator, selected transfer function, BRDF, blending -- so that the GPU box (no /root/reference) can render them:
    python tests/golden/make_scene_fixtures.py          (build container only)
Data only: the files keep the reference's JSON layout (pyrenderer.load_from_json reads them unchanged); entries of other modules
(rasterization, Monte-Carlo / iso evaluators, phase functions, unselected transfer functions, UI state) are dropped and the path of the
ground-truth volume (a .cvol that is not in the snapshot) is blanked.  tests/test_pyrenderer.py renders each of them with a golden network
in place of the volume -- what the reference's scripts do (inference.py:598) -- against the oracle fed from the same JSON."""
import json
import os

SRC = "/root/reference/applications/config-files"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")
# one per transfer-function kind and stepsize convention, two shaded ones (Phong; Phong + magnitude scaling + 2D pre-integration)
SCENES = ["Miranda-v1-dvr", "plume100-v1-dvr", "RichtmyerMeshkov-t20-v1-dvr", "skull-v5-dvr", "ejecta70-v6-dvr", "LuBerger-Jet-v3-shaded",
          "ejecta1024-v7-shaded"]


def trim(d):
    sel_tf = d["RayEvaluation"]["DVR"]["selectedTF"]
    out = {"root": d.get("root", "Simple"), "version": d.get("version"),
           "ImageEvaluator": {"Simple": d["ImageEvaluator"]["Simple"]},
           "RayEvaluation": {"DVR": d["RayEvaluation"]["DVR"]},
           "camera": {"Sphere": d["camera"]["Sphere"]},
           "tf": {sel_tf: d["tf"][sel_tf]}}
    for k in ("blending", "brdf"):
        if k in d:
            out[k] = d[k]
    sel_vol = d["ImageEvaluator"]["Simple"].get("selectedVolume")
    if sel_vol and "volume" in d and sel_vol in d["volume"] and isinstance(d["volume"][sel_vol], dict):
        v = dict(d["volume"][sel_vol])
        v["volumePath"] = ""
        out["volume"] = {sel_vol: v}
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name in SCENES:
        d = json.load(open(os.path.join(SRC, name + ".json")))
        t = trim(d)
        json.dump(t, open(os.path.join(OUT, name + ".json"), "w"), indent=1, sort_keys=True)
        print("wrote", name, "TF", d["RayEvaluation"]["DVR"]["selectedTF"], os.path.getsize(os.path.join(OUT, name + ".json")), "bytes")


if __name__ == "__main__":
    main()
