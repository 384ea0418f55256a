"""The PyTorch-CPU port used as bench.py's cpu_baseline reproduces the reference's own outputs."""
import numpy as np
import pytest
import torch

import util
from oracle import torch_port


def port_from_golden(d, meta):
    n = len(meta["layers"].split(":")) + 1
    return torch_port.TorchSRN(d["B"], [d["W%d" % i] for i in range(n)], [d["b%d" % i] for i in range(n)],
                               meta["activation"], meta["activation_param"], meta["output_mode"], d.get("grid"))


@pytest.mark.parametrize("name", ["g1_c32l4_relu_density", "g1_c32l4_snakealt_rgbo", "g1_c32l4_snake_rgbo-direct",
                                  "g1_c32l4_sine_density-direct", "g1_c32l4_grid16r8_snakealt_rgbo",
                                  "g1_c64l6_grid16r8_snakealt_density-direct"])
def test_port_matches_reference_forward(name):
    d, meta = util.load_golden(name)
    net = port_from_golden(d, meta)
    with torch.no_grad():
        out = net(torch.from_numpy(d["positions"]), "world").numpy()
    assert np.abs(out - d["out_fp32"]).max() < 2e-5


def test_port_trace_matches_reference_trace():
    d, meta = util.load_golden("g3_trace_rgbo_32x32")
    net = port_from_golden(d, meta)
    rs = torch.from_numpy(d["ray_start"]).reshape(-1, 3)
    rd = torch.from_numpy(d["ray_dir"]).reshape(-1, 3)
    rgba, n = torch_port.trace(net, rs, rd, meta["box_min"], meta["box_size"], meta["stepsize"])
    img = rgba.reshape(meta["H"], meta["W"], 4).permute(2, 0, 1).numpy()
    assert np.abs(img - d["image"]).max() < 1e-5
    assert n == 32 * 32 * int(np.sqrt(3) / meta["stepsize"])
