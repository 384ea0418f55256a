"""
PyTorch-CPU port of the reference's Python SRN path.            *** TEST INFRASTRUCTURE ONLY ***

Used for (a) the ``cpu_baseline`` leg of bench.py ("kind": "port") and (b) cross-checks in tests/.
It restates the op sequence of
  * SceneRepresentationNetwork.forward    applications/volnet/network.py:998-1096
      InputParametrization.forward         :123-169   [x, cos(Bx), sin(Bx), extra]
      F.grid_sample(align_corners=False, padding_mode='border')   :1080-1084
      nn.Linear chain + activation (InnerNetwork :340-420, CustomActivations :239-261)
      OutputParametrization.forward        :204-237
  * Raytracing._full_trace_forward         applications/volnet/raytracing.py:275-329
      one network call per step over ALL rays, Beer-Lambert blend (_blend :159-166),
      box clipping intersection_aabb :79-92, max_steps = int(max(tmax-tmin)/stepsize)
with the same torch ops (matmul / addmm via F.linear, cos/sin, grid_sample), so its speed on the
host cores is what the reference's PyTorch path costs.  The reference itself cannot travel to the
GPU box; tests/test_torch_port.py checks this port against the golden vectors the reference produced.
"""
from __future__ import annotations

import time
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


class TorchSRN(torch.nn.Module):
    def __init__(self, B: np.ndarray, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], activation: str,
                 activation_param: float, output_mode: str, grid: Optional[np.ndarray] = None):
        super().__init__()
        self.register_buffer("B", torch.from_numpy(np.asarray(B, np.float32)))
        self.weights = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(np.asarray(w, np.float32)), False) for w in weights])
        self.biases = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(np.asarray(b, np.float32)), False) for b in biases])
        self.activation, self.p, self.output_mode = activation, float(activation_param), output_mode
        if grid is not None:
            g = torch.from_numpy(np.asarray(grid, np.float32))
            self.register_buffer("grid", g if g.dim() == 5 else g.unsqueeze(0))
        else:
            self.register_buffer("grid", None)

    def act(self, x):
        if self.activation == "ReLU":
            return torch.relu(x)
        if self.activation == "Sine":
            return torch.sin(self.p * x)
        if self.activation == "Snake":
            return x + (1.0 / self.p) * (torch.sin(self.p * x) ** 2)
        if self.activation == "SnakeAlt":
            return (x + 1 - torch.cos(2 * self.p * x)) / (2.0 * self.p)
        raise ValueError(self.activation)

    def forward(self, x, mode: str = "screen"):
        """x: (N,3) positions in the unit box."""
        parts = [x]
        f = torch.matmul(self.B, x.t()).t()
        parts += [torch.cos(f), torch.sin(f)]
        if self.grid is not None:
            gp = x.unsqueeze(0).unsqueeze(1).unsqueeze(1)  # 1,N,1,1,3
            lat = F.grid_sample(self.grid.to(x.dtype), gp * 2 - 1, align_corners=False, padding_mode="border")
            parts.append(lat[0, :, 0, 0, :].t())
        y = torch.cat(parts, dim=1)
        n = len(self.weights)
        for i in range(n):
            y = F.linear(y, self.weights[i].to(y.dtype), self.biases[i].to(y.dtype))
            if i < n - 1:
                y = self.act(y)
        if self.output_mode == "density":
            return torch.sigmoid(y)
        if self.output_mode == "density:direct":
            return torch.clamp(y, 0, 1) if mode == "screen" else y
        rgb, a = y[..., :3], y[..., 3:]
        if self.output_mode == "rgbo":
            rgb, a = torch.sigmoid(rgb), F.softplus(a)
        elif mode == "screen":
            rgb, a = torch.clamp(rgb, 0, 1), torch.clamp(a, min=0)
        return torch.cat((rgb, a), dim=-1)


def camera_rays(eye, right, up, fov_y, W, H):
    """kernel::CameraReferenceFrame::eval (renderer/renderer_camera.cuh:33-52) for every pixel."""
    eye, right, up = [torch.as_tensor(np.asarray(v, np.float32)) for v in (eye, right, up)]
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    ndcx = 2 * (xs + 0.5) / W - 1
    ndcy = 2 * (ys + 0.5) / H - 1
    front = torch.linalg.cross(up, right)
    ty = float(np.tan(fov_y / 2))
    tx = ty * (W / H)
    d = front + (ndcx * tx)[..., None] * right + (ndcy * ty)[..., None] * up
    d = d / torch.linalg.norm(d, dim=-1, keepdim=True)
    return eye.expand_as(d).reshape(-1, 3), d.reshape(-1, 3)


def trace(net: TorchSRN, ray_start, ray_dir, box_min, box_size, stepsize: float, *, tf_identity=None, dtype=torch.float32,
          budget_s: Optional[float] = None):
    """Raytracing._full_trace_forward; tf_identity=(scale_absorption, scale_emission) maps a density network
    through the Identity TF (renderer_tf_identity.cuh:36-54), None = rgbo network.
    Returns (rgba (N,4), number of network samples evaluated = N * steps done).  budget_s stops the step loop
    early (timing runs only: the image is then incomplete)."""
    box_min = torch.as_tensor(np.asarray(box_min, np.float32)).unsqueeze(0)
    box_size = torch.as_tensor(np.asarray(box_size, np.float32)).unsqueeze(0)
    inv = 1.0 / ray_dir
    t135 = (box_min - ray_start) * inv
    t246 = (box_min + box_size - ray_start) * inv
    tmin = torch.max(torch.minimum(t135, t246), dim=1, keepdim=True)[0]
    tmax = torch.min(torch.maximum(t135, t246), dim=1, keepdim=True)[0]
    max_steps = int(torch.max(tmax - tmin).item() / stepsize)
    n = ray_start.shape[0]
    color = torch.zeros((n, 3))
    alpha = torch.zeros((n, 1))
    t_start = time.perf_counter()
    done = 0
    with torch.no_grad():
        for t in range(max_steps):
            if budget_s is not None and t > 0 and time.perf_counter() - t_start > budget_s:
                break
            done += 1
            tcur = tmin + t * stepsize
            pos = ((ray_start + tcur * ray_dir) - box_min) / box_size
            pred = net(pos.to(dtype), "screen").float()
            if tf_identity is not None:
                d = torch.clamp(pred, 0, 1)
                c = torch.cat([d * tf_identity[1]] * 3 + [d * tf_identity[0] * stepsize], dim=1)
            else:
                c = torch.cat([pred[:, :3], pred[:, 3:] * stepsize], dim=1)
            a = 1 - torch.exp(-c[:, 3:])
            a = torch.where(tcur < tmax, a, torch.zeros(1, 1))
            color = color + (1 - alpha) * c[:, :3] * a
            alpha = alpha + (1 - alpha) * a
    return torch.cat((color, alpha), dim=1), n * done


def time_cpu_baseline(net: TorchSRN, eye, right, up, fov_y, box_min, box_size, *, width, height, stepsize, tf_identity,
                      threads: int, budget_s: float = 15.0, dtype=torch.float32):
    """Times the port for at most ~budget_s seconds of step loop; returns dict(value=samples/s, seconds, samples, cores)."""
    torch.set_num_threads(threads)
    rs, rd = camera_rays(eye, right, up, fov_y, width, height)
    trace(net, rs, rd, box_min, box_size, stepsize, tf_identity=tf_identity, dtype=dtype, budget_s=0.5)  # warm up
    t0 = time.perf_counter()
    _, samples = trace(net, rs, rd, box_min, box_size, stepsize, tf_identity=tf_identity, dtype=dtype, budget_s=budget_s)
    dt = time.perf_counter() - t0
    return {"value": samples / dt, "seconds": dt, "samples": samples, "cores": threads}
