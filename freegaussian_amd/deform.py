"""Deformation / control MLPs that feed the rasterizer (dense GEMMs: they stay on PyTorch-ROCm /
hipBLASLt).  Behaviour and ``state_dict`` key names follow the reference's
``FreeGaussianDeformableModel`` / ``FreeGaussianControllableModel``
(freegaussian/freegaussian_model.py:1054-1145) so stage-1 checkpoints load unchanged; outputs are
checked against golden vectors produced by the reference classes (tests/golden/g_mlp.npz)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .utils import exp_se3, positional_encoding


def _trunk(in_ch: int, width: int, depth: int, skip_at: int) -> nn.ModuleList:
    """depth linears of `width`; the one AFTER index `skip_at` also takes the re-injected input."""
    layers = [nn.Linear(in_ch, width)]
    for i in range(depth - 1):
        layers.append(nn.Linear(width + in_ch if i == skip_at else width, width))
    return nn.ModuleList(layers)


class _TallLinear(torch.autograd.Function):
    """x [N,in] -> x W^T + b for N in the hundreds of thousands.  Same library GEMMs forward and for the input
    gradient; the WEIGHT gradient go^T x is a [out, N] x [N, in] product -- 256 x 256 outputs, N-long dot products --
    which the BLAS runs as one small-tile kernel at a fraction of its rate (three quarters of the MLP's backward on an
    MI355X: 13.6 ms of 18 at 300k Gaussians).  Here it is a batched product over chunks of 8192 rows (shapes the library
    is good at) and a sum of the partial [out, in] matrices."""

    CHUNK = 8192

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        go = go.contiguous()
        gx = go @ weight if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            N, C = x.shape[0], _TallLinear.CHUNK
            B = N // C
            n0 = B * C
            gw = torch.bmm(go[:n0].view(B, C, -1).transpose(1, 2), x[:n0].view(B, C, -1)).sum(0)
            if n0 < N:
                gw = gw + go[n0:].t() @ x[n0:]
        gb = go.sum(0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


def _linear(layer: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    if x.is_cuda and x.dim() == 2 and x.shape[0] >= 4 * _TallLinear.CHUNK and layer.bias is not None and x.is_contiguous():
        return _TallLinear.apply(x, layer.weight, layer.bias)
    return layer(x)


def _run_trunk(layers: nn.ModuleList, inp: torch.Tensor, skip_at: int) -> torch.Tensor:
    h = inp
    for i, layer in enumerate(layers):
        h = torch.relu(_linear(layer, h))
        if i == skip_at:
            h = torch.cat([inp, h], dim=-1)
    return h


class FreeGaussianDeformableModel(nn.Module):
    """(x [N,3], t [N,1]) -> (SE(3) per Gaussian [N,4,4], d_rotation [N,4], d_scaling [N,3])."""

    def __init__(self, D: int = 8, W: int = 256, multires: int = 10, is_blender: bool = False):
        super().__init__()
        self.D, self.W, self.is_blender = D, W, is_blender
        self.multires = multires
        self.t_multires = 6 if is_blender else 10
        self.skip_at = D // 2
        xyz_ch = 3 * (1 + 2 * multires)
        t_ch = 1 + 2 * self.t_multires
        if is_blender:
            self.time_out = 30
            self.timenet = nn.Sequential(nn.Linear(t_ch, 256), nn.ReLU(inplace=True), nn.Linear(256, self.time_out))
            t_ch = self.time_out
        self.input_ch = xyz_ch + t_ch
        self.linear = _trunk(self.input_ch, W, D, self.skip_at)
        self.branch_w = nn.Linear(W, 3)
        self.branch_v = nn.Linear(W, 3)
        self.gaussian_rotation = nn.Linear(W, 4)
        self.gaussian_scaling = nn.Linear(W, 3)

    def forward(self, x: torch.Tensor, t: torch.Tensor):
        t_emb = positional_encoding(t, self.t_multires)
        if self.is_blender:
            t_emb = self.timenet(t_emb)
        inp = torch.cat([positional_encoding(x, self.multires), t_emb], dim=-1)
        h = _run_trunk(self.linear, inp, self.skip_at)
        w, v = _linear(self.branch_w, h), _linear(self.branch_v, h)
        theta = w.norm(dim=-1, keepdim=True)
        # the reference adds 1e-5 AFTER the division (freegaussian_model.py:1106-1107)
        screw = torch.cat([w / theta + 1e-5, v / theta + 1e-5], dim=-1)
        return exp_se3(screw, theta), _linear(self.gaussian_rotation, h), _linear(self.gaussian_scaling, h)


class FreeGaussianControllableModel(nn.Module):
    """(control points [M,3], control value [M,3]) -> (d_xyz [M,3], d_rot [M,4], d_scale [M,3])."""

    def __init__(self, D: int = 8, W: int = 256, multires: int = 10):
        super().__init__()
        self.D, self.W, self.multires = D, W, multires
        self.skip_at = D // 2
        self.input_ch = 2 * 3 * (1 + 2 * multires)
        self.linear = _trunk(self.input_ch, W, D, self.skip_at)
        self.d_xyz = nn.Linear(W, 3)
        self.d_scale = nn.Linear(W, 3)
        self.d_rot = nn.Linear(W, 4)

    def forward(self, x: torch.Tensor, value: torch.Tensor):
        inp = torch.cat([positional_encoding(x, self.multires), positional_encoding(value, self.multires)], dim=-1)
        h = _run_trunk(self.linear, inp, self.skip_at)
        return _linear(self.d_xyz, h), _linear(self.d_rot, h), _linear(self.d_scale, h)
