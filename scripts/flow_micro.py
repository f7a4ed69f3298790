"""Micro-benchmark of the flow-derivative kernels: fg_camera_flow at 1080p, fg_flow_fwd/bwd at 1M."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
H, W, N = 1080, 1920, 1_000_000
depth = torch.rand(H, W, device=dev) * 5 + 1
K = torch.tensor([[1200.0, 0, 960], [0, 1200.0, 540], [0, 0, 1]], device=dev)
v, w = torch.tensor([0.02, 0.0, 0.01], device=dev), torch.tensor([0.0, 0.01, 0.0], device=dev)
m2 = (torch.rand(N, 2, device=dev) * torch.tensor([W, H], device=dev)).requires_grad_(True)
d = (torch.rand(N, device=dev) * 5 + 1).requires_grad_(True)
vel = torch.randn(N, 3, device=dev).requires_grad_(True)
radii = torch.ones(N, dtype=torch.int32, device=dev)
g1, g2 = torch.randn(N, 2, device=dev), torch.randn(N, 2, device=dev)


def step():
    ops.camera_flow(depth, K, v, w)
    a, b = ops.gaussian_flow(m2, d, vel, K, v, w, radii)
    torch.autograd.backward([a, b], [g1, g2])
    m2.grad = d.grad = vel.grad = None


for _ in range(5):
    step()
torch.cuda.synchronize()
ops.default_context.stage_timer = ops.StageTimer()
for _ in range(30):
    step()
st = ops.default_context.stage_timer.summary()
ops.default_context.stage_timer = None
bytes_ = {"fg_camera_flow": H * W * (4 + 8), "fg_flow_fwd": N * (8 + 4 + 12 + 4 + 16), "fg_flow_bwd": N * (8 + 4 + 12 + 4 + 16 + 8 + 4 + 12)}
for k, ms in st.items():
    print(f"{k}: {ms * 1e3:.1f} us, {bytes_[k] / 1e6:.0f} MB algorithmic -> {bytes_[k] / ms / 1e6:.0f} GB/s")
