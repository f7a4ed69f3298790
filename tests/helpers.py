"""Shared comparison helpers for the parity tests."""
import torch

# Floating-point tolerance of the parity bar (BASELINE.json north_star: "within 1e-4 rel fp32").
REL_TOL = 1e-4


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|  (scale-relative max error)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / max(b.norm().item(), 1e-30)).item()


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    mse = ((a.detach().double().cpu() - b.detach().double().cpu()) ** 2).mean().item()
    return float("inf") if mse == 0 else -10.0 * torch.log10(torch.tensor(mse)).item()
