"""Micro-benchmark: fg_preprocess_{fwd,bwd} vs fg_preprocess_raw_{fwd,bwd} on the bench scene."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from freegaussian_amd import ops  # noqa: E402
from freegaussian_amd.scenes import synthetic_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
sc = synthetic_scene(n, 1920, 1080, n_views=8, sh_degree=3, seed=42)
dev = torch.device("cuda", 0)
vm, K = sc.viewmats[0].to(dev), sc.Ks[0].to(dev)
act = [t.to(dev).requires_grad_(True) for t in (sc.means, sc.quats, sc.scales, sc.opacities, sc.colors)]
raw = [t.to(dev).requires_grad_(True) for t in (sc.means, sc.quats * 1.3, sc.scales.log(),
                                                 torch.logit(sc.opacities.clamp(1e-6, 1 - 1e-6)),
                                                 sc.colors[:, 0].contiguous(), sc.colors[:, 1:].contiguous())]
vs = torch.randn(n, 16, device=dev)


def run_act():
    out = ops.preprocess(*act, None, vm, K, 1920, 1080, sh_degree=3)
    torch.autograd.backward([out[5]], [vs])


def run_raw():
    out = ops.preprocess_raw(raw[0], raw[1], raw[2], raw[3], raw[4], raw[5], vm, K, 1920, 1080, 3)
    torch.autograd.backward([out[5]], [vs])


for f in (run_act, run_raw):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ops.default_context.stage_timer = ops.StageTimer()
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    print(f.__name__, {k: round(v, 4) for k, v in ops.default_context.stage_timer.summary().items()})
    ops.default_context.stage_timer = None
