"""refinement_after at 1M Gaussians: HIP passes vs the reference's torch op sequence."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_densify import _setup  # noqa: E402
from densify_torch_sequence import refine_torch  # noqa: E402
from freegaussian_amd.densify import refinement_after  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for fused in (True, False, True, False):
    model, opts = _setup(n=n, step=3500, device="cuda")
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    out = refinement_after(model, opts, 3500, 60, refine=None if fused else refine_torch)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print(f"fused={fused}: {dt:.2f} ms, {out['before']} -> {out['after']} Gaussians, "
          f"peak extra memory {(torch.cuda.max_memory_allocated() - base) / 1e6:.0f} MB")
