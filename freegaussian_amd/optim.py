"""Adam for the Gaussian parameter groups with the update of a tensor in ONE launch (csrc/adam.hip).

``FusedAdam`` IS a ``torch.optim.Adam`` -- same constructor, same ``param_groups`` / ``state`` layout
(``state[p] = {"step", "exp_avg", "exp_avg_sq"}``), same ``state_dict`` -- so the schedules (``harness.apply_schedules``
writes ``group["lr"]``), the densification's surgery on the optimizer state (``densify.py``: rows removed, duplicated,
zeroed) and the checkpoints (``io.py``) work on it unchanged; only ``step()`` is replaced.  torch's step of the six
groups of the bench scene (59 floats x 1M Gaussians) takes 1.42 ms on an MI355X, as long as render + loss + backward
together; by bytes it is a 0.3 ms job (``scripts/train_step_bench.py``).

The reference attaches ``AdamOptimizerConfig(lr=..., eps=1e-15)`` to every group (freegaussian_config.py); betas,
weight decay and amsgrad stay at torch's defaults, which is what this step implements.  Anything else raises."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _call, _ptr, _stream


class FusedAdam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise _lib.FgRasterError("FusedAdam implements torch.optim.Adam's defaults (no weight decay / amsgrad / maximize)")
            beta1, beta2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()) or g.is_sparse:
                    raise _lib.FgRasterError("FusedAdam updates dense contiguous float32 parameters on the GPU")
                st = self.state[p]
                if len(st) == 0:  # (as torch: a host-side step counter, moments like the parameter)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):  # (state surgery may leave views behind)
                    m, v = st["exp_avg"], st["exp_avg_sq"] = m.contiguous(), v.contiguous()
                g = g.contiguous() if g.dtype == torch.float32 else g.float().contiguous()
                if g.data_ptr() % 16:  # (a view into a flat gradient buffer -- viewdp.FlatGaussianParams -- at an odd offset)
                    g = g.clone()
                _call("fg_adam_step", p.numel(), _ptr(p), _ptr(g), _ptr(m), _ptr(v), lr, float(beta1), float(beta2), eps,
                      int(st["step"]), _stream(), stage="fg_adam_step")  # fmt: skip
        return loss
