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

import ctypes

import torch

from . import _lib
from .ops import _call, _ptr, _stream


class _AdamTensor(ctypes.Structure):  # fg_adam_tensor (include/fgraster.h)
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int64), ("lr", ctypes.c_double), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", ctypes.c_double), ("step", ctypes.c_int64)]


class FusedAdam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = self._collect()
        if work:
            _launch(work)
        return loss

    def _collect(self):
        """Return [(p, g, m, v, lr, beta1, beta2, eps, step)] of this optimizer's update and advance the step counters --
        after everything has been validated, so an error leaves no counter ahead of its moments.  The tensors in the list
        are kept alive by the caller until the launch is enqueued."""
        todo = []
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise _lib.FgRasterError("FusedAdam implements torch.optim.Adam's defaults (no weight decay / amsgrad / maximize)")
            beta1, beta2 = group["betas"]
            lr, eps = float(group["lr"]), float(group["eps"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()) or p.grad.is_sparse:
                    raise _lib.FgRasterError("FusedAdam updates dense contiguous float32 parameters on the GPU")
                if p.data_ptr() % 16:
                    raise _lib.FgRasterError("FusedAdam: a parameter's storage must start on a 16-byte boundary (a view into a "
                                             "flat buffer at an odd offset: pad the views, as viewdp.ModelViewDP does)")
                todo.append((p, lr, float(beta1), float(beta2), eps))
        work = []
        for p, lr, beta1, beta2, eps in todo:
            st = self.state[p]
            if len(st) == 0:  # (as torch: a host-side step counter, moments like the parameter)
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if not (m.is_contiguous() and v.is_contiguous()) or m.data_ptr() % 16 or v.data_ptr() % 16:
                # (state surgery may leave views behind)
                m, v = st["exp_avg"], st["exp_avg_sq"] = m.contiguous().clone(), v.contiguous().clone()
            g = p.grad
            g = g.contiguous() if g.dtype == torch.float32 else g.float().contiguous()
            if g.data_ptr() % 16:  # (a view into a flat gradient buffer -- viewdp.FlatGaussianParams -- at an odd offset)
                g = g.clone()
            st["step"] += 1
            work.append((p, g, m, v, lr, beta1, beta2, eps, int(st["step"])))
        return work


def _launch(work) -> None:
    arr = (_AdamTensor * len(work))()
    for a, (p, g, m, v, lr, b1, b2, eps, step) in zip(arr, work):
        a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
        a.n, a.lr, a.beta1, a.beta2, a.eps, a.step = p.numel(), lr, b1, b2, eps, step
    _call("fg_adam_step_multi", len(work), ctypes.cast(arr, ctypes.c_void_p), _stream(), stage="fg_adam_step")


def step_all(optimizers) -> None:
    """``step()`` of every optimizer; the tensors of all ``FusedAdam`` instances among them go into ONE launch (every 16
    tensors one: ``fg_adam_step_multi``) -- the reference's six Gaussian parameter groups are six optimizers of one tensor
    each, and at its low resolutions an iteration is bound by launches.  This is the harness's own shortcut: it goes
    past ``Optimizer.step``'s wrapper, so step pre/post hooks registered on a FusedAdam do not run here (a trainer that
    relies on them calls ``FusedAdam.step()`` per optimizer, as INTEGRATION.md shows for nerfstudio); ``_opt_called``
    is set, so torch's LR schedulers see the call order they check for."""
    work, others = [], []
    for o in optimizers:
        if isinstance(o, FusedAdam):
            work += o._collect()
            o._opt_called = True
        else:
            others.append(o)
    if work:
        _launch(work)
    for o in others:
        o.step()
