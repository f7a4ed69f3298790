"""On-disk formats of the stage-1 -> stage-2 hand-off, readable and writable without nerfstudio
(SURVEY.md section 8f row 4):

* ``gaussian_mask_NxM.npy``  -- ``np.save`` of a bool ``[N,M]`` array, written by
  preprocess/knn_gaussian.py:162-165, read by freegaussian_pipeline.py:45-47;
* ``interflow_n{k}/<frame>.npy`` -- ``np.save`` of the per-pixel flow ``[H,W,2]`` the data parser
  looks up from the image path (freegaussian_dataparser.py:1165), written by
  preprocess/epipolar_flow.py:349-385;
* ``step-%09d.ckpt`` -- nerfstudio's trainer checkpoint: ``torch.save`` of
  ``{"step", "pipeline", "optimizers", "schedulers"}`` where the model's state lives under the
  ``_model.`` prefix of ``"pipeline"`` (``module.`` in front under DDP); this is the layout
  ``FreeGaussianControlModel.load_deformable_checkpoint`` unpacks
  (freegaussian_control_model.py:34-52).  nerfstudio's writer is not readable here; the layout is
  the one that reader implies."""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch


def save_gaussian_mask(data_dir: str, gaussian_masks: torch.Tensor, crop: bool = False) -> str:
    path = os.path.join(data_dir, "gaussian_mask_NxM_crop.npy" if crop else "gaussian_mask_NxM.npy")
    np.save(path, gaussian_masks.detach().cpu().numpy().astype(bool))
    return path


def load_gaussian_mask(data_dir: str, device="cpu") -> torch.Tensor:
    path = os.path.join(data_dir, "gaussian_mask_NxM.npy")
    assert os.path.exists(path), path  # freegaussian_pipeline.py:46
    return torch.from_numpy(np.load(path)).to(device)


def interflow_path(data_dir: str, file_path: str, interval: int) -> str:
    """freegaussian_dataparser.py:1165: './images/xxx.png' -> 'interflow_n{k}/xxx.png.npy'."""
    return os.path.join(data_dir, file_path.replace("./images", f"interflow_n{interval}") + ".npy")


def save_interflow(data_dir: str, file_path: str, interval: int, flow: torch.Tensor) -> str:
    path = interflow_path(data_dir, file_path, interval)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    arr = flow.detach().cpu().numpy()
    assert arr.ndim == 3 and arr.shape[-1] == 2
    np.save(path, arr)
    return path


def load_interflow(data_dir: str, file_path: str, interval: int) -> torch.Tensor:
    return torch.from_numpy(np.load(interflow_path(data_dir, file_path, interval)))


def checkpoint_path(checkpoint_dir: str, step: int) -> str:
    return os.path.join(checkpoint_dir, f"step-{step:09d}.ckpt")


def save_checkpoint(checkpoint_dir: str, step: int, model: torch.nn.Module,
                    optimizers: Optional[Dict[str, torch.optim.Optimizer]] = None) -> str:  # fmt: skip
    os.makedirs(checkpoint_dir, exist_ok=True)
    path = checkpoint_path(checkpoint_dir, step)
    state = {
        "step": step,
        "pipeline": {"_model." + k: v.detach().cpu() for k, v in model.state_dict().items()},
        "optimizers": {k: o.state_dict() for k, o in (optimizers or {}).items()},
        "schedulers": {},
    }
    torch.save(state, path)
    return path


def model_state_from_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """The key surgery of load_deformable_checkpoint (freegaussian_control_model.py:35-51)."""
    loaded = torch.load(path, map_location="cpu", weights_only=False)
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in loaded["pipeline"].items()}
    is_ddp, model_state = True, {}
    for k, v in state.items():
        if k.startswith("_model."):
            model_state[k[len("_model."):]] = v
            if not k.startswith("_model.module."):
                is_ddp = False
    if is_ddp:
        model_state = {k[len("module."):]: v for k, v in model_state.items()}
    return model_state


def load_deformable_checkpoint(model: torch.nn.Module, path: str) -> int:
    """Stage-2 initialisation from a stage-1 checkpoint (``strict=False`` as the reference, :52).
    The Gaussian parameters are re-allocated to the checkpoint's count first -- what the
    reference's ``load_state_dict`` override does for ``gauss_params``.  Returns the step."""
    state = model_state_from_checkpoint(path)
    gp = getattr(model, "gauss_params", None)
    if gp is not None:
        for name in list(gp.keys()):
            key = f"gauss_params.{name}"
            if key in state and state[key].shape != gp[name].shape:
                gp[name] = torch.nn.Parameter(torch.empty_like(state[key], device=gp[name].device))
    model.load_state_dict(state, strict=False)
    return int(torch.load(path, map_location="cpu", weights_only=False)["step"])
