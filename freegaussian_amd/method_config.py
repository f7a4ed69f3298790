"""Method specifications of the two plugin entry points (pyproject.toml:14-17 of the reference):

    freegaussian          -> freegaussian_config.py:28-95          (stage 1)
    freegaussian-control  -> DANGLING upstream: pyproject.toml:16 names
                             `freegaussian_config:freegaussian_control_method`, which the file never
                             defines (SURVEY.md §0 finding 4).  Defined here as SURVEY §8f-1 infers
                             it: the stage-1 trainer with the control model, optimizers minus
                             "deform" (freegaussian_control_model.py:215-218), and the stage-1
                             checkpoint + gaussian_mask_NxM.npy passed to the pipeline
                             (freegaussian_pipeline.py:25,43-50).

The tables are plain data so the build's own harness (harness.py) can use them without
nerfstudio; with nerfstudio installed the entry points resolve through ``nerfstudio_adapter``
(pyproject.toml of this repo), which reuses the reference's own ``TrainerConfig`` -- pipeline,
data manager and parser included -- and only rebinds the raster call."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional


@dataclass(frozen=True)
class OptimSpec:
    """Adam group + optional exponential decay (nerfstudio ExponentialDecayScheduler semantics)."""

    lr: float
    eps: float = 1e-15
    lr_final: Optional[float] = None
    max_steps: int = 30000


# reference freegaussian_config.py:48-90
STAGE1_OPTIMIZERS: Dict[str, OptimSpec] = {
    "means": OptimSpec(1.6e-4 * 5, lr_final=1.6e-6 * 5, max_steps=30000),
    "features_dc": OptimSpec(0.0025),
    "features_rest": OptimSpec(0.0025 / 20),
    "opacities": OptimSpec(0.05),
    "scales": OptimSpec(0.001 * 5),
    "quats": OptimSpec(0.001),
    "deform": OptimSpec(1.6e-4 * 5, lr_final=1.6e-6, max_steps=30000),
    "control": OptimSpec(1.6e-4 * 5, lr_final=1.6e-6, max_steps=15000),
    # NOT in the reference's table: its `use_bilateral_grid` switch adds a "bilateral_grid" parameter group
    # (freegaussian_model.py:617-618) that its own method spec has no optimizer for, so the switch cannot train upstream.
    # This is the entry of nerfstudio's splatfacto method, which the model class inherits the branch from (without its
    # 1000-step warm-up); harness.build_optimizers uses it only when the model has the group.
    "bilateral_grid": OptimSpec(2e-3, lr_final=1e-4, max_steps=30000),
}
STAGE2_OPTIMIZERS: Dict[str, OptimSpec] = {k: v for k, v in STAGE1_OPTIMIZERS.items() if k != "deform"}

TRAINER = dict(steps_per_eval_image=100, steps_per_eval_batch=0, steps_per_save=2000, steps_per_eval_all_images=1000,
               max_num_iterations=30000, mixed_precision=False)  # fmt: skip  (freegaussian_config.py:30-36)

METHODS = {
    "freegaussian": dict(model="FreeGaussianModel", optimizers=STAGE1_OPTIMIZERS, trainer=TRAINER,
                         description="FreeGaussian model for dynamic scenes with lang control"),
    "freegaussian-control": dict(model="FreeGaussianControlModel", optimizers=STAGE2_OPTIMIZERS, trainer=TRAINER,
                                 description="FreeGaussian stage 2: control MLP on masked Gaussians "
                                             "(needs --pipeline.load-deformable-checkpoint and gaussian_mask_NxM.npy)"),
}  # fmt: skip


def nerfstudio_method_specs():
    """MethodSpecification objects for both entry points -- complete ``TrainerConfig``s including
    ``pipeline=`` (the reference's FreeGaussianPipelineConfig with its data manager / parser, the
    model's raster call rebound to this package): ``nerfstudio_adapter`` builds them from the
    reference's own spec.  Requires nerfstudio and the reference package to be importable."""
    from . import nerfstudio_adapter as A

    return {name: getattr(A, attr) for name, attr in A.METHOD_ENTRY_POINTS.items()}
