"""Every collective call site through a 1-rank ``nccl`` (= RCCL) process group on the box's one GPU
(scripts/rccl_one_rank.py; the reference's only distributed hook is the DDP pass-through of
freegaussian_pipeline.py:36-40, :62)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_forced_collectives_match_the_step_without_them_on_both_streams():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), FG_ONE_RANK_STEPS="20")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_one_rank.py")], capture_output=True, text=True, env=env, timeout=900)
    assert res.stdout.strip(), res.stderr[-2000:]
    # (RCCL prints its version banner to stdout when the process exits: the result is the last line that is JSON)
    out = json.loads([ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")][-1])
    print(json.dumps(out))
    assert res.returncode == 0 and out["ok"], out["mismatches"]
    assert out["backend"] == "nccl" and out["world"] == 1 and out["head_slices"] == int(os.environ.get("FG_DP_HEAD_SLICES", "4"))
    for leg in ("default_stream", "side_stream"):
        assert out[leg]["model_sparse_info"]["overflows"] >= 1 and out[leg]["model_sparse_info"]["sparse_steps"] >= 10
        assert out[leg]["model_per_view_means_info"]["densify_stats_identity"]
