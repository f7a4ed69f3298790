"""bench.py's own launcher (`--gpus N` without WORLD_SIZE): N rank processes, started before the
parent touches the GPU; runs on CPU through the FG_BENCH_ECHO self-test leg."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_environments_are_the_env_rendezvous_contract():
    import bench

    envs = bench.rank_environments(4, 29511, base={"PATH": "/usr/bin"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)  # dmabuf IPC for RCCL


def test_gpus_flag_spawns_that_many_ranks_and_relays_rank0():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FG_BENCH_ECHO"] = "1"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "2"],
                         env=env, capture_output=True, text=True, timeout=120)  # fmt: skip
    assert res.returncode == 0, res.stderr
    lines = [json.loads(x) for x in res.stdout.strip().splitlines()]
    assert len(lines) == 1  # rank 0 only on stdout
    assert lines[0]["RANK"] == "0" and lines[0]["WORLD_SIZE"] == "3" and lines[0]["n_gpus"] == 3
    others = [json.loads(x) for x in res.stderr.strip().splitlines() if x.startswith("{")]
    assert sorted(o["RANK"] for o in others) == ["1", "2"]
    assert all(o["WORLD_SIZE"] == "3" and o["MASTER_PORT"] == lines[0]["MASTER_PORT"] for o in others)


def test_launcher_does_not_import_the_package_or_touch_the_gpu():
    """The parent must decide and spawn before libfgraster.so is mapped or HIP is initialised."""
    code = (
        "import sys, os; sys.argv=['bench.py','--gpus','2']; os.environ['FG_BENCH_ECHO']='1';"
        "os.environ.pop('WORLD_SIZE', None);"
        f"sys.path.insert(0, {ROOT!r}); import bench\n"
        "try:\n    bench.main(['--gpus','2'])\nexcept SystemExit as e:\n    assert e.code == 0, e.code\n"
        "assert 'freegaussian_amd' not in sys.modules and 'freegaussian_amd._lib' not in sys.modules\n"
        "import torch; assert not torch.cuda.is_initialized()\n"
    )
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr


def test_nccl_launch_refuses_to_share_devices():
    """--gpus N under the RCCL backend with fewer than N visible GPUs exits non-zero with a message
    (this container has none)."""
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("needs a box with fewer than 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FG_BENCH_BACKEND", "FG_BENCH_ECHO")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=120)  # fmt: skip
    assert res.returncode == 2 and "only" in res.stderr and res.stdout.strip() == ""
