"""CPU checks: the C-ABI library loads and exports exactly what include/fgraster.h declares."""
import os
import re
import subprocess

import pytest

from freegaussian_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "fgraster.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    syms = _header_symbols()
    assert len(syms) >= 19
    lib = _lib.load()
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = set(re.findall(r" T (fg_\w+)", out))
    for s in syms:
        assert s in exported, f"{s} declared in fgraster.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
        assert getattr(lib, s) is not None
    assert set(_lib.SIGNATURES) == set(syms)


def test_abi_version_and_error_strings():
    lib = _lib.load()
    assert lib.fg_abi_version() == _lib.ABI_VERSION
    assert lib.fg_error_string(0) == b"ok"
    assert b"invalid" in lib.fg_error_string(-1)
    with pytest.raises(_lib.FgRasterError):
        _lib.check(-3, "x")


def test_workspace_queries_need_no_gpu():
    lib = _lib.load()
    assert lib.fg_scan_workspace_bytes(1_000_000) >= 8 * (1_000_000 // 2048)
    n = 5_000_000
    assert lib.fg_sort_workspace_bytes(n) >= n * 12


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch, so this is safe on a CPU-only box."""
    lib = _lib.load()
    assert lib.fg_project_fwd(-1, *([None] * 5), 1, 1, 0.3, 0.01, 1e10, 0.0, 16, *([None] * 7)) == -1
    assert lib.fg_raster_fwd(3, 64, 64, 8, *([None] * 7)) in (-1, -4)
    assert lib.fg_sh_fwd(10, 4, 16, *([None] * 6)) == -1
    assert lib.fg_sort_pairs(10, None, None, 70, None, 0, None) == -1


def test_product_does_not_import_oracle():
    """The product path must never route through the CPU oracle."""
    pkg = os.path.join(ROOT, "freegaussian_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "raster_oracle" not in src, f


def test_ops_refuse_cpu_tensors():
    import torch

    from freegaussian_amd import rasterization

    N = 4
    with pytest.raises(_lib.FgRasterError):
        rasterization(torch.zeros(N, 3), torch.ones(N, 4), torch.ones(N, 3), torch.ones(N), torch.ones(N, 3),
                      torch.eye(4)[None], torch.eye(3)[None], 32, 32, packed=False)  # fmt: skip
