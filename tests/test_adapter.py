"""The plugin surface (reference pyproject.toml:14-21) without nerfstudio: entry-point names,
the gsplat module shims and the rebinding of the reference's raster imports, exercised on a
stand-in package that imports exactly what freegaussian_model.py:15-21 imports."""
import importlib
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pyproject_declares_the_five_reference_entry_points():
    import tomli

    cfg = tomli.load(open(os.path.join(ROOT, "pyproject.toml"), "rb"))
    eps = cfg["project"]["entry-points"]
    from freegaussian_amd import nerfstudio_adapter as A

    assert eps["nerfstudio.method_configs"] == {k: f"freegaussian_amd.nerfstudio_adapter:{v}" for k, v in A.METHOD_ENTRY_POINTS.items()}
    assert eps["nerfstudio.dataparser_configs"] == {k: f"freegaussian_amd.nerfstudio_adapter:{v}" for k, v in A.DATAPARSER_ENTRY_POINTS.items()}
    assert set(eps["nerfstudio.method_configs"]) == {"freegaussian", "freegaussian-control"}
    assert set(eps["nerfstudio.dataparser_configs"]) == {"freegaussian-conerf-data", "freegaussian-real-data", "freegaussian-sim-data"}


def test_entry_points_fail_loudly_without_nerfstudio():
    from freegaussian_amd import nerfstudio_adapter as A
    from freegaussian_amd.method_config import nerfstudio_method_specs

    if importlib.util.find_spec("nerfstudio") is not None:
        pytest.skip("nerfstudio is installed here")
    for name in list(A.METHOD_ENTRY_POINTS.values()) + list(A.DATAPARSER_ENTRY_POINTS.values()):
        with pytest.raises(ImportError, match="nerfstudio"):
            getattr(A, name)
    with pytest.raises(ImportError):
        nerfstudio_method_specs()
    with pytest.raises(AttributeError):
        A.no_such_entry_point


def test_install_shims_gsplat_and_rebinds_the_reference_imports(tmp_path, monkeypatch):
    """A stand-in `freegaussian` package with the reference's three gsplat imports
    (freegaussian_model.py:15,18,21) becomes importable on a box without gsplat, and its names are
    bound to this package's functions."""
    import freegaussian_amd as fa
    from freegaussian_amd import nerfstudio_adapter as A

    if importlib.util.find_spec("gsplat") is not None:
        pytest.skip("gsplat is installed here")
    pkg = tmp_path / "freegaussian"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "freegaussian_model.py").write_text(textwrap.dedent("""
        from gsplat.cuda_legacy._torch_impl import quat_to_rotmat
        from gsplat.rendering import rasterization
        from gsplat.cuda_legacy._wrapper import num_sh_bases
        def dim_sh(d):
            return num_sh_bases(d)
    """))
    monkeypatch.syspath_prepend(str(tmp_path))
    saved = {k: v for k, v in sys.modules.items() if k.startswith(("gsplat", "freegaussian."))or k == "freegaussian"}
    try:
        with pytest.raises(ImportError):
            importlib.import_module("freegaussian.freegaussian_model")
        sys.modules.pop("freegaussian.freegaussian_model", None)
        A.install(force=True)
        m = importlib.import_module("freegaussian.freegaussian_model")
        assert m.rasterization is fa.rasterization and m.quat_to_rotmat is fa.quat_to_rotmat
        assert m.dim_sh(3) == 16
        # a module imported BEFORE install (bound to something else) is rebound
        m.rasterization = None
        A.install(force=True)
        assert m.rasterization is fa.rasterization
    finally:
        for k in [k for k in sys.modules if k.startswith("gsplat") or k.startswith("freegaussian.") or k == "freegaussian"]:
            del sys.modules[k]
        sys.modules.update(saved)
