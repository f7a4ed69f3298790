import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The CPU oracle is thousands of small tensor ops; on a many-core host (the GPU box has 256)
    # torch's default thread count makes every one of them a 256-way fork/join and the oracle
    # several times slower.  Cap it.
    import torch

    torch.set_num_threads(min(os.cpu_count() or 1, 16))


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built once per session if it is not there yet (hipcc cross-compiles
    without a GPU)."""
    lib = os.path.join(ROOT, "freegaussian_amd", "libfgraster.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g

        g.build()
    # The process-wide default context takes its launch policy from the environment when it is first used: make that
    # moment NOW, not inside the first test that happens to touch it with FG_RASTER_* variables monkeypatched (the
    # patched policy would then be what `monkeypatch` restores for the rest of the session: order-dependent tests).
    from freegaussian_amd import ops

    ops.default_context
    yield


def pytest_sessionfinish(session, exitstatus):
    """FG_PARITY_REPORT=<file>: the measured value of every parity comparison of the session, worst per
    (test, line), as JSON lines."""
    path = os.environ.get("FG_PARITY_REPORT")
    if not path:
        return
    import json

    import helpers

    worst = {}
    for test, where, kind, value in helpers.RECORDS:
        key = (test, where, kind)
        worst[key] = max(worst.get(key, float("-inf")), value)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        for (test, where, kind), value in sorted(worst.items()):
            f.write(json.dumps({"test": test, "at": where, "kind": kind, "value": value}) + "\n")
