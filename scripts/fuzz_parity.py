"""Randomised parity sweep, the long form of tests/test_gpu_parity.py::test_randomised_parity_*: HIP rasterization vs the
CPU oracle over random sizes, image shapes, SH degrees, render modes, raster modes, packed / unpacked, with hostile
Gaussians mixed in.  Same cases, same bar, same fp64 arbitration as the gate (tests/fuzz_cases.py).
Usage: python scripts/fuzz_parity.py [n_cases] [seed] [big]  -> one line per case + a summary; FUZZ_ONLY=k: case k only.
`big`: images of 640x360 ... 1920x1080 (900 ... 8160 tiles: the mixed launches with job lists, strips, list shares and
liveness) with up to 40000 Gaussians; the oracle composites with the C restatement there."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases  # noqa: E402

torch.set_num_threads(min(os.cpu_count() or 1, 16))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"
only = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None
bad = 0
for case in range(n_cases):
    if only is not None and case != only:
        continue
    ok, msg = fuzz_cases.check(fuzz_cases.Case(seed0, case, big))
    bad += 0 if ok else 1
    print(msg, flush=True)
print(f"fuzz: {n_cases - bad}/{n_cases} cases ok (seed {seed0}; gradients at {fuzz_cases.REL_TOL:g} or through the fp64 arbiter)")
sys.exit(1 if bad else 0)
