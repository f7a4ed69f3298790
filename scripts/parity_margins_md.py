"""profiles/rNN_parity_margins.md from the JSON lines `FG_PARITY_REPORT=<file> python -m pytest tests -m gpu` writes.
Usage: python scripts/parity_margins_md.py gpurun_out/r04_final/parity_margins.jsonl [r04] > profiles/r04_parity_margins.md"""
import json
import statistics
import sys

rows = [json.loads(l) for l in open(sys.argv[1])]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
out = [f"# {rnd}: parity margins of the GPU suite -- every comparison, measured value against the bar\n",
       "`FG_PARITY_REPORT=<file> python -m pytest tests -m gpu` (`tests/helpers.py` records every `rel_err`, `rel_l2` and\n"
       "`close_except_knife_edge` call; `tests/conftest.py` writes the worst value per (test, line)); this file:\n"
       "`scripts/parity_margins_md.py`.  MI355X, the round's final kernels.  Bar: 1e-4 scale-relative (BASELINE.json\n"
       "`north_star`); integers are `torch.equal`.  The oracle's backward is the reference's order (T rebuilt from\n"
       "`1 - alpha_out`), the same as the C compositor's and the kernels': no multiplier on any GPU assert\n"
       "(`grep -n \"\\* REL_TOL\" tests/` -> only the `0.2 * REL_TOL` lines of `test_oracle.py`, TIGHTER than the bar).  Looser\n"
       "bounds remain, each with its reason in place: the 200-step graphed-vs-eager training trajectories (chaotic: same\n"
       "counts within 3%, losses within 10%).\n"]
for kind, title in (("rel_l2", "relative L2 (gradients, images)"), ("rel_err", "max error / max value"),
                    ("knife_edge_pixels", "knife-edge pixels per image (count; allowed: 2e-5 of the pixels, at least 2)"),
                    ("knife_edge_rel_l2", "relative L2 of whole images, knife-edge pixels included")):
    rs = sorted([r for r in rows if r["kind"] == kind], key=lambda r: -r["value"])
    out.append(f"\n## {title}: {len(rs)} comparisons\n")
    out.append("| worst values | test | at |\n|---|---|---|")
    for r in rs[:12]:
        out.append(f"| {r['value']:.3g} | `{r['test'].split('::')[-1]}` | `{r['at']}` |")
    vals = [r["value"] for r in rs]
    if kind != "knife_edge_pixels" and vals:
        out.append(f"\nmedian {statistics.median(vals):.2e}; {sum(v < 1e-5 for v in vals)} of {len(vals)} below 1e-5; "
                   f"{sum(v >= 1e-4 for v in vals)} at or above 1e-4.")
print("\n".join(out))
