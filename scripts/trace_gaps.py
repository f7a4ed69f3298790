"""Idle time between the kernels of a step, from a rocprofv3 --kernel-trace CSV.

Usage: python scripts/trace_gaps.py <kernel_trace.csv> [skip_steps]
A step starts at every preprocess_fwd launch; per position in the step's launch sequence the table
gives the kernel, its mean duration and the mean idle gap before it (start - previous end)."""
import csv
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
    return n[-60:]


def main(path, skip=12):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    steps, cur = [], None
    for s, e, n in rows:
        if "preprocess_fwd_kernel" in n:
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append((s, e, n))
    steps = steps[skip:-1]
    lens = defaultdict(int)
    for st in steps:
        lens[len(st)] += 1
    L = max(lens, key=lens.get)
    steps = [st for st in steps if len(st) == L]
    print(f"{len(steps)} steps of {L} launches (launch-count histogram {dict(lens)})")
    tot_gap = tot_dur = 0.0
    wall = sum(steps[i + 1][0][0] - steps[i][0][0] for i in range(len(steps) - 1) if True) / max(1, len(steps) - 1)
    for k in range(L):
        dur = sum(st[k][1] - st[k][0] for st in steps) / len(steps)
        if k:
            gap = sum(st[k][0] - st[k - 1][1] for st in steps) / len(steps)
        else:
            gap = 0.0
        tot_gap += gap
        tot_dur += dur
        print(f"{k:3d} {short(steps[0][k][2]):60s} dur {dur / 1e3:8.2f} us   gap before {gap / 1e3:7.2f} us")
    print(f"sum of durations {tot_dur / 1e3:.1f} us, gaps inside a step {tot_gap / 1e3:.1f} us, "
          f"step-start to step-start (consecutive kept steps only approx) {wall / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
