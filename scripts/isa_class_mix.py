"""Static instruction-class mix of the two dominant raster kernels, from the ISA of the built library.

The vector-issue roofline of bench.py weighs a kernel's vector instructions by what each class was measured to cost on
MI355X with >= 4 wavefronts per SIMD (profiles/r01_valu_issue_rates.md): ~2.3 clocks for full-rate fp32 / integer /
moves, ~4 for compares, selects, min / max, conversions, packed and DPP forms, ~8 for transcendentals (v_exp, v_rcp,
v_log, v_sqrt) and permlane swaps.  The hardware counter (SQ_INSTS_VALU) gives the TOTAL per launch, not the classes;
the shares come from here: every VALU instruction of the kernel's code object, counted once (a static mix -- the
per-slot body is most of the code, but this is not a dynamic count and says so in the bench line).
Usage: python scripts/isa_class_mix.py [out.json]   (compiles csrc/raster.hip device-only with the Makefile's flags)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "freegaussian_amd", "csrc", "raster.o")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = {"fg_raster_bwd": "raster_bwd_mixed_kernelILi3ELb1E", "fg_raster_fwd": "raster_fwd_mixed_kernelILi3E"}
COST = {"full": 2.3, "half": 4.0, "quarter": 8.0}
QUARTER = re.compile(r"^v_(exp|rcp|rsq|log|sqrt|sin|cos|permlane)")
HALF = re.compile(r"^v_(cmp|cmpx|cndmask|min|max|med3|cvt|pk_|mad_u64|mul_lo|mul_hi|readlane|readfirstlane|writelane|bfe|alignbit|perm_)")


def classify(mnemonic, operands):
    if QUARTER.match(mnemonic):
        return "quarter"
    if HALF.match(mnemonic) or "dpp" in mnemonic or "row_" in operands or "quad_perm" in operands:
        return "half"
    return "full"


def main():
    tmp = "/tmp/fg_raster_dev.s"  # the device code alone, as assembly, with the Makefile's flags for raster.o
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize",
                           "--cuda-device-only", "-S", os.path.join(ROOT, "freegaussian_amd", "csrc", "raster.hip"), "-o", tmp],
                          stderr=subprocess.DEVNULL)
    asm = open(tmp).read()
    out = {"source": "hipcc -S of csrc/raster.hip (device only, gfx950, the Makefile's flags), every VALU instruction of the kernel counted once",
           "cost_clocks": COST, "kernels": {}}
    cur = None
    counts = {}
    for line in asm.splitlines():
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = next((k for k, v in KERNELS.items() if v in m.group(1)), None)
            if cur:
                counts[cur] = {"full": 0, "half": 0, "quarter": 0, "salu": 0, "lds": 0, "vmem": 0}
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
        if not cur or not line.startswith("\t"):
            continue
        parts = line.split(";")[0].strip().split()
        if not parts:
            continue
        mn, ops = parts[0], " ".join(parts[1:])
        if mn.startswith("v_"):
            counts[cur][classify(mn, ops)] += 1
        elif mn.startswith("s_"):
            counts[cur]["salu"] += 1
        elif mn.startswith("ds_"):
            counts[cur]["lds"] += 1
        elif mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
            counts[cur]["vmem"] += 1
    for k, c in counts.items():
        v = c["full"] + c["half"] + c["quarter"]
        c["valu_total"] = v
        c["shares"] = {x: c[x] / v for x in ("full", "half", "quarter")}
        c["weighted_clocks_per_valu_instr"] = sum(c["shares"][x] * COST[x] for x in COST)
        out["kernels"][k] = c
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
