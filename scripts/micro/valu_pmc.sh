#!/bin/bash
# Calibrates the SQ counters against kernels of known instruction mix (valu_rate pmc mode).
# Usage: gpurun -- 'bash scripts/micro/valu_pmc.sh'
out=gpurun_out/valu_pmc
mkdir -p $out
export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $out/a -o c -- ./scripts/micro/valu_rate pmc > $out/a.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_WAIT_ANY --output-format csv -d $out/b -o c -- ./scripts/micro/valu_rate pmc > $out/b.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o c -- ./scripts/micro/valu_rate pmc > $out/t.log 2>&1
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(dict)
for f in glob.glob("$out/*/*counter_collection.csv")+glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]]=float(r["Counter_Value"])
dur={}
for f in glob.glob("$out/t/*kernel_stats.csv")+glob.glob("$out/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0]]=float(r["AverageNs"])
for k in sorted(agg):
    c=agg[k]; n=c.get("SQ_INSTS_VALU",0)
    print(k, "dur_us=%.1f"%(dur.get(k,0)/1e3), " ".join(f"{a}={b:.4g}" for a,b in sorted(c.items())), "| ACTIVE_VALU/INSTS=%.3f"%(c.get("SQ_ACTIVE_INST_VALU",0)/max(n,1)))
PY
