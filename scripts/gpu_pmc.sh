#!/bin/bash
# PMC passes (own runs, no tracing flags besides what --pmc needs) for the raster kernels.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_pmc.sh <tag>'
tag=${1:-pmc}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o c -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $out/$name.json 2> $out/$name.err
}
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
run sq2 SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum
run grbm GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections, os
out="$out"
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+"/*/*counter_collection.csv")+glob.glob(out+"/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:40]
        if "at::native" in k or "rocclr" in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
import json
traffic={}
for k,c in agg.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f=sum(c["FETCH_SIZE"])/len(c["FETCH_SIZE"]); w=sum(c["WRITE_SIZE"])/len(c["WRITE_SIZE"])
        # MI355X_MICROARCH.md "HBM": counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B read
        # requests at 64 B -> double it; WRITE_SIZE is exact for streaming stores and float atomics
        traffic[k.replace("void ","")]={"FETCH_SIZE_KiB":f,"WRITE_SIZE_KiB":w,"hbm_bytes_per_launch":(2*f+w)*1024}
        if "SQ_INSTS_VALU" in c:
            traffic[k.replace("void ","")]["valu_wave_instr_per_launch"]=sum(c["SQ_INSTS_VALU"])/len(c["SQ_INSTS_VALU"])
json.dump({"workload_key":"1000000x1920x1080xsh3","round":"$tag".split("_")[0],"source":"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of: python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph",
           "correction":"bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024","kernels":traffic}, open(out+"/pmc_traffic.json","w"), indent=1)
with open(out+"/pmc_summary.txt","w") as fo:
    for k in sorted(agg):
        line=k+": "+", ".join(f"{c}={sum(v)/len(v):.4g}" for c,v in sorted(agg[k].items()))
        print(line); fo.write(line+"\n")
PY
find $out -name "*counter_collection.csv" -size +2M -delete
