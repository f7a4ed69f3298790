#!/bin/bash
# HBM traffic of the MODEL path's per-Gaussian passes (fg_preprocess_raw_*: features_dc and features_rest are separate
# arrays) by SH degree.  Usage: gpurun -- 'bash scripts/gpu_model_sh_traffic.sh <tag>'
tag=${1:-model_sh}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for st in 0 1000 2000 3000; do
  for c in FETCH_SIZE WRITE_SIZE; do
    FG_MODEL_STEP=$st timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_${st}_$c -o c -- python3 scripts/model_step_bench.py 1000000 3 1920 1080 > /dev/null 2> $out/pmc_${st}_$c.err
  done
done
python3 - <<PY | tee $out/model_sh_traffic.txt
import csv, glob, collections
out="$out"
for st in (0, 1000, 2000, 3000):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        for f in glob.glob(f"{out}/pmc_{st}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"]
                if "preprocess" in k:
                    agg["fwd" if "fwd" in k else "bwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    line=f"sh degree {st//1000}:"
    for k in ("fwd","bwd"):
        cs={c:sum(v)/len(v) for c,v in agg[k].items()}
        line+=f"  {k} fetch {2*cs.get('FETCH_SIZE',0)*1024/1e6:.1f} MB write {cs.get('WRITE_SIZE',0)*1024/1e6:.1f} MB total {(2*cs.get('FETCH_SIZE',0)+cs.get('WRITE_SIZE',0))*1024/1e6:.1f} MB"
    print(line)
PY
find $out -name "*counter_collection.csv" -size +1M -delete
