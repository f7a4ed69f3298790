#!/bin/bash
# Round-3 side measurements: SH-degree-aware slab loads (PMC FETCH_SIZE of the per-Gaussian passes by degree), the
# model step eager vs graphed at the launch-bound sizes, the half-strip estimate, other image sizes, binning A/B.
# Usage: gpurun --timeout 1800 -- 'bash scripts/gpu_r03_extras.sh <tag>'
tag=${1:-r03_extras}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for d in 0 1 2 3; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_d${d}_$c -o c -- python3 bench.py --sh-degree $d --steps 3 --warmup 2 --no-cpu-baseline --no-graph > /dev/null 2> $out/pmc_d${d}_$c.err
  done
done
python3 - <<PY
import csv, glob, collections, json
out="$out"
res={}
for d in range(4):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        for f in glob.glob(f"{out}/pmc_d{d}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k=r["Kernel_Name"]
                if "preprocess" in k:
                    agg["fwd" if "fwd" in k else "bwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res[d]={k:{c:sum(v)/len(v) for c,v in cs.items()} for k,cs in agg.items()}
    for k,cs in res[d].items():
        cs["hbm_MB"]=(2*cs.get("FETCH_SIZE",0)+cs.get("WRITE_SIZE",0))*1024/1e6
json.dump(res, open(out+"/sh_degree_traffic.json","w"), indent=1)
for d in res: print("sh degree", d, {k:{c:round(v,1) for c,v in cs.items()} for k,cs in res[d].items()})
PY
find $out -name "*counter_collection.csv" -size +1M -delete
for sz in "100000 50 480 270" "300000 50 960 540" "1000000 30 1920 1080"; do timeout 300 python scripts/model_step_bench.py $sz 2>/dev/null | tr -d '\n ' > $out/model_step_$(echo $sz | tr ' ' '_').json; cut -c1-260 $out/model_step_$(echo $sz | tr ' ' '_').json; echo; done
timeout 600 python scripts/half_strip_estimate.py 2>/dev/null | tee $out/half_strip_estimate.txt | tail -4
bash scripts/gpu_sizes.sh 2>/dev/null | tee $out/sizes.txt
for b in depthfirst supertile depthfirst supertile; do
  FG_BINNING=$b timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-graph 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$b', round(d['value'],1), round(d['ms_per_step'],4), d['stage_ms'])" | tee -a $out/binning_ab.txt
done
