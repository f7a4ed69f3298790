#!/bin/bash
# round 5, visit J: the cheaper footprint-mask rows (one cross-section per tile boundary) -- parity, instruction count, times
out=gpurun_out/r05_j
mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -s -k "footprint or randomised_parity or cfg4_whole" 2>&1 | grep -E "isotropic:|needles:|passed|failed|FAILS|MISMATCH|Error|assert" | cut -c1-300 | tail -20
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $out/sq -o c -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-graph > $out/sq.json 2> $out/sq.err
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/sq/*counter_collection.csv")+glob.glob("$out/sq/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if any(s in k for s in ("preprocess_fwd","sb_count","sb_scatter","raster_fwd_mixed","raster_bwd_mixed")):
        print(k, {c: "%.4g" % (sum(v)/len(v)) for c,v in agg[k].items()})
PY
rm -rf $out/sq
for lay in uniform needles:0.3:10; do
  f=$out/b_${lay//[:.+]/_}
  timeout 200 python bench.py --layout $lay --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $f.json 2> $f.err
  python3 -c "
import json,sys; d=json.loads(open('$f.json').read().strip().splitlines()[-1]); s=d['stage_ms']; print('$lay', 'median', round(d['host_step_ms']['median'],4), 'mean', round(d['ms_per_step'],4), s, 'I_raster', d['config']['I_raster'])" || tail -2 $f.err
done
