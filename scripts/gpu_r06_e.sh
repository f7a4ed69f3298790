#!/bin/bash
# round 6, visit E: the 1-rank RCCL test + the gloo lockstep tests with the sliced head all-reduce; the trained scene's tables
out=gpurun_out/r06_e
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=data/trained_scene_r06.npz
timeout 900 python -m pytest tests/test_rccl_one_rank.py -m gpu -x -q -s 2>&1 | tail -12 > $out/rccl_test.txt; cat $out/rccl_test.txt | cut -c1-1500
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "view_dp or exchange or world_size or dp" 2>&1 | tail -5 > $out/dp_tests.txt; cat $out/dp_tests.txt
FG_DP_FORCE_COLLECTIVES=1 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $out/bench_forced.json 2> $out/bench_forced.err
python3 -c "
import json; d=json.loads(open('$out/bench_forced.json').read().strip().splitlines()[-1]); print('forced 1-rank bench:', round(d['ms_per_step'],4), d.get('exchange'))"
timeout 300 python bench.py --layout trained:$T --steps 64 --warmup 10 --no-cpu-baseline --no-graph --stage-events all > $out/bench_trained_file.json 2> $out/bench_trained_file.err
python3 -c "
import json; d=json.loads(open('$out/bench_trained_file.json').read().strip().splitlines()[-1]); c=d['config']
print('trained(file):', round(d['ms_per_step'],4), d['stage_ms'], d['host_step_ms']); print({k:c[k] for k in ('N','V','I','I_raster','longest_tile_list','long_segment_calls','heavy_tile_steps','list_capacity_redos_in_timed_region')}); print(c['scene_statistics'])"
make -C freegaussian_amd/csrc stats > $out/make_stats.log 2>&1
for v in 0 1 2 3 4 5 6 7; do python scripts/raster_stats.py 0 $T $v > $out/raster_stats_trained_v$v.json 2>>$out/stats.err; done
python3 -c "
import json
for v in range(8):
    d=json.load(open('$out/raster_stats_trained_v%d.json'%v)); print(v, d.get('N'), d.get('V'), d['I'], d['I_raster'], 'bwd lanes/slot', round(d['bwd']['lanes_per_live_slot'],1), 'slots/entry', round(d['bwd']['live_slots_per_walked_entry'],2), 'fwd lanes/slot', round(d['fwd']['lanes_per_live_slot'],1))"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o stats -- python3 $R/bench.py --layout trained:$R/$T --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof.json 2> $R/$out/prof.err
find $R/$out/prof -name "*kernel_stats*" -exec cp {} $R/$out/kernel_stats_trained.csv \;
rm -rf $R/$out/prof
cd $R
python3 - <<PY
import csv
rows=list(csv.reader(open("$out/kernel_stats_trained.csv")))
for r in rows[1:16]: print(r[0].replace("(anonymous namespace)::","")[:44].ljust(46), r[1].rjust(6), round(float(r[3])/1e3,1))
PY
