#!/bin/bash
# round 6, visit B: the trained scene through the bench (stage table, kernel table, work counters), the in-process static-trained
# variant, the default bench line on the same box
out=gpurun_out/r06_b
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=data/trained_scene_r06.npz
timeout 300 python bench.py --layout trained:$T --steps 64 --warmup 10 --no-cpu-baseline --no-graph --stage-events all > $out/bench_trained_file.json 2> $out/bench_trained_file.err
python3 -c "
import json; d=json.loads(open('$out/bench_trained_file.json').read().strip().splitlines()[-1]); c=d['config']
print('trained(file):', round(d['ms_per_step'],4), d['stage_ms'], d['host_step_ms']); print({k:c[k] for k in ('N','V','I','I_raster','longest_tile_list','long_segment_calls','heavy_tile_steps','list_capacity_redos_in_timed_region')}); print(c['scene_statistics'])"
FG_BENCH_TRAIN_HERE=1 timeout 300 python bench.py --layout trained:auto --steps 64 --warmup 10 --no-cpu-baseline --no-graph > $out/bench_trained_here.json 2> $out/bench_trained_here.err
python3 -c "
import json; d=json.loads(open('$out/bench_trained_here.json').read().strip().splitlines()[-1]); c=d['config']
print('trained(here):', round(d['ms_per_step'],4), d['stage_ms'], d['host_step_ms']); print(c['scene']); print({k:c[k] for k in ('N','V','I','I_raster','longest_tile_list')}); print(c['scene_statistics'])"
make -C freegaussian_amd/csrc stats > $out/make_stats.log 2>&1
for v in 0 3 5; do python scripts/raster_stats.py 0 $T $v > $out/raster_stats_trained_v$v.json 2>>$out/stats.err; done
python scripts/raster_stats.py > $out/raster_stats_uniform.json 2>>$out/stats.err
python3 -c "
import json
for f in ('trained_v0','trained_v3','trained_v5','uniform'):
    d=json.load(open('$out/raster_stats_%s.json'%f)); print(f, d.get('N'), d.get('V'), d['I'], 'bwd lanes/slot', round(d['bwd']['lanes_per_live_slot'],1), 'slots/entry', round(d['bwd']['live_slots_per_walked_entry'],2), 'fwd lanes/slot', round(d['fwd']['lanes_per_live_slot'],1))"
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o stats -- python3 $R/bench.py --layout trained:$R/$T --steps 40 --warmup 10 --no-cpu-baseline --no-graph --no-clustered > $R/$out/prof.json 2> $R/$out/prof.err
find $R/$out/prof -name "*kernel_stats*" -exec cp {} $R/$out/kernel_stats_trained.csv \;
rm -rf $R/$out/prof
head -16 $R/$out/kernel_stats_trained.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-70,200-260
cd $R
timeout 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err
python3 -c "
import json; d=json.loads(open('$out/bench_default.json').read().strip().splitlines()[-1])
print('default:', round(d['value'],1), round(d['ms_per_step'],4), d['stage_ms'], d['host_step_ms'])
for k,v in d['clustered_layouts'].items(): print(k, v.get('ms_per_step'), v.get('ms_per_step_median'), v.get('stage_ms'), v.get('error'))"
