timeout 900 python bench.py > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err; echo rc=$?
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_r04a.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "hip_event_mpix", d["hip_event_mpix_per_s"])
r=d["roofline"]; print({k:r[k] for k in ("bound","kernel","frac","frac_on_walked_lists","valu_frac","avg_ms") if k in r})
print(json.dumps(d["vector_issue_roofline"])[:900])
print(json.dumps(d.get("clustered_layouts"))[:1500])
print(d["stage_ms"])
print(d["cpu_baseline"]["value"] if "cpu_baseline" in d else None, d.get("graphed",{}).get("value"))
PY
tail -3 gpurun_out/bench_r04a.err
