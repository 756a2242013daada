bash tools/prof/ab_round.sh r4m 50000000 new lsd:ELP_TUNE=tie_rounds=1
timeout 600 python bench.py > gpurun_out/r4m/bench_full.json 2> gpurun_out/r4m/bench_full.err; echo "full rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4m/bench_full.json"))
print(d["ms_per_step"], d["value"], d["stage_ms_per_step"], d.get("host_finalize_ms_per_step"), d.get("host_finalize_exposed_ms_per_step"))
for k,e in d["extra"].items():
    if isinstance(e,dict) and "value" in e: print(k, e["value"], e.get("ms_per_step"), e.get("host_finalize_ms_per_step"), e.get("host_finalize_exposed_ms_per_step"))
print(d["verify"]["ok"])
PY
