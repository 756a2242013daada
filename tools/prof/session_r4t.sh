# round-4 session t: host finalisation on the caller's thread for small tables; one full bench line
OUT=gpurun_out/r4t; mkdir -p $OUT
python tools/prof/host_finalize_speed.py > $OUT/host_speed.txt 2>&1; cat $OUT/host_speed.txt
timeout 600 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "full rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4t/bench_full.json"))
print(d["ms_per_step"], d["value"], d["stage_ms_per_step"], d.get("host_finalize_ms_per_step"), d.get("host_finalize_exposed_ms_per_step"), d["verify"]["ok"])
for k,e in d["extra"].items():
    if isinstance(e,dict) and "value" in e: print(k, e["value"], e.get("ms_per_step"), e.get("host_finalize_ms_per_step"), e.get("host_finalize_exposed_ms_per_step"))
PY
