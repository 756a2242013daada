# round-4 session n: timeline of the current build + the extras with the page-locked LUT / table buffers
bash tools/prof/timeline_round.sh r4n 50000000
timeout 400 python bench.py --reads 16000000 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r4n/bench16.json 2> gpurun_out/r4n/bench16.err; echo "b16 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4n/bench16.json"))
print(d["ms_per_step"], d["value"], d["stage_ms_per_step"], d.get("host_finalize_ms_per_step"), d.get("host_finalize_exposed_ms_per_step"))
for k,e in d["extra"].items():
    if isinstance(e,dict) and "value" in e: print(k, e["value"], e.get("ms_per_step"), e.get("stage_ms_per_step"), e.get("host_finalize_ms_per_step"), e.get("host_finalize_exposed_ms_per_step"))
PY
