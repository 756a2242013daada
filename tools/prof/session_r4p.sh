# round-4 session p: radix tiles of 16384 keys (A/B against 8192), the extras with the LUT uploaded from the caller's page-locked buffer
OUT=gpurun_out/r4p; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
SKIP_TESTS=1 bash tools/prof/ab_round.sh r4p 50000000 t8k t16k:ELP_TUNE=radix_tile=3
timeout 400 python bench.py --reads 16000000 --steps 4 --warmup 1 --no-cpu-baseline > $OUT/bench16.json 2> $OUT/bench16.err; echo "b16 rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4p/bench16.json"))
print(d["ms_per_step"], d["value"], d["stage_ms_per_step"], d.get("host_finalize_ms_per_step"), d.get("host_finalize_exposed_ms_per_step"))
for k,e in d["extra"].items():
    if isinstance(e,dict) and "value" in e: print(k, e["value"], e.get("ms_per_step"), e.get("host_finalize_ms_per_step"), e.get("host_finalize_exposed_ms_per_step"))
PY
