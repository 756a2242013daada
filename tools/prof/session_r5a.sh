# round-5 session a: all GPU tests (no -x: every failure is wanted), then the default bench line
OUT=gpurun_out/r5a; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5a/bench.json"))
    print("ms/step", d["ms_per_step"], "value", d["value"], "stages", d["stage_ms_per_step"])
    print("kern", d["kernel_ms_per_step"])
    for k, v in d.get("extra", {}).items():
        print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "stage_ms_per_step", "error", "host_finalize_exposed_ms_per_step")})
    print("verify", d.get("verify"))
except Exception as e:
    print("no bench json:", e)
PY
