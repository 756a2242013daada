# round-5 session d: tests of the rows form, the merge across ranks, many read groups (apply record kernels changed); bench extras
OUT=gpurun_out/r5d; mkdir -p $OUT
timeout 500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round3.py tests/test_gpu_harness.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
timeout 500 python bench.py --reads 8000000 --c4-reads 0 --cpu-reads 2000000 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5d/bench.json"))
    print("main ms/step", d["ms_per_step"], "value", d["value"], "host", d["host_finalize_ms_per_step"], d["host_finalize_exposed_ms_per_step"], "verify", d.get("verify", {}).get("ok"))
    for k, v in d.get("extra", {}).items():
        if k != "pcie_inclusive":
            print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "stage_ms_per_step", "error", "host_finalize_ms_per_step", "host_finalize_exposed_ms_per_step")}, v.get("kernel_ms_per_step"))
except Exception as e:
    print("no json:", e)
PY
