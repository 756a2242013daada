#!/bin/bash
# GPU session: parity tests, then the bench path with count3 / apply3 and with round 2's k_bqsr_count / k_bqsr_apply_flat side by side on
# one box.  Usage: count3_round.sh <tag> [reads]
TAG=${1:-c3}; R=${2:-50000000}; RA=${3:-24000000}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
run() { # name reads env...
  local name=$1; local reads=$2; shift; shift
  env "$@" timeout 300 python bench.py --reads $reads --steps 4 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); k=d["kernel_ms_per_step"]
    print("$name", d["ms_per_step"], "count", k.get("bqsr_count"), "prologue_fast", k.get("bqsr_prologue_fast"), "apply", k.get("bqsr_apply"), d["stage_ms_per_step"])
except Exception as e: print("$name failed", e)
PY
}
run new $R A=1
run old $R ELP_COUNT_KERNEL=1 ELP_APPLY_KERNEL=1
