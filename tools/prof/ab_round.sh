#!/bin/bash
# GPU session: parity tests (unless SKIP_TESTS=1), then the bench path once per variant on one box.  A variant is NAME[:ENV=VAL[,ENV=VAL...]].
# Usage: ab_round.sh <tag> <reads> <variant>...     e.g.  ab_round.sh s12 50000000 new score_old:ELP_SCORE_KERNEL=1
TAG=$1; R=$2; shift; shift
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then timeout 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log; fi
for v in "$@"; do
  name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
  env $envs timeout 300 python bench.py --reads $R --steps 4 --warmup 1 --no-extra --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$name.json")); k=d["kernel_ms_per_step"]
    print("$name", d["ms_per_step"], {x: k[x] for x in list(k)[:12]}, d["stage_ms_per_step"], d.get("verify", {}).get("ok"))
except Exception as e: print("$name failed", e)
PY
done
