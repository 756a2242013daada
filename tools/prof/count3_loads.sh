#!/bin/bash
# which of count3's load streams sets its floor: the kernel stopped behind the eligible-base mask (ELP_C3_DEBUG 8) with one stream
# after the other left out (16 reference, 32 known-site bits, 64 SEQ, 128 QUAL).  Timing only.  Usage: count3_loads.sh <tag> [reads]
TAG=${1:-c3l}; R=${2:-24000000}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
for d in 0 8 24 40 72 136 248 9; do
  ELP_C3_DEBUG=$d timeout 200 python bench.py --reads $R --steps 4 --warmup 1 --no-extra --no-cpu-baseline > $OUT/b_$d.json 2> $OUT/b_$d.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$d.json")); k=d["kernel_ms_per_step"]; print("dbg $d: count", k.get("bqsr_count"))
except Exception as e: print("dbg $d failed", e)
PY
done
