#!/bin/bash
# Round 5: where the host's table path spends its time in the side runs (3 steps after 1 warm-up report 2 ms for tables the main line
# finalises in 0.45 ms): per-step parts with 8 steps after 2 warm-ups.
TAG=${1:-r5j}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ELP_BENCH_HOST_PARTS=1 ELP_BENCH_SIDE_STEPS=8 ELP_BENCH_SIDE_WARMUP=2 timeout 600 python bench.py --reads 16000000 --steps 6 --warmup 2 --no-cpu-baseline --c4-reads 0 > $OUT/bench.json 2> $OUT/bench.err; echo "rc=$?"
TAG=$TAG python - <<'PY'
import json
d=json.loads(open('gpurun_out/'+__import__('os').environ.get('TAG','r5j')+'/bench.json').read().strip().splitlines()[-1])
print('main', d['ms_per_step'], d.get('host_parts_ms'))
for k,v in d['extra'].items():
    if 'host_parts_ms' in v: print(k, v['value'], v['ms_per_step'], v['host_finalize_ms_per_step'], v['host_finalize_exposed_ms_per_step'], v['host_parts_ms'])
PY
TAG=$TAG python - <<'PY'
import json, os
d=json.loads(open('gpurun_out/'+os.environ.get('TAG','r5j')+'/bench.json').read().strip().splitlines()[-1])
for k in ('rg16','rg32'):
    e=d['extra'][k]; print(k, e['value'], e['stage_ms_per_step']); print(sorted(e['kernel_ms_per_step'].items(), key=lambda kv:-kv[1])[:16])
PY
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ragged.py -q -p no:cacheprovider 2>&1 | tail -3
