#!/bin/bash
# Round 5: the sr-tagged copies' table inserts of an sfm context listed by k_mate_pairs and made by a dense pass (k_mate_table).
TAG=${1:-r5q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_sfm.py tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 400 python bench.py --mode sfm --no-cpu-baseline --no-extra > $OUT/bench_sfm1.json 2> $OUT/sfm1.err; echo "sfm1 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5q/bench_sfm1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['stage_ms_per_step']); print({k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('md_')})
PY
