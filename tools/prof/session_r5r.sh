#!/bin/bash
# Round 5: the sfm step's table path in rows form (as the filter step's): tests, the one-rank sfm line, two ranks over gloo.
TAG=${1:-r5r}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_sfm.py tests/test_gpu_round5.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 400 python bench.py --mode sfm --no-cpu-baseline --no-extra > $OUT/bench_sfm1.json 2> $OUT/sfm1.err; echo "sfm1 rc=$?"; tail -2 $OUT/sfm1.err
ELP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --reads 8000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_sfm_gloo2.json 2> $OUT/sfm.err; echo "sfm2 rc=$?"; tail -2 $OUT/sfm.err
TAG=$TAG python - <<'PY'
import json, os
for f in ("bench_sfm1", "bench_sfm_gloo2"):
    d=json.loads(open('gpurun_out/%s/%s.json' % (os.environ['TAG'], f)).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d['value'], d['stage_ms_per_step'], d.get('allreduce_ms_per_step'))
    print(sorted(d['kernel_ms_per_step'].items(), key=lambda kv:-kv[1])[:14])
PY
