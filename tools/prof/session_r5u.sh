#!/bin/bash
# Round 5, the committed build (stream waits by polling, the deferred table inserts, the sfm step's rows form): all GPU tests, smoke, the
# default bench line, the one-rank sfm line, two ranks over gloo.
TAG=${1:-r5u}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-260 $OUT/bench.json
timeout 400 python bench.py --mode sfm --no-cpu-baseline --no-extra > $OUT/bench_sfm1.json 2> $OUT/sfm1.err; echo "sfm1 rc=$?"; cut -c1-260 $OUT/bench_sfm1.json
ELP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --reads 8000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_sfm_gloo2.json 2> $OUT/sfm.err; echo "sfm2 rc=$?"; cut -c1-260 $OUT/bench_sfm_gloo2.json
