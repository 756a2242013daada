#!/bin/bash
# One GPU-box session: parity tests, the bench line, and a rocprofv3 kernel-trace summary.  Usage: gpu_round.sh <tag> [bench_reads] [prof_reads]
TAG=${1:-run}; BR=${2:-50000000}; PR=${3:-16000000}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
nproc > $OUT/host.txt; free -g >> $OUT/host.txt; rocm-smi --showmeminfo vram >> $OUT/host.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 900 python bench.py --reads $BR > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; tail -3 $OUT/bench.err
if [ "$PR" != "0" ]; then
  # the kernel trace is taken over the bench command itself (same reads, steps and warm-up), without the CPU baseline leg
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --reads $PR --no-cpu-baseline > $OUT/prof.log 2>&1; echo "prof rc=$?")
  find $OUT/prof -name "*kernel_stats*" | head; find $OUT/prof -name "*kernel_trace*" -size +20M -delete
fi
