#!/bin/bash
# kernel trace + timeline of the default bench command without its side runs (one GPU box session).  Usage: quick_r6.sh <tag> [bench args]
TAG=${1:-quick}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra "$@" > $OUT/bench_prof.json 2> $OUT/prof.err; echo "prof rc=$?")
DB=$(find $OUT/prof -name "*results.db" | head -1)
python tools/prof/db_to_csv.py $DB $OUT/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra $*; the 8 timed steps (the 2 warm-up steps left out)" 2
python tools/prof/timeline.py $DB $OUT/timeline.csv; head -2 $OUT/timeline.csv
python tools/prof/timeline.py $DB $OUT/timeline_last_timed_step.csv 9; head -1 $OUT/timeline_last_timed_step.csv  # (2 warm-up + 8 timed steps: step 9)
find $OUT/prof -size +1M -delete
