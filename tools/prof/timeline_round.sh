#!/bin/bash
# Kernel trace of the bench command and the timeline of its last step.  Usage: timeline_round.sh <tag> [reads]
TAG=${1:-tl}; R=${2:-50000000}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --reads $R --no-cpu-baseline --no-extra > $OUT/bench_prof.json 2> $OUT/prof.err; echo "prof rc=$?")
DB=$(find $OUT/prof -name "*results.db" | head -1)
python tools/prof/db_to_csv.py $DB $OUT/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra ($R reads); the 8 timed steps (the 2 warm-up steps left out: their first launches run cold)" 2
python tools/prof/timeline.py $DB $OUT/timeline.csv; head -2 $OUT/timeline.csv
find $OUT/prof -size +20M -delete
python -c "
import json; d=json.load(open('$OUT/bench_prof.json')); print(d['ms_per_step'], d['stage_ms_per_step'])"
