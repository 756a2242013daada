#!/bin/bash
# Round 5, the committed build: rocprofv3 kernel trace of the bench command and the last step's timeline (final_round.sh's trace part).
TAG=${1:-r5w}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $OUT/bench_prof.json 2> $OUT/prof.err; echo "prof rc=$?")
DB=$(find $OUT/prof -name "*results.db" | head -1)
python tools/prof/db_to_csv.py $DB $OUT/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra (50M reads, the committed build of round 5); the 8 timed steps (the 2 warm-up steps left out: their first launches run cold)" 2
python tools/prof/db_to_csv.py $DB $OUT/kernel_stats_all_steps.csv "rocprofv3 top_kernels summary of the same trace: all 10 steps incl. warm-up"
python tools/prof/timeline.py $DB $OUT/timeline.csv; head -2 $OUT/timeline.csv
find $OUT/prof -size +20M -delete
cut -c1-200 $OUT/bench_prof.json; head -8 $OUT/kernel_stats.csv
