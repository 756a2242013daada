#!/bin/bash
# rocprofv3 kernel trace of elp_stage_bgzf with ONE decoder launch behind the whole H2D copy (tuning bgzf_copy_chunk = 2^30): a launch's duration is
# then the decoder's own time.  (The default - chunks on the copy stream, launches alternating between two streams - overlaps the launches with
# each other and with the copies: their durations add up to more than the time the call takes.)   Usage: bgzf_trace_single.sh <tag>
TAG=${1:-bgzf_single}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
(cd /tmp && ELP_TUNE=bgzf_copy_chunk=1073741824 timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/tools/prof/bgzf_speed.py 4000000 1 > $OUT/prof.log 2>&1; echo "prof rc=$?")
DBZ=$(find $OUT/prof -name "*results.db" | head -1)
python tools/prof/db_to_csv.py $DBZ $OUT/kernel_stats_bgzf_single_launch.csv "ELP_TUNE=bgzf_copy_chunk=1073741824 rocprofv3 --kernel-trace --stats -- python tools/prof/bgzf_speed.py 4000000 1: elp_stage_bgzf x 3 (1.25 GB inflated, 19105 blocks each; ONE k_bgzf_tokens launch per call behind the whole copy), elp_emit_sorted_bgzf x 2; all dispatches"
head -8 $OUT/kernel_stats_bgzf_single_launch.csv
find $OUT/prof -name "*.db" -delete
