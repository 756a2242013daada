#!/bin/bash
# Final-state evidence of a round: default bench line (with extras and the CPU baseline), kernel trace of the bench command, PMC passes,
# the sfm bench path on two ranks sharing the one GPU (gloo), the strong-scaling mode.  Usage: final_round.sh <tag>
TAG=${1:-final}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -2 $OUT/bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extra > $OUT/bench_prof.json 2> $OUT/prof.err; echo "prof rc=$?")
DB=$(find $OUT/prof -name "*results.db" | head -1)
python tools/prof/db_to_csv.py $DB $OUT/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-extra (50M reads); the 8 timed steps (the 2 warm-up steps in front and the serial-order steps behind them left out)" 2 8
python tools/prof/db_to_csv.py $DB $OUT/kernel_stats_all_steps.csv "rocprofv3 top_kernels summary of the same trace: all 10 steps incl. warm-up"
python tools/prof/timeline.py $DB $OUT/timeline.csv; head -2 $OUT/timeline.csv
find $OUT/prof -size +20M -delete
timeout 400 python bench.py --mode sfm --no-cpu-baseline --no-extra > $OUT/bench_sfm1.json 2> $OUT/sfm1.err; echo "sfm1 rc=$?"; cat $OUT/bench_sfm1.json | cut -c1-400
ELP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --reads 8000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_sfm_gloo2.json 2> $OUT/sfm.err; echo "sfm rc=$?"; cat $OUT/bench_sfm_gloo2.json; tail -2 $OUT/sfm.err
ELP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 bench.py --gpus 2 --scaling strong --total-reads 12000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_sfm_strong2.json 2> $OUT/sfm2.err; echo "strong rc=$?"; cat $OUT/bench_sfm_strong2.json; tail -2 $OUT/sfm2.err
# the BGZF route: kernel trace of elp_stage_bgzf / elp_emit_sorted_bgzf on 4 M reads (zlib level 1), their wall times with the records checked
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_bgzf -o trace -- python $GRAFT_REPO_ROOT/tools/prof/bgzf_speed.py 4000000 1 > $OUT/bgzf_prof.log 2>&1; echo "bgzf prof rc=$?")
DBZ=$(find $OUT/prof_bgzf -name "*results.db" | head -1)
python tools/prof/db_to_csv.py $DBZ $OUT/kernel_stats_bgzf.csv "rocprofv3 --kernel-trace --stats -- python tools/prof/bgzf_speed.py 4000000 1: elp_stage_bgzf x 3 (1.25 GB inflated each), elp_emit_sorted_bgzf x 2; all dispatches"
find $OUT/prof_bgzf -size +20M -delete
for lvl in 1 6; do timeout 400 python tools/prof/bgzf_speed.py 4000000 $lvl check > $OUT/bgzf_speed_level$lvl.txt 2>&1; echo "bgzf speed $lvl rc=$?"; head -3 $OUT/bgzf_speed_level$lvl.txt | cut -c1-300; done
bash tools/prof/pmc_round.sh $TAG/pmc 8000000 > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log
python tools/prof/pmc_to_csv.py $OUT/pmc.csv $(find $OUT/pmc -name "*results.db") > $OUT/pmc_csv.log 2>&1; tail -2 $OUT/pmc_csv.log; wc -l $OUT/pmc.csv
find $OUT/pmc -name "*.db" -delete
