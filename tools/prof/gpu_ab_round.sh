export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r2o; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 600 python tools/prof/path_ab.py 24000000 build_ab/libD_r2n.so build_ab/libE_incr.so build_ab/libF_pingpong.so build_ab/libD_r2n.so > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
timeout 300 python tools/prof/path_ab.py 16000000 build_ab/libE_incr.so build_ab/libF_pingpong.so --quals full > $OUT/ab_full.txt 2>&1; cat $OUT/ab_full.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_prof.json 2> $OUT/prof.err; echo "prof rc=$?")
cat $OUT/bench_prof.json
DB=$(find $OUT/prof -name "*results.db" | head -1); echo db=$DB
python tools/prof/db_to_csv.py $DB $OUT/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra (50M reads)"; head -8 $OUT/kernel_stats.csv
python tools/prof/timeline.py $DB $OUT/timeline.csv; head -3 $OUT/timeline.csv
find $OUT/prof -size +20M -delete
