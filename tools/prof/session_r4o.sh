# round-4 session o: radix tiles of 8192 keys (A/B), the new tests, the host's table path on the box's cores
OUT=gpurun_out/r4o; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "radix or long_runs or pair_buckets" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python tools/prof/host_finalize_speed.py > $OUT/host_speed.txt 2>&1; cat $OUT/host_speed.txt
SKIP_TESTS=1 bash tools/prof/ab_round.sh r4o 50000000 t8k t4k:ELP_TUNE=radix_tile=1 t8k_b
