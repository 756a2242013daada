OUT=gpurun_out/r5c; mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_round5.py -m gpu -q -p no:cacheprovider -k "merge_phase or many_read_groups_one_length or split_forms or different_sizes or distinct_lut" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest.log
