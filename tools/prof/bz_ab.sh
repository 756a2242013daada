mkdir -p gpurun_out/bz1
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py -m gpu -x -q -k "bgzf or bam" > gpurun_out/bz1/tests.log 2>&1; tail -3 gpurun_out/bz1/tests.log
for lvl in 1 6; do timeout 600 python tools/prof/bgzf_speed.py 4000000 $lvl check 2>&1 | head -4 | cut -c1-300; done
ELP_TUNE=bgzf_piece=201326592 timeout 600 python tools/prof/bgzf_speed.py 4000000 1 check 2>&1 | head -3 | cut -c1-200
