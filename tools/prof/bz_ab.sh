mkdir -p gpurun_out/bz1
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -x -q -k "bgzf or zlib or damaged" > gpurun_out/bz1/tests.log 2>&1; tail -3 gpurun_out/bz1/tests.log
ELP_TUNE=bgzf_copy_chunk=1000000 timeout 600 python tools/prof/bgzf_speed.py 4000000 1 check 2>&1 | head -3 | cut -c1-200
ELP_TUNE=bgzf_copy_chunk=1000000 timeout 600 python tools/prof/bgzf_speed.py 4000000 6 2>&1 | head -2 | tail -1 | cut -c1-120
timeout 600 python tools/prof/bgzf_speed.py 4000000 1 2>&1 | head -1 | cut -c1-200
