mkdir -p gpurun_out/bz1
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round5.py -m gpu -x -q -k "bgzf" > gpurun_out/bz1/tests.log 2>&1; tail -3 gpurun_out/bz1/tests.log
timeout 600 python tools/prof/bgzf_speed.py 4000000 1 2>&1 | tail -2 | cut -c1-300
