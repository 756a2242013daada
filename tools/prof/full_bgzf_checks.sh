mkdir -p gpurun_out/r6f
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r6f/tests.log 2>&1; tail -4 gpurun_out/r6f/tests.log
ELP_DEBUG_POISON=221 timeout 900 python -m pytest tests -m gpu -x -q -k "bgzf or bam or zlib or damaged" > gpurun_out/r6f/tests_poison.log 2>&1; tail -3 gpurun_out/r6f/tests_poison.log
ELP_DEBUG_GUARD=1 timeout 900 python -m pytest tests -m gpu -x -q -k "bgzf or bam or zlib or damaged" > gpurun_out/r6f/tests_guard.log 2>&1; tail -3 gpurun_out/r6f/tests_guard.log
