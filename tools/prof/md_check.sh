mkdir -p gpurun_out/r6h
timeout 1500 python -m pytest tests -m gpu -x -q -k "markdup or mark_dup or duplicates or shuffled or sfm or mate or fuzz or round2 or round3" > gpurun_out/r6h/tests.log 2>&1; tail -3 gpurun_out/r6h/tests.log
timeout 900 python tools/fuzz_parity.py 400 40 > gpurun_out/r6h/fuzz.log 2>&1; tail -1 gpurun_out/r6h/fuzz.log
timeout 900 python tools/fuzz_reuse.py 2 3 > gpurun_out/r6h/reuse.log 2>&1; tail -1 gpurun_out/r6h/reuse.log
timeout 600 python tools/prof/shuffled_serial.py 16000000 4 2>&1 | head -6
