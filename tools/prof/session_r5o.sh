#!/bin/bash
# Round 5: the radix sort's histograms and the scan's partial sums in buffers of their own (they shared scratch slots with live data of
# their callers).  The new regression test against a library with the OLD radix.hip (must fail), then everything on the new build.
TAG=${1:-r5o}; OUT=gpurun_out/$TAG; mkdir -p $OUT
echo "== old radix.hip, new test (expected: failed)"; ELP_HIP_SO=$PWD/elprep_amd/libelprep_hip_oldradix.so timeout 300 python -m pytest tests/test_gpu_round5.py -q -p no:cacheprovider -k small_read_set_behind 2>&1 | tail -2
echo "== new build"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
ELP_DEBUG_POISON=0x5A timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_poison.log 2>&1; echo "pytest (poisoned buffers) rc=$?"; tail -2 $OUT/pytest_poison.log
timeout 400 python tools/fuzz_parity.py 8000 40 > $OUT/fuzz_parity.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_parity.txt
