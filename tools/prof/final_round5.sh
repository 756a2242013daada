#!/bin/bash
# Final-state evidence of round 5 (one GPU box session): all GPU tests (also with poisoned device buffers) + smoke, the randomised parity
# sweeps (fresh contexts, and one context for many read sets), then tools/prof/final_round.sh (default bench line with extras and CPU
# baseline, rocprofv3 kernel trace of the bench command, timeline, sfm runs, PMC passes).
TAG=${1:-round5}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
ELP_DEBUG_POISON=0xA5 timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_poisoned_buffers.log 2>&1; echo "pytest (ELP_DEBUG_POISON=0xA5) rc=$?"; tail -1 $OUT/pytest_poisoned_buffers.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 500 python tools/fuzz_parity.py 5000 80 > $OUT/fuzz_parity_80_seeds.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_parity_80_seeds.txt
timeout 400 python tools/fuzz_ragged.py 5200 30 > $OUT/fuzz_ragged_30_seeds.txt 2>&1; echo "fuzz ragged rc=$?"; tail -1 $OUT/fuzz_ragged_30_seeds.txt
timeout 400 python tools/fuzz_ragged.py 5300 30 one > $OUT/fuzz_one_length_30_seeds.txt 2>&1; echo "fuzz one rc=$?"; tail -1 $OUT/fuzz_one_length_30_seeds.txt
timeout 400 python tools/fuzz_reuse.py 300 6 8 > $OUT/fuzz_reuse_6_sessions.txt 2>&1; echo "fuzz reuse rc=$?"; tail -1 $OUT/fuzz_reuse_6_sessions.txt
bash tools/prof/final_round.sh $TAG > $OUT.log 2>&1; grep "rc=" $OUT.log
