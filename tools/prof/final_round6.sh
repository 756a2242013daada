#!/bin/bash
# Final-state evidence of round 6 (one GPU box session): all GPU tests - every case [fresh] and [reused], tests/conftest.py - also with poisoned
# device buffers, smoke, the randomised parity sweeps (fresh contexts, one context for many read sets, sfm contexts), then
# tools/prof/final_round.sh (default bench line with extras and CPU baseline, rocprofv3 kernel trace of the bench command, timeline, sfm runs,
# PMC passes -> traffic).
TAG=${1:-round6}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
ELP_DEBUG_POISON=0xA5 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_poisoned_buffers.log 2>&1; echo "pytest (ELP_DEBUG_POISON=0xA5) rc=$?"; tail -1 $OUT/pytest_poisoned_buffers.log
ELP_DEBUG_GUARD=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_guarded_buffers.log 2>&1; echo "pytest (ELP_DEBUG_GUARD=1) rc=$?"; tail -1 $OUT/pytest_guarded_buffers.log
ELP_TUNE=md_fused=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_separate_passes.log 2>&1; echo "pytest (md_fused=1) rc=$?"; tail -1 $OUT/pytest_separate_passes.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 500 python tools/fuzz_parity.py 6000 80 > $OUT/fuzz_parity_80_seeds.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_parity_80_seeds.txt
timeout 400 python tools/fuzz_ragged.py 6200 30 > $OUT/fuzz_ragged_30_seeds.txt 2>&1; echo "fuzz ragged rc=$?"; tail -1 $OUT/fuzz_ragged_30_seeds.txt
timeout 400 python tools/fuzz_ragged.py 6300 30 one > $OUT/fuzz_one_length_30_seeds.txt 2>&1; echo "fuzz one rc=$?"; tail -1 $OUT/fuzz_one_length_30_seeds.txt
timeout 400 python tools/fuzz_reuse.py 400 6 8 > $OUT/fuzz_reuse_6_sessions.txt 2>&1; echo "fuzz reuse rc=$?"; tail -1 $OUT/fuzz_reuse_6_sessions.txt
timeout 400 python tools/fuzz_reuse.py 500 4 6 sfm > $OUT/fuzz_reuse_sfm_4_sessions.txt 2>&1; echo "fuzz reuse sfm rc=$?"; tail -1 $OUT/fuzz_reuse_sfm_4_sessions.txt
bash tools/prof/final_round.sh $TAG > $OUT.log 2>&1; grep "rc=" $OUT.log
