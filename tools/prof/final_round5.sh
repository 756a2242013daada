#!/bin/bash
# Final-state evidence of round 5 (one GPU box session): all GPU tests + smoke, the randomised parity sweeps, then tools/prof/final_round.sh
# (default bench line with extras and CPU baseline, rocprofv3 kernel trace of the bench command, timeline, sfm runs, PMC passes).
TAG=${1:-round5}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 500 python tools/fuzz_parity.py 5000 120 > $OUT/fuzz_parity_120_seeds.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_parity_120_seeds.txt
timeout 400 python tools/fuzz_ragged.py 5200 40 > $OUT/fuzz_ragged_40_seeds.txt 2>&1; echo "fuzz ragged rc=$?"; tail -1 $OUT/fuzz_ragged_40_seeds.txt
timeout 400 python tools/fuzz_ragged.py 5300 40 one > $OUT/fuzz_one_length_40_seeds.txt 2>&1; echo "fuzz one rc=$?"; tail -1 $OUT/fuzz_one_length_40_seeds.txt
bash tools/prof/final_round.sh $TAG > $OUT.log 2>&1; grep "rc=" $OUT.log
