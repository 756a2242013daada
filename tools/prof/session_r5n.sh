#!/bin/bash
# Round 5: the prologue's covariate-split append aggregated per workgroup (one global atomic per covariate and tile instead of per wave):
# all GPU tests, the gather's kernel times with 4 / 16 / 32 read groups, the 16 M-read bench line with its side runs.
TAG=${1:-r5n}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 300 python tools/prof/rg_gather.py 16000000 4 16 32 2>&1 | grep "read groups" | tee $OUT/rg_gather.txt
bash tools/prof/session_r5j.sh $TAG | grep -v "host_parts\|^main\|full_quals\|shuffled" | cut -c1-900
timeout 300 python tools/fuzz_ragged.py 7000 30 > $OUT/fuzz_ragged.txt 2>&1; echo "fuzz ragged rc=$?"; tail -1 $OUT/fuzz_ragged.txt
timeout 300 python tools/fuzz_ragged.py 7100 30 one > $OUT/fuzz_one.txt 2>&1; echo "fuzz one rc=$?"; tail -1 $OUT/fuzz_one.txt
