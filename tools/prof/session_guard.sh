#!/bin/bash
# 4 KB of pattern behind every device buffer (ELP_DEBUG_GUARD=1), checked at release / regrowth / Engine.close: the whole -m gpu suite and
# the reuse sweep; a kernel that writes past the end of its buffer aborts.
TAG=${1:-guard}; OUT=gpurun_out/$TAG; mkdir -p $OUT
ELP_DEBUG_GUARD=1 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_guard.log 2>&1; echo "pytest (ELP_DEBUG_GUARD=1) rc=$?"; grep "ELP_DEBUG_GUARD" $OUT/pytest_guard.log | head -5; tail -2 $OUT/pytest_guard.log
ELP_DEBUG_GUARD=1 timeout 400 python tools/fuzz_reuse.py 500 4 8 > $OUT/fuzz_reuse_guard.txt 2>&1; echo "fuzz reuse (guard) rc=$?"; tail -1 $OUT/fuzz_reuse_guard.txt
