#!/bin/bash
# Every new device buffer pre-filled with a byte (ELP_DEBUG_POISON): a kernel that reads memory nothing wrote fails a parity test
# instead of depending on what the allocator hands out.
TAG=${1:-poison}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for pz in 0xA5 0xFF; do
  ELP_DEBUG_POISON=$pz timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_$pz.log 2>&1; echo "poison $pz: pytest rc=$?"; tail -12 $OUT/pytest_$pz.log | grep -v "^$" | cut -c1-200
done
