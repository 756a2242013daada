#!/bin/bash
# Round 5, confirmation of the final build: every GPU test, smoke, the default bench line.
OUT=gpurun_out/r5i; mkdir -p $OUT
timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json
