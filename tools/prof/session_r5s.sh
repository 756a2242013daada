#!/bin/bash
# Round 5, the last build: all GPU tests, smoke, the sweeps once more (the deferred table inserts and the sfm rows form came after the
# round5_final evidence run), the default bench line.
TAG=${1:-r5s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 400 python tools/fuzz_parity.py 9000 60 > $OUT/fuzz_parity_60_seeds.txt 2>&1; echo "fuzz rc=$?"; tail -1 $OUT/fuzz_parity_60_seeds.txt
timeout 300 python tools/fuzz_ragged.py 9200 20 > $OUT/fuzz_ragged_20_seeds.txt 2>&1; echo "fuzz ragged rc=$?"; tail -1 $OUT/fuzz_ragged_20_seeds.txt
timeout 300 python tools/fuzz_reuse.py 700 4 8 > $OUT/fuzz_reuse_4_sessions.txt 2>&1; echo "fuzz reuse rc=$?"; tail -1 $OUT/fuzz_reuse_4_sessions.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-300 $OUT/bench.json
