#!/bin/bash
# The committed build once more at the end of a round (after the evidence session): all GPU tests, smoke, the default bench line, the BGZF
# route's kernel trace (one decoder launch per call) and its checked wall times.  Usage: final_confirm.sh <tag>
TAG=${1:-confirm}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -1 $OUT/bench.err
bash tools/prof/bgzf_trace_single.sh $TAG/bgzf_single
for lvl in 1 6; do timeout 400 python tools/prof/bgzf_speed.py 4000000 $lvl check > $OUT/bgzf_speed_level$lvl.txt 2>&1; echo "bgzf speed $lvl rc=$?"; head -3 $OUT/bgzf_speed_level$lvl.txt | cut -c1-300; tail -2 $OUT/bgzf_speed_level$lvl.txt | cut -c1-200; done
