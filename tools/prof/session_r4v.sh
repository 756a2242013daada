# round-4 session v: all GPU tests, smoke(), the evidence set of the final build
OUT=gpurun_out/r4v; mkdir -p $OUT
timeout 700 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
bash tools/prof/final_round.sh r4v > $OUT.log 2>&1; grep "rc=" $OUT.log
