OUT=$PWD/gpurun_out/r3b; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout 600 python tools/prof/path_ab.py 50000000 build_ab/libH7.so build_ab/libH8.so build_ab/libH7.so build_ab/libH8.so > $OUT/ab.txt 2>&1; cut -c1-330 $OUT/ab.txt
