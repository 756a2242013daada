OUT=$PWD/gpurun_out/r2t; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
timeout 600 python tools/prof/path_ab.py 50000000 build_ab/libH3.so build_ab/libH4.so build_ab/libH3.so build_ab/libH4.so > $OUT/ab.txt 2>&1; cat $OUT/ab.txt
