# round-4 session u: randomised parity sweep with random kernel choices; 150 M reads in one context
OUT=gpurun_out/r4u; mkdir -p $OUT
timeout 400 python tools/fuzz_parity.py 4000 80 > $OUT/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz.txt | cut -c1-250
timeout 400 python tools/fuzz_ragged.py 4000 30 > $OUT/fuzz_ragged.txt 2>&1; echo "fuzz_ragged rc=$?"; tail -1 $OUT/fuzz_ragged.txt | cut -c1-250
timeout 900 python bench.py --reads 150000000 --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $OUT/bench150.json 2> $OUT/bench150.err; echo "b150 rc=$?"; cut -c1-500 $OUT/bench150.json
