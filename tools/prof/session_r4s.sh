# round-4 session s: no second pair_win fill / code scan when nothing went through the mate table; the result line alone on stdout
OUT=gpurun_out/r4s; mkdir -p $OUT
bash tools/prof/ab_round.sh r4s 50000000 new
timeout 300 python bench.py --mode sfm --reads 8000000 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/sfm1.out 2> $OUT/sfm1.err; echo "sfm1 rc=$? lines on stdout: $(wc -l < $OUT/sfm1.out)"
ELP_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --reads 8000000 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/sfm2.out 2> $OUT/sfm2.err; echo "sfm2 rc=$? lines on stdout: $(wc -l < $OUT/sfm2.out)"; cut -c1-200 $OUT/sfm2.out
