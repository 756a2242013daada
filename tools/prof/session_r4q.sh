# round-4 session q: the coordinate sort on words key << b | index (A/B against pairs), all GPU tests
OUT=gpurun_out/r4q; mkdir -p $OUT
bash tools/prof/ab_round.sh r4q 50000000 words pairs:ELP_TUNE=sort_pairs=1 words_b
