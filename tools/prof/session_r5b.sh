# round-5 session b: the tests of the code changed since session a, a reduced bench (extras: rg16 / rg32 after the wave-aggregated appends,
# the BGZF route with the device's DEFLATE), the sfm step through the C-ABI split phase with one rank and with two ranks on the one GPU
OUT=gpurun_out/r5b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_sfm.py tests/test_clean_sam_kat.py tests/test_gpu_round3.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest.log
timeout 500 python bench.py --reads 8000000 --c4-reads 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
timeout 300 python bench.py --mode sfm --reads 8000000 --no-extra --no-cpu-baseline --steps 4 > $OUT/bench_sfm1.json 2> $OUT/bench_sfm1.err; echo "sfm1 rc=$?"; tail -3 $OUT/bench_sfm1.err
ELP_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --reads 4000000 --steps 3 --warmup 1 > $OUT/bench_sfm2.json 2> $OUT/bench_sfm2.err; echo "sfm2 rc=$?"; tail -5 $OUT/bench_sfm2.err
python - <<'PY'
import json
for f in ("bench", "bench_sfm1", "bench_sfm2"):
    try:
        d = json.load(open("gpurun_out/r5b/%s.json" % f))
        print(f, "ms/step", d["ms_per_step"], "value", d["value"], "stages", d["stage_ms_per_step"], d.get("staging"), d.get("per_rank_reads"))
        for k, v in d.get("extra", {}).items():
            if k == "pcie_inclusive":
                print(k, v)
            else:
                print(k, {kk: v.get(kk) for kk in ("value", "ms_per_step", "stage_ms_per_step", "error", "host_finalize_exposed_ms_per_step")}, v.get("kernel_ms_per_step"))
    except Exception as e:
        print(f, "no json:", e)
PY
