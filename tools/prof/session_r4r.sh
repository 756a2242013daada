# round-4 session r: opt_eval with a few thousand workgroups, tie_scan with 32 tiles per workgroup
OUT=gpurun_out/r4r; mkdir -p $OUT
bash tools/prof/ab_round.sh r4r 50000000 new new_b
python - <<'PY'
import json
for n in ("new","new_b"):
    d=json.load(open("gpurun_out/r4r/bench_%s.json"%n)); k=d["kernel_ms_per_step"]
    print(n, {x:k[x] for x in k if x.startswith("mx_") or x.startswith("tie_")})
PY
