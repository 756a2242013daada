#!/bin/bash
# Round 5: waiting for a stream by polling (ELP_SYNC_SPIN=1, the default) against hipStreamSynchronize (0), same box, back to back.
TAG=${1:-r5t}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do for spin in 0 1; do
  ELP_SYNC_SPIN=$spin timeout 300 python bench.py --no-extra --no-cpu-baseline > $OUT/bench_spin${spin}_$rep.json 2> $OUT/err_spin${spin}_$rep.txt; echo "spin $spin rep $rep rc=$?"
done; done
ELP_SYNC_SPIN=0 timeout 300 python bench.py --stages c2 --no-extra --no-cpu-baseline > $OUT/c2_spin0.json 2>/dev/null; ELP_SYNC_SPIN=1 timeout 300 python bench.py --stages c2 --no-extra --no-cpu-baseline > $OUT/c2_spin1.json 2>/dev/null
TAG=$TAG python - <<'PY'
import json, os, glob
for f in sorted(glob.glob('gpurun_out/%s/*.json' % os.environ['TAG'])):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'unreadable'); continue
    ks=sum(d['stage_ms_per_step'].values())
    print(os.path.basename(f), 'ms', d['ms_per_step'], 'kernels', round(ks,3), 'without a kernel', round(d['ms_per_step']-ks,3), 'host', d.get('host_finalize_ms_per_step'), d.get('host_finalize_exposed_ms_per_step'))
PY
