#!/bin/bash
# Round 5: (1) the apply record passes on 1024 persistent workgroups (their counters' one cache line took an atomic per workgroup and
# covariate); (2) timing probes of the prologue's covariate-split append: without the global atomic, and without the groups too.
TAG=${1:-r5m}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ragged.py -q -p no:cacheprovider 2>&1 | tail -3
timeout 300 python tools/prof/rg_gather.py 16000000 4 16 32 2>&1 | grep "read groups" | tee $OUT/rg_gather.txt
for v in 1 2; do echo "probe $v"; ELP_HIP_SO=$PWD/elprep_amd/libelprep_hip_pfprobe$v.so timeout 300 python tools/prof/rg_gather.py 16000000 16 32 2>&1 | grep "read groups\|rror" | tee -a $OUT/rg_gather_probe$v.txt; done
bash tools/prof/session_r5j.sh $TAG | grep -v "host_parts\|^main\|full_quals\|shuffled" | cut -c1-900
