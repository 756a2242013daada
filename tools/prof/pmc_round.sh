#!/bin/bash
# PMC counter passes (separate from the kernel-trace timing run, as the guide prescribes).  Usage: pmc_round.sh <tag> [reads]
TAG=${1:-pmc}; PR=${2:-8000000}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
run() { # name counters...
  local name=$1; shift
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc -- python $GRAFT_REPO_ROOT/tools/prof/run_path.py $PR 1 > $OUT/$name.log 2>&1; echo "$name rc=$?")
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
ls -la $OUT/*/ | head -30
