#!/bin/bash
# PMC passes (instruction mix, waits) of elp_stage_bgzf's kernels.  Usage: pmc_bgzf.sh <tag> [reads]
TAG=${1:-pmc_bgzf}; PR=${2:-2000000}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
run() { local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc -- python $GRAFT_REPO_ROOT/tools/prof/bgzf_speed.py $PR 1 > $OUT/$name.log 2>&1; echo "$name rc=$?") }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
run sq3 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_IFETCH
python tools/prof/pmc_to_csv.py $OUT/pmc.csv $(find $OUT -name "*results.db") > /dev/null 2>&1
grep -E "^kernel|bgzf" $OUT/pmc.csv | cut -c1-700
find $OUT -name "*.db" -delete
