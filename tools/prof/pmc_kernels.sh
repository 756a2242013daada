#!/bin/bash
# PMC passes (instruction mix, waits, LDS conflicts) of the path on 8 M reads, printed for the kernels named.  Usage: pmc_kernels.sh <tag> <grep pattern>
TAG=${1:-pmc}; PAT=${2:-bqsr_}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
run() { local name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o pmc -- python $GRAFT_REPO_ROOT/tools/prof/run_path.py 8000000 1 > $OUT/$name.log 2>&1; echo "$name rc=$?") }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
if [ -n "$MEM" ]; then  # memory-side passes as well (MEM=1)
  run fetch FETCH_SIZE
  run write WRITE_SIZE
  run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
  run tcp TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
fi
python tools/prof/pmc_to_csv.py $OUT/pmc.csv $(find $OUT -name "*results.db") > /dev/null 2>&1
grep -E "^kernel|$PAT" $OUT/pmc.csv | cut -c1-420
find $OUT -name "*.db" -delete
