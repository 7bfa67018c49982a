#!/bin/bash
# tools/pmc_run.sh -- rocprofv3 PMC passes over bench.py (run on the GPU box via gpurun).
# Counters are collected in separate passes, with --kernel-trace only (no hip/hsa tracing).
# usage: tools/pmc_run.sh <outdir-under-gpurun_out> [bench args...]
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --prewarm 0 --no-cpu-baseline > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
ls -R $OUT | head -40
