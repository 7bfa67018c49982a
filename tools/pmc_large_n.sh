#!/bin/bash
# tools/pmc_large_n.sh <outdir-under-gpurun_out> <log2n> [split]  -- rocprofv3 PMC passes (one counter
# group per run, --kernel-trace only) over tools/large_n_probe.py: address translation and L2 counters
# of the passes of a large transform.  Summarise with tools/pmc_summary.py.
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; LOG2N=$2; SPLIT=$3
mkdir -p $OUT
[ -n "$SPLIT" ] && export SVENTT_SPLIT=$SPLIT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/tools/large_n_probe.py $LOG2N > $OUT/$name.log 2>&1 || echo "pass $name failed (counter not available?)"
}
run grbm GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY
run utcl1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tccw TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
run tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
run lat TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
