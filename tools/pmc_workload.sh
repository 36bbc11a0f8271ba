#!/bin/bash
# usage: tools/pmc_workload.sh WORKLOAD [STEPS]  -> instruction counters of the workload's kernel
set -e
W=${1:-glass_4k_d8}; K=${2:-2}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmcw
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcw -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps $K --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmcw.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py gpurun_out/pmcw
grep "^{" gpurun_out/pmcw.log | cut -c1-200
