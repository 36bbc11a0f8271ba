#!/bin/bash
# usage: tools/gpu_eval.sh  -> timing of the bench workload + instruction counters of its kernels
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u tools/time_one.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmcq
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmcq.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/pmc_summary.py gpurun_out/pmcq
