#!/bin/bash
# usage (GPU box, repo root): tools/pmc_quick.sh TAG WORKLOAD [WORKLOAD ...]
# Two counter passes (instruction mix / waits, then VALU lanes and memory instructions) of a
# short bench run of each workload (RT_LIB_OVERRIDE=lib.so: of another build of the library) (TRAFFIC=1: the two HBM traffic passes instead), reduced to per-kernel means under gpurun_out/TAG/.
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
    ARGS="--workload $W --no-cpu-baseline --no-extra --no-end-to-end --steps 2 --warmup 1"
    pass() {
        n=$1; shift
        timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc${n}_$W -- python3 $ROOT/tools/bench_with_lib.py $ARGS > $OUT/pmc${n}_$W.log 2>&1 || exit 1
        python3 $ROOT/tools/pmc_reduce.py $OUT/pmc${n}_$W > $OUT/pmc${n}_summary_$W.csv
        rm -rf $OUT/pmc${n}_$W
        grep "whitted_kernel<0, false" $OUT/pmc${n}_summary_$W.csv
    }
    pass 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY
    if [ -n "$TRAFFIC" ]; then pass 3 FETCH_SIZE; pass 4 WRITE_SIZE; continue; fi
    pass 2 SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM
done
