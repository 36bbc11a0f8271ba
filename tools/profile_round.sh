#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG
# Writes under gpurun_out/TAG/: the bench line of the headline workload, the
# rocprofv3 --kernel-trace --stats summary of the same command, and one counter
# collection per PMC pass (separate runs, no trace domains mixed in), each reduced to a
# per-kernel summary by tools/pmc_reduce.py.
set -e
TAG=${1:-prof}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python3 bench.py > $OUT/bench_default_4k_d5.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
cp $OUT/stats/*/*_kernel_stats.csv $OUT/kernel_stats_default_4k_d5.csv
pass() {
    n=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc$n -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/pmc$n.log 2>&1
    python3 $ROOT/tools/pmc_reduce.py $OUT/pmc$n > $OUT/pmc${n}_summary.csv
}
pass 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY
pass 2 SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass 3 FETCH_SIZE
pass 4 WRITE_SIZE
rm -rf $OUT/stats $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
cat $OUT/bench_default_4k_d5.json
grep whitted $OUT/kernel_stats_default_4k_d5.csv
grep "whitted.*false" $OUT/pmc*_summary.csv
