#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG [WORKLOAD ...]
# Writes under gpurun_out/TAG/, per workload W (default: the headline workload and the two
# others the bench line reports): the rocprofv3 --kernel-trace --stats summary of the bench
# command, and one counter collection per PMC pass (separate runs, no trace domains mixed
# in), each reduced to a per-kernel summary by tools/pmc_reduce.py; then
# tools/pmc_to_traffic.py folds the passes into gpurun_out/TAG/traffic.json.  For the
# headline workload the full bench line is recorded too.
# Copy what is to be kept to profiles/ (named per round) afterwards.
set -e
TAG=${1:-prof}; shift || true
WORKLOADS=${@:-default_4k_d5 spheres1024_4k_d5 glass_4k_d8}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 bench.py > $OUT/bench_default_4k_d5.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
for W in $WORKLOADS; do
    if [ $W = default_4k_d5 ]; then STEPS="--steps 200 --warmup 20"; PSTEPS="--steps 10 --warmup 2"
    else STEPS="--steps 4 --warmup 1"; PSTEPS="--steps 2 --warmup 1"; fi
    ARGS="--workload $W --no-cpu-baseline --no-extra --no-end-to-end"
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python3 $ROOT/bench.py $ARGS $STEPS > $OUT/stats_$W.log 2>&1
    cp $OUT/stats_$W/*/*_kernel_stats.csv $OUT/kernel_stats_$W.csv
    rm -rf $OUT/stats_$W
    pass() {
        n=$1; shift
        timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc${n}_$W -- python3 $ROOT/bench.py $ARGS $PSTEPS > $OUT/pmc${n}_$W.log 2>&1
        python3 $ROOT/tools/pmc_reduce.py $OUT/pmc${n}_$W > $OUT/pmc${n}_summary_$W.csv
        rm -rf $OUT/pmc${n}_$W
    }
    pass 1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY
    pass 2 SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
    pass 3 FETCH_SIZE
    pass 4 WRITE_SIZE
    echo "== $W"
    grep "whitted_kernel<0, false" $OUT/kernel_stats_$W.csv || true
done
cd $ROOT
python3 tools/pmc_to_traffic.py $OUT $TAG $WORKLOADS > $OUT/traffic.json
cat $OUT/traffic.json
cut -c1-1500 $OUT/bench_default_4k_d5.json
