# usage (GPU box, repo root): bash tools/ladder.sh LIB TAG -- the measured budget (tools/measured_budget.py) of
# raytracer_amd/LIB under rocprofv3 (kernel trace, then PMC), report in gpurun_out/budget_TAG/report.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/budget_$2; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export RT_LIB_OVERRIDE=$R/raytracer_amd/$1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/measured_budget.py > $O/events.json 2> $O/trace.log || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc -- python3 $R/tools/measured_budget.py > $O/pmc.out 2> $O/pmc.log || exit 1
cd $R && python3 tools/measured_budget.py report $O/pmc $O/trace $O/events.json > $O/report.txt; head -9 $O/report.txt
