#!/bin/bash
# usage (GPU box, repo root): tools/grid_ladder.sh [LIB ...]  -> where the frames of the two grid
# workloads spend their time (depth / lights ladder), grid resolution sweep, and the same for
# other builds of the library (paths relative to the repo root)
cd $GRAFT_REPO_ROOT
T="timeout -k 10 120 python -u tools/time_one.py"
for W in spheres1024_4k_d5 glass_4k_d8; do
    $T $W && $T --flags 16 $W && $T --depth 2 $W && $T --depth 2 --no-lights $W && $T --no-lights $W || exit 1
    for R in 16 32 64 256; do RT_GRID_RES=$R $T $W || exit 1; done
done
for L in "$@"; do
    RT_LIB_OVERRIDE=$L $T spheres1024_4k_d5 glass_4k_d8 || exit 1
done
