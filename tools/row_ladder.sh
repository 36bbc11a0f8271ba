#!/bin/bash
# Time of the lowest R image rows of the headline frame, for a ladder of R, with and without
# the primary-only pass: where in the frame the time goes.
# usage: tools/row_ladder.sh OUT.log
out=${1:-gpurun_out/row_ladder.log}
: > "$out"
for f in 0 128; do
  for r in 400 800 960 1088 1280 1600 2160; do
    timeout -k 10 100 python -u tools/time_one.py --flags $f --rows $r default_4k_d5 2>/dev/null >> "$out" || exit 1
  done
done
