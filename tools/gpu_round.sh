#!/bin/bash
# usage (GPU box, repo root): tools/gpu_round.sh TAG  -> GPU tests, then the round's profiles
set -e
TAG=${1:-r}
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/${TAG}_pytest.log 2>&1 || { tail -40 gpurun_out/${TAG}_pytest.log; exit 1; }
tail -3 gpurun_out/${TAG}_pytest.log
bash tools/profile_round.sh $TAG
