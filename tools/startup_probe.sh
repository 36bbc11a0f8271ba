#!/bin/bash
# usage (GPU box, repo root): tools/startup_probe.sh -> what the first Tracer() call of a process
# is made of (rt_main's own breakdown), three runs each, for the shipped library and for a build
# whose built-in workspace array is 4 KB instead of 52.5 MiB
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "== shipped, run $i"; timeout -k 5 60 raytracer_amd/host/rt_main --width 3840 --height 2160 --depth 5 --frames 3 --out /tmp/a.bmp || exit 1
done
if [ -f raytracer_amd/csrc/_ab/noscratch/librt_hip.so ]; then
for i in 1 2 3; do
  echo "== small built-in array, run $i"; LD_LIBRARY_PATH=raytracer_amd/csrc/_ab/noscratch timeout -k 5 60 raytracer_amd/host/rt_main --width 3840 --height 2160 --depth 5 --frames 3 --out /tmp/a.bmp || exit 1
done
fi
