"""GPU side of the per-phase instruction budget: render one frame of a workload with the
phase-counting build of the library (make -C raytracer_amd/csrc phases) and write how often a
wave enters each phase of the per-pixel walk, in the form tools/phase_budget.py --weights reads.

usage: phase_profile.py [WORKLOAD] [OUT.json] [MEASURED_SQ_INSTS_VALU]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from raytracer_amd import abi, device  # noqa: E402

device.LIB_PATH = os.path.join(ROOT, "raytracer_amd", "csrc", "_phase", "librt_hip_phase.so")
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "default_4k_d5"
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"phase_weights_{name}.json")
builder, w, h, d = bench.WORKLOADS[name]
prims, lights = builder()
t = device.DeviceTracer(0)
t.upload_scene(prims, lights)
params = abi.make_params(w, h, d)
params["flags"] = abi.RT_FLAG_PRUNE  # the walk of the image-only kernel, which the budget is about
r = t.render(params, want_ids=False, want_masks=False, want_counters=True)
c = [int(r["counters"][f]) for f in abi.COUNTER_FIELDS]
waves = ((w + 7) // 8) * ((h + 7) // 8)
n = len(prims)
slots_static = 4 + 3          # the loop body (a group of four) and the remainder (up to three)
slots_run = n                 # per pass over the scene (container-order loop)
it, near_ex, miss, hit, feel, feel_ex, child, lane_it = c[:8]
entries = {
    "prologue": waves, "epilogue": waves,
    "iter": it, "iter_tail": it, "pop": it,
    "near_filter": it * slots_run / slots_static,
    "near_exact": near_ex / slots_static,
    "miss": miss, "hit_setup": hit, "combine": hit,
    "feeler_dir": feel, "feeler_diffuse": feel,
    "feeler_loop": feel * slots_run / slots_static,
    "feeler_exact": feel_ex / slots_static,
    "children": child, "child_reflection": child, "child_refraction": child,
    "children_last": hit - child,
    "fallback": 0,
}
rec = {
    "workload": name,
    "waves": waves,
    "wave_entries_raw": dict(zip(["iterations", "near_exact_slots", "miss_branch", "hit_branch", "feelers",
                                  "feeler_exact_slots", "children_built", "lane_iterations"], c)),
    "lanes_walking_per_iteration": lane_it / (64.0 * it) if it else None,
    "note": "phases whose code exists once per object slot (near_filter, near_exact, feeler_loop, feeler_exact) are "
            "scaled by slots run / slots in the code; child_refraction is entered when any lane refracts and is "
            "counted like child_reflection (an upper bound)",
    "entries": entries,
    "measured_valu": float(sys.argv[3]) if len(sys.argv) > 3 else 0.0,
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec["wave_entries_raw"]))
