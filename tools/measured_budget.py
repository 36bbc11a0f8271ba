"""Where the instructions of the headline frame go, measured on the shipped kernel.

GPU side (run under rocprofv3 --pmc ... and under rocprofv3 --kernel-trace): renders a ladder of
configurations of the 4K frame with the image-only kernel, N launches each, in a fixed order;
every step of the ladder adds one part of the work, so that differences of the per-launch
counters (and kernel times) are the cost of that part:

  background   every object behind the camera: primary rays, one nearest-hit pass, no hit
  depth2_dark  the default scene, max_depth 2 (children stop at the depth limit), no lights
  depth2       the same with the two lights: adds the shadow feelers of the primary hits
  depth3..5    adds the rays of tree levels 2, 3, 4 (with their feelers); depth5 is the headline

usage:  measured_budget.py            (run the ladder; prints the order as JSON)
        measured_budget.py report PMC_DIR TRACE_DIR   (combine the two profiler outputs)
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCENE = os.environ.get("RT_BUDGET_SCENE", "default")  # "glass": the same ladder on the glass scene, up to depth 8
if SCENE == "glass":
    LADDER = ["background", "depth2_dark", "depth2", "depth3", "depth5", "depth8"]
    N = 3
    KERNEL = "whitted_kernel<0, false, false, "   # depth <= 5: <.., false, true>, depth 8: <.., true, true>
else:
    LADDER = ["background", "depth2_dark", "depth2", "depth3", "depth4", "depth5"]
    N = 5
    KERNEL = "whitted_kernel<0, false, false, false, false, false>"
W, H = 3840, 2160


def run():
    import torch

    from raytracer_amd import abi, device, scenes

    if os.environ.get("RT_LIB_OVERRIDE"):  # another build of the library, for comparisons on one box
        device.LIB_PATH = os.path.abspath(os.environ["RT_LIB_OVERRIDE"])
    from raytracer_amd.device import DeviceTracer

    prims, lights = scenes.glass_scene() if SCENE == "glass" else scenes.default_scene()
    far = prims.copy()
    pp, qq = far["p"].copy(), far["q"].copy()
    pp[:, 2] -= 1000.0
    pp[0] = (0.0, 0.0, -1000.0)
    qq[0] = (0.0, 0.0, 1.0)
    far["p"], far["q"] = pp, qq
    cfg = {
        "background": (far, lights, 5),
        "depth2_dark": (prims, lights[:0], 2),
        "depth2": (prims, lights, 2),
        "depth3": (prims, lights, 3),
        "depth4": (prims, lights, 4),
        "depth5": (prims, lights, 5),
        "depth8": (prims, lights, 8),
    }
    t = DeviceTracer(0)
    out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream()
    rec = {}
    for name in LADDER:
        pr, lt, depth = cfg[name]
        t.upload_scene(pr, lt)
        p = abi.make_params(W, H, depth)
        for _ in range(N):
            t.render_device(p, st.cuda_stream, out.data_ptr())
        torch.cuda.synchronize()
        pc = p.copy()
        pc["flags"] = abi.RT_FLAG_PRUNE
        r = t.render(pc, want_ids=False, want_masks=False, want_counters=True)   # another kernel: not in the ladder
        rec[name] = {f: int(r["counters"][f]) for f in abi.COUNTER_FIELDS}
    print(json.dumps({"ladder": LADDER, "launches_each": N, "events": rec}))


def per_config(values):
    """values: per-dispatch numbers of the ladder kernel in launch order -> mean per configuration
    (the first launch of each configuration is left out: cold caches after the scene upload)."""
    assert len(values) == N * len(LADDER), f"{len(values)} dispatches of the ladder kernel, expected {N * len(LADDER)}"
    return {name: sum(values[i * N + 1:(i + 1) * N]) / (N - 1) for i, name in enumerate(LADDER)}


def report(pmc_dir, trace_dir, events_json=None):
    counters = collections.defaultdict(dict)
    f = glob.glob(f"{pmc_dir}/*/*_counter_collection.csv")[0]
    rows = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for c, v in rows.items():
        # one row per dispatch and counter instance: sum the instances of a dispatch
        per = collections.OrderedDict()
        for d, x in sorted(v):
            per[d] = per.get(d, 0.0) + x
        for name, m in per_config(list(per.values())).items():
            counters[name][c] = m
    f = glob.glob(f"{trace_dir}/*/*_kernel_trace.csv")[0]
    durs = []
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            durs.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    times = per_config([d for _, d in sorted(durs)])
    events = json.load(open(events_json))["events"] if events_json else {}
    waves = ((W + 7) // 8) * ((H + 7) // 8)
    print(f"Measured on the shipped image-only kernel, {W}x{H}, {waves} tiles of 64 pixels per launch (one wave each, or taken "
          f"from the queue by a fixed grid); per launch, mean of {N - 1} launches.")
    print(f"{'configuration':14s}{'kernel us':>11s}{'VALU insts':>14s}{'SALU insts':>14s}{'VALU/tile':>11s}{'rays':>12s}{'hits':>11s}{'feelers':>11s}")
    for name in LADDER:
        c, e = counters[name], events.get(name, {})
        print(f"{name:14s}{times[name]:11.1f}{c.get('SQ_INSTS_VALU', 0):14.0f}{c.get('SQ_INSTS_SALU', 0):14.0f}"
              f"{c.get('SQ_INSTS_VALU', 0) / waves:11.1f}{e.get('rays', 0):12d}{e.get('hits', 0):11d}{e.get('feelers', 0):11d}")
    print()
    print("Differences (what each part of the work costs in the headline frame):")
    total_v, total_t = counters[LADDER[-1]]["SQ_INSTS_VALU"], times[LADDER[-1]]
    ev = lambda n, k: events.get(n, {}).get(k, 0)
    steps = [
        ("primary rays + one nearest-hit pass + store, all pixels", None, "background", "rays"),
        ("primary hits: normal, own colour, child set-up", "background", "depth2_dark", "hits"),
        ("shadow feelers of the primary hits", "depth2_dark", "depth2", "feelers"),
    ] + [(f"tree levels up to {b[5:]} (rays, hits, feelers)", a, b, "rays") for a, b in zip(LADDER[2:-1], LADDER[3:])]
    print(f"{'part':58s}{'VALU insts':>13s}{'share':>8s}{'us':>8s}{'share':>8s}{'events':>11s}{'VALU/event/64':>15s}")
    for label, a, b, key in steps:
        dv = counters[b]["SQ_INSTS_VALU"] - (counters[a]["SQ_INSTS_VALU"] if a else 0.0)
        dt = times[b] - (times[a] if a else 0.0)
        de = ev(b, key) - (ev(a, key) if a else 0)
        per = dv * 64.0 / de if de else float("nan")
        print(f"{label:58s}{dv:13.0f}{dv / total_v:8.3f}{dt:8.1f}{dt / total_t:8.3f}{de:11d}{per:15.1f}")
    print("(the background configuration has the frame's objects out of sight: its primary pass rules every object out "
          "in the float filter, as the upper half of the headline frame does; 'VALU/event/64' = wave instructions x 64 "
          "lanes per event, i.e. lane-instructions per ray, hit or feeler if every lane were busy)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "report":
        report(*sys.argv[2:])
    else:
        run()
