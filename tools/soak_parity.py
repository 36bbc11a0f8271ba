"""Soak test: random scenes (spheres of all sizes, planes at all angles, random materials with
zero coefficients, 0-6 lights), random frame sizes and depths, all four object-loop modes
(container order / sphere tree, each with and without the anchor grids), against
the CPU checker with the rules of tests/parity.py; the image-only render of every mode against
the same frame, with and without the pass that finishes pixels ending at the primary ray in
single precision.  Stops at the first difference and prints the seed.
usage: [RT_SOAK_MAX_OBJECTS=N] [RT_SOAK_FLOORS=1] soak_parity.py [SECONDS] [FIRST_SEED]
(RT_SOAK_FLOORS=1: few objects, most planes without reflection and transparency, depth <= 5 - the
scenes that pass applies to)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from raytracer_amd import abi, scenes  # noqa: E402
from raytracer_amd.device import DeviceTracer  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tests.parity import assert_parity  # noqa: E402


FLOORS = os.environ.get("RT_SOAK_FLOORS") == "1"


def random_scene(rng):
    objs = []
    for _ in range(int(rng.integers(1, int(os.environ.get("RT_SOAK_MAX_OBJECTS", "9" if FLOORS else "29")) + 1))):
        kind = rng.random() * (0.5 if FLOORS else 1.0) + (0.5 if FLOORS else 0.0)
        refl = 0.0 if rng.random() < 0.3 else float(rng.uniform(0, 1))
        tr = 0.0 if rng.random() < 0.3 else float(rng.uniform(0, 1))
        if FLOORS and kind >= 0.8 and rng.random() < 0.7:
            refl = tr = 0.0
        kd = tuple(rng.uniform(0.05, 1.0, 3))
        ri = float(rng.uniform(1.0, 1.8))
        if kind < 0.8:
            z = float(rng.uniform(-4.0, 40.0))
            spread = 0.5 + abs(z) * 0.7
            c = (float(rng.uniform(-spread, spread)), float(rng.uniform(-spread, spread)), z)
            r = float(rng.choice([0.0, 0.05, 0.3, 1.0, 3.0, 10.0])) * float(rng.uniform(0.5, 1.5))
            objs.append(scenes.sphere(c, r, kd, reflectivity=refl, transparency=tr, refraction_index=ri))
        else:
            n = tuple(rng.normal(0, 1, 3))
            o = tuple(rng.uniform(-8.0, 8.0, 3)) if rng.random() < 0.8 else (0.0, 0.0, 0.0)
            objs.append(scenes.plane(o, n, kd, reflectivity=refl, transparency=tr, refraction_index=ri))
    lights = [scenes.point_light(tuple(rng.uniform(0.2, 1.0, 3)), float(rng.uniform(0.1, 1.0)),
                                 tuple(rng.uniform(-15.0, 15.0, 3))) for _ in range(int(rng.integers(0, 7)))]
    return scenes._stack(objs, abi.PRIM_DTYPE), scenes._stack(lights, abi.LIGHT_DTYPE)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t = DeviceTracer(0)
    t0 = time.time()
    done = pixels = off_by_one = front_moved = shared_moved = 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        prims, lights = random_scene(rng)
        w, h, d = int(rng.integers(8, 160)), int(rng.integers(8, 120)), int(rng.integers(2, 6 if FLOORS else 9))
        t.upload_scene(prims, lights)
        for mode in (abi.RT_FLAG_LINEAR | abi.RT_FLAG_NOGRID, abi.RT_FLAG_TREE | abi.RT_FLAG_NOGRID,
                     abi.RT_FLAG_LINEAR | abi.RT_FLAG_GRID, abi.RT_FLAG_TREE | abi.RT_FLAG_GRID):
            for prune in (0, abi.RT_FLAG_PRUNE):
                p = abi.make_params(w, h, d)
                p["flags"] = mode | prune
                dev = t.render(p, want_counters=True)
                ora = oracle_lib.render(prims, lights, p, want_counters=True)
                what = f"seed {seed} {w}x{h} depth {d} flags {mode | prune}"
                try:
                    assert_parity(dev, ora, what)
                except AssertionError as e:
                    # the 99.9 % rule is a statement about frames, not about a few hundred
                    # pixels: on a small frame allow what it allows on 10,000 pixels
                    differing = int(np.count_nonzero(np.abs(dev["bgr"].astype(int) - ora["bgr"].astype(int)).max(axis=-1)))
                    if "of the pixels are identical" not in str(e) or differing > 10:
                        raise
                    off_by_one += differing
                    print(f"  below 99.9 % identical (small frame): {what}: {differing} pixel(s) off by one", flush=True)
                q = p.copy()
                q["flags"] = mode | prune | abi.RT_FLAG_NOFRONT
                img = t.render(q, want_ids=False, want_masks=False)
                if mode & abi.RT_FLAG_GRID:
                    # (queue launches with the grids let the lanes of a wave share rays: a pixel's
                    # colour is then summed in another order)
                    delta = np.abs(img["bgr"].astype(int) - dev["bgr"].astype(int)).max(axis=-1)
                    assert delta.max(initial=0) <= 1, f"{what}: walked image-only render off by {delta.max()}"
                    moved = int(np.count_nonzero(delta))
                    shared_moved += moved
                    assert moved <= max(10, w * h // 1000), f"{what}: walked image-only render differs in {moved} pixels"
                else:
                    assert np.array_equal(img["bgr"], dev["bgr"]), f"seed {seed}: image-only kernel differs"
                # as callers get it: pixels ending at the primary ray finished in single precision
                img = t.render(p, want_ids=False, want_masks=False)
                delta = np.abs(img["bgr"].astype(int) - dev["bgr"].astype(int)).max(axis=-1)
                assert delta.max(initial=0) <= 1, f"{what}: image-only render off by {delta.max()}"
                moved = int(np.count_nonzero(delta))
                front_moved += moved
                assert moved <= max(10, w * h // 1000), f"{what}: image-only render differs in {moved} pixels"
        done += 1
        pixels += w * h
        seed += 1
        if done % 20 == 0:
            print(f"{done} scenes, {pixels} pixels, {time.time() - t0:.0f} s, next seed {seed}", flush=True)
    print(f"soak ok: {done} scenes, {pixels} pixels x 8 renders (4 object-loop modes, full and pruned ray tree), seeds up to {seed - 1}; small frames below 99.9 % "
          f"identical pixels held {off_by_one} pixels off by one; "
          f"the image-only renders differ from the walked frames in {front_moved} pixels (by one); "
          f"walked image-only renders of the grid modes (rays shared between lanes) in {shared_moved} pixels (by one)")


if __name__ == "__main__":
    main()
