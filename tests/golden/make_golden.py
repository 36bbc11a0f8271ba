#!/usr/bin/env python3
"""Regenerates the golden vectors in this directory FROM THE CPU CHECKER (oracle/).

The reference holds no tests, images or fixtures and cannot be built in this image, so
these files are not reference outputs: they pin the checker (and through it the device
library) against accidental change.  The reference-derived pin is
survey_probe_counts.json.  Run from the repository root: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from raytracer_amd import abi, scenes  # noqa: E402
from tests import oracle_lib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def digest(r):
    h = hashlib.sha256()
    for k in ("bgr", "hit_id", "hit_mask", "spawn_mask"):
        h.update(np.ascontiguousarray(r[k]).tobytes())
    return h.hexdigest()


def main():
    prims, lights = scenes.default_scene()
    small = {}
    for (w, h) in ((64, 64), (160, 90)):
        for d in (2, 4, 5, 8):
            r = oracle_lib.render(prims, lights, abi.make_params(w, h, d))
            key = f"default_{w}x{h}_d{d}"
            for k in ("bgr", "hit_id", "hit_mask", "spawn_mask"):
                small[f"{key}/{k}"] = r[k]
    gp, gl = scenes.glass_scene()
    r = oracle_lib.render(gp, gl, abi.make_params(48, 27, 8))
    for k in ("bgr", "hit_id", "hit_mask", "spawn_mask"):
        small[f"glass_48x27_d8/{k}"] = r[k]
    sp, sl = scenes.random_spheres_scene()
    r = oracle_lib.render(sp, sl, abi.make_params(48, 27, 5))
    for k in ("bgr", "hit_id", "hit_mask", "spawn_mask"):
        small[f"spheres1024_48x27_d5/{k}"] = r[k]
    np.savez_compressed(os.path.join(HERE, "oracle_frames.npz"), **small)

    hashes = {}
    for (w, h, d) in ((640, 640, 4), (3840, 2160, 5)):
        r = oracle_lib.render(prims, lights, abi.make_params(w, h, d), want_counters=True)
        hashes[f"default_{w}x{h}_d{d}"] = {
            "sha256_bgr_hitid_hitmask_spawnmask": digest(r),
            "counters": {f: int(r["counters"][f]) for f in abi.COUNTER_FIELDS},
        }
    hashes["scene_bytes"] = {
        "default": hashlib.sha256(prims.tobytes() + lights.tobytes()).hexdigest(),
        "spheres1024": hashlib.sha256(sp.tobytes() + sl.tobytes()).hexdigest(),
        "glass": hashlib.sha256(gp.tobytes() + gl.tobytes()).hexdigest(),
    }
    with open(os.path.join(HERE, "oracle_hashes.json"), "w") as f:
        json.dump(hashes, f, indent=1, sort_keys=True)
    print("wrote", len(small), "arrays and", len(hashes), "hash records")


if __name__ == "__main__":
    main()
