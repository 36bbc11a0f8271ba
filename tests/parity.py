"""Comparison rules shared by the parity tests.

Integer outputs (hit object id, hit mask, spawn mask, event counts) must be identical.
8-bit colour: the device adds the ray tree's contributions in another order than the
reference's recursion (and evaluates pow(x,3) as a correctly rounded cube), so channels
may differ where the double colour sits within ~1e-15 of a truncation boundary.
Tolerance (stricter than the <=1 LSB on >=99.9% of pixels the target states):
every channel within 1 LSB, and at least 99.9% of the pixels exactly equal.
"""
import numpy as np

MAX_LSB = 1
MIN_EXACT_FRACTION = 0.999


def assert_parity(dev: dict, ora: dict, what: str = ""):
    if ora.get("hit_id") is not None and dev.get("hit_id") is not None:
        bad = np.count_nonzero(dev["hit_id"] != ora["hit_id"])
        assert bad == 0, f"{what}: {bad} hit ids differ"
    for key in ("hit_mask", "spawn_mask"):
        if ora.get(key) is not None and dev.get(key) is not None:
            bad = np.count_nonzero(dev[key] != ora[key])
            assert bad == 0, f"{what}: {bad} {key} words differ"
    a = dev["bgr"].astype(np.int16)
    b = ora["bgr"].astype(np.int16)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.size:
        diff = np.abs(a - b)
        assert diff.max() <= MAX_LSB, f"{what}: max channel difference {diff.max()} LSB"
        exact = float((diff.max(axis=-1) == 0).mean())
        assert exact >= MIN_EXACT_FRACTION, f"{what}: only {exact:.6f} of the pixels are identical"
    if ora.get("counters") is not None and dev.get("counters") is not None:
        for f in ora["counters"].dtype.names:
            assert int(dev["counters"][f]) == int(ora["counters"][f]), (
                f"{what}: counter {f}: device {int(dev['counters'][f])} vs checker {int(ora['counters'][f])}"
            )
