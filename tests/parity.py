"""Comparison rules shared by the parity tests.

Integer outputs (hit object id, hit mask, spawn mask, event counts) must be identical.
8-bit colour: every hit, miss and spawn decision is taken on the reference's doubles, but the
colour chain itself (shading terms, Fresnel weights, path weights, accumulation) runs in
single precision with fused multiply-adds on the device, so a channel may differ by one
where the reference's colour times 255 sits within ~1e-4 of an integer.
Tolerance (the target's "<= 1 LSB on >= 99.9% of pixels", and stricter): every channel within
1 LSB, and at least 99.9% of the pixels exactly equal.  The fractions seen are recorded in
parity.SEEN (printed by the test session's summary).
"""
import numpy as np

MAX_LSB = 1
MIN_EXACT_FRACTION = 0.999
SEEN = []  # (what, pixels, fraction of pixels byte-identical)


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
        SEEN.append((what, int(diff.shape[0] * diff.shape[1]) if diff.ndim == 3 else int(diff.size // 3), exact))
        assert exact >= MIN_EXACT_FRACTION, f"{what}: only {exact:.6f} of the pixels are identical"
    if ora.get("counters") is not None and dev.get("counters") is not None:
        for f in ora["counters"].dtype.names:
            assert int(dev["counters"][f]) == int(ora["counters"][f]), (
                f"{what}: counter {f}: device {int(dev['counters'][f])} vs checker {int(ora['counters'][f])}"
            )
