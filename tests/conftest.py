import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def tracer():
    """One device context for the whole GPU session (a single process, a single rt_ctx)."""
    from raytracer_amd.device import DeviceTracer

    t = DeviceTracer(0)
    yield t
    t.close()


def pytest_terminal_summary(terminalreporter):
    """How close the 8-bit frames came to the checker's, over every parity comparison made."""
    from tests import parity

    if not parity.SEEN:
        return
    px = sum(n for _, n, _ in parity.SEEN)
    same = sum(n * f for _, n, f in parity.SEEN)
    worst = min(parity.SEEN, key=lambda r: r[2])
    terminalreporter.write_line(
        f"parity: {len(parity.SEEN)} frame comparisons, {px} pixels, {same / px:.6f} byte-identical overall; "
        f"lowest {worst[2]:.6f} ({worst[0]}, {worst[1]} pixels)"
    )
