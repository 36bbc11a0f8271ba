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
