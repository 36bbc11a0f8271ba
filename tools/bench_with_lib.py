"""bench.py on another build of the library: RT_LIB_OVERRIDE=path/to/lib.so bench_with_lib.py [bench args]
(measurement aid for same-box A/B runs, also under rocprofv3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import device  # noqa: E402

if os.environ.get("RT_LIB_OVERRIDE"):
    device.LIB_PATH = os.path.abspath(os.environ["RT_LIB_OVERRIDE"])
import bench  # noqa: E402

bench.main()
