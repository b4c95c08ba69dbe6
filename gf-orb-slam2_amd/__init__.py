"""MI355X-native ORB front-end for GF-ORB-SLAM2 (host-side mirror of the reference interface).

The product is `libgfo.so` (HIP kernels for gfx950 behind the C ABI of include/gfo.h); this
package is the thin host layer tests and bench.py drive it through.  It never imports the CPU
oracle and has no CPU fallback: without the built library and a gfx950 device it raises.
"""
from ._lib import GfoError, KEYPOINT_DTYPE, MAP_POINT_DTYPE, PROJ_QUERY_DTYPE, lib_path, load_library  # noqa: F401
from .extractor import ORBextractor  # noqa: F401
from .matcher import ORBmatcher, ORBVocabulary, StereoParams, FrameBounds  # noqa: F401
from .build import build_library, build_variants  # noqa: F401

# images per GPU per step of bench.py's headline workload AND of the parity test at that shape (tests/test_gpu_properties.py):
# one constant, so that what is timed is what is compared with the oracle
HEADLINE_BATCH = 256

__all__ = ["HEADLINE_BATCH", "ORBextractor", "ORBmatcher", "ORBVocabulary", "StereoParams", "FrameBounds", "GfoError", "KEYPOINT_DTYPE",
           "MAP_POINT_DTYPE", "build_library", "build_variants", "load_library", "lib_path"]
