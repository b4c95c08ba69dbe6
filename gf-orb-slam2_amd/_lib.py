"""ctypes binding of include/gfo.h."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
MAP_POINT_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"),
                            ("view_cos", "<f4"), ("level", "<i4"), ("flags", "<i4")])
PROJ_QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("min_level", "<i4"),
                             ("max_level", "<i4"), ("angle", "<f4"), ("flags", "<i4")])
GFO_MAX_LEVELS = 16
GFO_STAGE_MAX = 16


class GfoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gfo error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_batch", C.c_int32)]


class StereoParamsC(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("mbf", C.c_float), ("mb", C.c_float), ("min_x", C.c_float)]


class FrameBoundsC(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


class VocabularyC(C.Structure):
    _fields_ = [("first_child", C.c_void_p), ("n_children", C.c_void_p), ("descriptors", C.c_void_p), ("word_id", C.c_void_p),
                ("weight", C.c_void_p), ("n_nodes", C.c_int32), ("depth", C.c_int32), ("weight64", C.c_void_p)]


class BowModeC(C.Structure):
    _fields_ = [("weighting", C.c_int32), ("norm", C.c_int32)]


class ProjectionBatchC(C.Structure):
    _fields_ = [("mps", C.c_void_p), ("kp_taken", C.c_void_p), ("on_device", C.c_int32), ("stereo", C.c_int32),
                ("th", C.c_float), ("nn_ratio", C.c_float), ("bounds", FrameBoundsC)]


class ProjModeC(C.Structure):
    _fields_ = [("use_ratio", C.c_int32), ("nn_ratio", C.c_float), ("th_dist", C.c_int32), ("check_orientation", C.c_int32), ("max_matches", C.c_int32)]


class FeatureVectorC(C.Structure):
    _fields_ = [("node_ids", C.c_void_p), ("node_start", C.c_void_p), ("items", C.c_void_p), ("n_nodes", C.c_int32)]


class DeliveryC(C.Structure):
    _fields_ = [("nimg", C.c_int32), ("kp_stride", C.c_int32), ("stereo", C.c_int32)] + \
               [(k, C.c_size_t) for k in ("off_flags", "off_counts", "off_kp", "off_desc", "off_u_right", "off_depth", "off_best_dist",
                                          "off_best_idx", "off_nmatched", "bytes")]


class StageTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("ms", C.c_double), ("launches", C.c_int)]


# every symbol include/gfo.h declares (tests check the library exports all of them)
SYMBOLS = [
    "gfo_version", "gfo_build_variant", "gfo_ctx_create", "gfo_ctx_destroy", "gfo_last_error", "gfo_ctx_set_stream",
    "gfo_ctx_synchronize", "gfo_ctx_chain", "gfo_ctx_tables", "gfo_ctx_max_keypoints", "gfo_extract", "gfo_extract_batch",
    "gfo_extract_stereo", "gfo_extract_batch_device", "gfo_batch_counts", "gfo_batch_fetch", "gfo_batch_device_views",
    "gfo_compute_pyramid", "gfo_pyramid_level", "gfo_hamming256", "gfo_stereo_match",
    "gfo_stereo_match_batch", "gfo_stereo_match_sad_batch", "gfo_stereo_fetch", "gfo_search_by_projection", "gfo_search_by_projection_points", "gfo_projection_points_prefix", "gfo_projection_candidates", "gfo_match_candidates", "gfo_search_by_projection_queries_points", "gfo_search_for_fusion", "gfo_search_by_projection_queries",
    "gfo_map_upload", "gfo_search_by_projection_batch", "gfo_projection_fetch", "gfo_projection_device_views", "gfo_search_by_bow", "gfo_search_by_bow_budget", "gfo_search_by_bow_keyframes", "gfo_search_for_triangulation", "gfo_search_for_initialization", "gfo_vocabulary_upload", "gfo_bow_transform", "gfo_compute_bow", "gfo_profile_enable",
    "gfo_profile_read", "gfo_debug_blurred_level", "gfo_debug_level_candidates",
    "gfo_contexts_created", "gfo_arenas_planned", "gfo_kernels_preloaded", "gfo_ctx_id", "gfo_vocabulary_nodes", "gfo_ctx_set_combining", "gfo_combiner_stats", "gfo_ctx_pair", "gfo_combiner_counters", "gfo_tuning_set", "gfo_tuning_get",
    "gfo_batch_deliver", "gfo_deliver_wait", "gfo_host_register", "gfo_host_unregister",
]


def lib_path():
    # GFO_LIB: an instrumented build of the same ABI (tools/pmc_fast_phases.sh); the product is always libgfo.so
    return os.environ.get("GFO_LIB") or os.path.join(_HERE, "libgfo.so")


_lib = None


def _share_hip_runtime_with_torch():
    """One HIP/HSA runtime per process.  PyTorch wheels bundle their own libamdhip64.so.7; if libgfo.so
    pulled in /opt/rocm's copy first, a later `import torch` would bring up a second runtime and fail
    with "No HIP GPUs are available".  Pre-loading torch's copy (same SONAME) makes libgfo bind to it.
    No torch import happens here and nothing changes when torch is absent."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def mapped_hip_runtimes():
    """Distinct files of the HIP / HSA runtime mapped into this process: {soname stem: set(paths)}."""
    found = {"libamdhip64": set(), "libhsa-runtime64": set()}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1] if "/" in line else ""
                base = os.path.basename(path)
                for stem in found:
                    if base.startswith(stem + ".so"):
                        found[stem].add(os.path.realpath(path))
    except OSError:
        pass
    return found


def _check_single_hip_runtime():
    """Two copies of libamdhip64 / libhsa-runtime64 in one process (PyTorch's bundled one plus /opt/rocm's) each
    bring up their own runtime; the second then reports "No HIP GPUs are available" or stalls.  Refuse loudly."""
    # libamdhip64 only: a second libhsa-runtime64 is mapped legitimately by tools that preload /opt/rocm's own copy
    # (rocprofv3's rocprofiler-sdk does) and was never the failing case
    dup = {k: sorted(v) for k, v in mapped_hip_runtimes().items() if len(v) > 1 and k == "libamdhip64"}
    if dup:
        raise GfoError(-2, "two HIP runtimes are mapped into this process: " + "; ".join(f"{k}: {v}" for k, v in dup.items()) +
                       " -- import gf_orb_slam2_amd (or torch) before anything else loads /opt/rocm's libamdhip64, so "
                       "that libgfo.so binds to the copy PyTorch ships")


def load_library():
    """Loads libgfo.so; raises if it has not been built (no fallback of any kind).

    Import order: any order of `import torch` / `import gf_orb_slam2_amd` works -- this function pre-loads PyTorch's
    bundled libamdhip64 / libhsa-runtime64 (same SONAMEs as /opt/rocm's) so that libgfo.so and a later `import torch`
    share ONE runtime, then verifies through /proc/self/maps that a single copy of each is mapped."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise GfoError(-2, f"{p} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    _share_hip_runtime_with_torch()
    L = C.CDLL(p)
    _check_single_hip_runtime()
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    ip = C.POINTER(C.c_int)
    L.gfo_version.restype = i
    L.gfo_ctx_create.argtypes = [C.POINTER(Params), i, C.POINTER(vp)]
    L.gfo_build_variant.argtypes = [i]
    L.gfo_ctx_destroy.argtypes = [vp]
    L.gfo_ctx_destroy.restype = None
    L.gfo_last_error.argtypes = [vp]
    L.gfo_last_error.restype = C.c_char_p
    L.gfo_ctx_set_stream.argtypes = [vp, vp]
    L.gfo_ctx_synchronize.argtypes = [vp]
    L.gfo_ctx_chain.argtypes = [vp, vp, C.c_int]
    L.gfo_ctx_tables.argtypes = [vp, vp, vp, vp, vp, vp]
    L.gfo_ctx_max_keypoints.argtypes = [vp]
    L.gfo_extract.argtypes = [vp, vp, i, i, i, vp, vp, i, ip]
    L.gfo_extract_batch.argtypes = [vp, vp, i, i, i, i, vp, vp, i, vp]
    L.gfo_extract_stereo.argtypes = [vp, vp, vp, i, i, i, C.POINTER(StereoParamsC), vp, vp, vp, vp, i, ip, ip, vp, vp, vp, vp, ip]
    L.gfo_extract_batch_device.argtypes = [vp, vp, i, i, i, sz, sz]
    L.gfo_batch_counts.argtypes = [vp, vp, vp]
    L.gfo_batch_fetch.argtypes = [vp, i, vp, vp, i, ip]
    L.gfo_batch_device_views.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), ip]
    L.gfo_compute_pyramid.argtypes = [vp, vp, i, i, i]
    L.gfo_pyramid_level.argtypes = [vp, i, i, i, vp, i, ip, ip]
    L.gfo_hamming256.argtypes = [vp, vp]
    L.gfo_stereo_match.argtypes = [vp, vp, vp, i, vp, vp, i, vp, i, C.POINTER(StereoParamsC), vp, vp, vp, vp, vp, vp, ip]
    L.gfo_stereo_match_batch.argtypes = [vp, C.POINTER(StereoParamsC)]
    L.gfo_stereo_match_sad_batch.argtypes = [vp, f, f]
    L.gfo_stereo_fetch.argtypes = [vp, i, vp, vp, vp, vp, i, ip]
    L.gfo_search_by_projection.argtypes = [vp, vp, vp, vp, i, vp, i, C.POINTER(FrameBoundsC), vp, vp, i, f, f, vp, vp, vp, ip]
    L.gfo_search_by_projection_points.argtypes = [vp, vp, vp, vp, i, vp, i, C.POINTER(FrameBoundsC), vp, vp, i, f, f, vp, vp, vp, vp, ip]
    L.gfo_projection_points_prefix.argtypes = [vp, i, i, i, vp, vp, ip]
    L.gfo_search_by_projection_queries_points.argtypes = [vp, vp, vp, vp, vp, i, C.POINTER(FrameBoundsC), vp, vp, i, C.POINTER(ProjModeC), vp, vp, vp, vp, ip]
    L.gfo_search_for_fusion.argtypes = [vp, vp, vp, vp, i, C.POINTER(FrameBoundsC), vp, i, vp, vp, i, i, vp]
    L.gfo_projection_candidates.argtypes = [vp, vp, vp, vp, i, vp, i, C.POINTER(FrameBoundsC), vp, vp, i, f, vp, vp, i, ip]
    L.gfo_match_candidates.argtypes = [vp, i, vp, i, f, ip]
    L.gfo_search_by_projection_queries.argtypes = [vp, vp, vp, vp, vp, i, C.POINTER(FrameBoundsC), vp, vp, i, C.POINTER(ProjModeC),
                                                   vp, vp, vp, ip]
    L.gfo_map_upload.argtypes = [vp, vp, i]
    L.gfo_search_by_projection_batch.argtypes = [vp, C.POINTER(ProjectionBatchC)]
    L.gfo_projection_fetch.argtypes = [vp, i, vp, vp, i, ip]
    L.gfo_projection_device_views.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), ip, ip]
    L.gfo_search_by_bow.argtypes = [vp, vp, vp, vp, i, C.POINTER(FeatureVectorC), vp, vp, i, C.POINTER(FeatureVectorC), f, i, vp, ip]
    L.gfo_search_for_initialization.argtypes = [vp, vp, vp, i, vp, vp, vp, i, C.POINTER(FrameBoundsC), i, f, i, vp, ip]
    L.gfo_search_for_triangulation.argtypes = [vp, vp, vp, vp, vp, i, C.POINTER(FeatureVectorC), vp, vp, vp, vp, i, C.POINTER(FeatureVectorC), vp, vp, i, vp, f, f, i, i, vp, ip]
    L.gfo_search_by_bow_keyframes.argtypes = [vp, vp, vp, vp, i, C.POINTER(FeatureVectorC), vp, vp, vp, i, C.POINTER(FeatureVectorC), f, i, vp, ip]
    L.gfo_search_by_bow_budget.argtypes = [vp, vp, vp, vp, i, C.POINTER(FeatureVectorC), vp, vp, i, C.POINTER(FeatureVectorC), f, i, i, vp, ip]
    L.gfo_vocabulary_upload.argtypes = [vp, C.POINTER(VocabularyC)]
    L.gfo_bow_transform.argtypes = [vp, vp, i, i, vp, vp, vp]
    L.gfo_compute_bow.argtypes = [vp, vp, i, i, C.POINTER(BowModeC), vp, vp, ip, vp, vp, vp, ip]
    L.gfo_profile_enable.argtypes = [vp, i]
    L.gfo_profile_read.argtypes = [vp, C.POINTER(StageTime), i, ip, i]
    L.gfo_debug_blurred_level.argtypes = [vp, i, i, vp, i]
    L.gfo_debug_level_candidates.argtypes = [vp, i, i, vp, i, ip]
    L.gfo_batch_deliver.argtypes = [vp, vp, sz, C.POINTER(DeliveryC)]
    L.gfo_deliver_wait.argtypes = [vp]
    L.gfo_host_register.argtypes = [vp, sz]
    L.gfo_host_unregister.argtypes = [vp]
    L.gfo_contexts_created.restype = i
    L.gfo_arenas_planned.restype = i
    L.gfo_kernels_preloaded.restype = i
    L.gfo_ctx_id.argtypes = [vp]
    L.gfo_ctx_id.restype = C.c_uint64
    L.gfo_vocabulary_nodes.argtypes = [vp]
    L.gfo_ctx_set_combining.argtypes = [vp, i]
    L.gfo_combiner_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.gfo_ctx_pair.argtypes = [vp, vp, vp]
    L.gfo_combiner_counters.argtypes = [vp, C.POINTER(C.c_int64), i]
    L.gfo_tuning_set.argtypes = [C.c_char_p, C.c_long]
    L.gfo_tuning_get.argtypes = [C.c_char_p]
    L.gfo_tuning_get.restype = C.c_long
    _lib = L
    return L


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def check(L, ctx, rc, allow=()):
    if rc != 0 and rc not in allow:
        msg = L.gfo_last_error(ctx)
        raise GfoError(rc, msg.decode() if msg else "")
    return rc
