"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (gf_orb_slam2_amd -> libgfo.so) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORB_ORACLE_LIB: another build of the same file (oracle/Makefile `san`: -fsanitize=address,undefined, for tests/test_sanitizers.py)
_LIB_PATH = os.environ.get("ORB_ORACLE_LIB") or os.path.join(_HERE, "liborb_oracle.so")

KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
MAP_POINT_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"),
                            ("view_cos", "<f4"), ("level", "<i4"), ("flags", "<i4")])
PROJ_QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("radius", "<f4"), ("min_level", "<i4"),
                             ("max_level", "<i4"), ("angle", "<f4"), ("flags", "<i4")])

_VARIANTS_PATH = os.path.join(_HERE, "ocv_variants.json")
OCV_KEYS = {"resize": 0, "atan_fma": 1, "blur_round": 2}     # ORC_OCV_* of orb_oracle.h
TRIG_SHARED, TRIG_LIBM = 0, 1
ROT_UNFUSED, ROT_FMA = 0, 1


class StereoParams(C.Structure):
    _fields_ = [("n_rows", C.c_int), ("mbf", C.c_float), ("mb", C.c_float), ("min_x", C.c_float)]


class FrameBounds(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


def build(force=False):
    if os.environ.get("ORB_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("orb_oracle.c", "orb_oracle.h", "Makefile")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i, f = C.c_void_p, C.c_int, C.c_float
        L.orc_create.restype = vp
        L.orc_create.argtypes = [i, f, i, i, i]
        L.orc_destroy.argtypes = [vp]
        L.orc_set_variant.argtypes = [vp, i, i]
        L.orc_nlevels.argtypes = [vp]
        for name, rt in (("orc_scale_factors", C.POINTER(C.c_float)), ("orc_inv_scale_factors", C.POINTER(C.c_float)),
                         ("orc_level_sigma2", C.POINTER(C.c_float)), ("orc_inv_level_sigma2", C.POINTER(C.c_float)),
                         ("orc_features_per_level", C.POINTER(C.c_int)), ("orc_umax", C.POINTER(C.c_int))):
            getattr(L, name).restype = rt
            getattr(L, name).argtypes = [vp]
        L.orc_extract.argtypes = [vp, vp, i, i, i, vp, vp, i]
        L.orc_compute_pyramid.argtypes = [vp, vp, i, i, i]
        L.orc_level_size.argtypes = [vp, i, C.POINTER(i), C.POINTER(i)]
        L.orc_get_level.argtypes = [vp, i, vp, i]
        L.orc_get_level_padded.argtypes = [vp, i, vp, i]
        L.orc_get_blurred_level.argtypes = [vp, i, vp, i]
        L.orc_level_candidates.argtypes = [vp, i, vp, i]
        L.orc_level_keypoint_count.argtypes = [vp, i]
        L.orc_resize_linear_u8.argtypes = [vp, i, i, i, vp, i, i, i]
        L.orc_gaussian_blur7_u8.argtypes = [vp, i, i, i, vp, i]
        L.orc_fast9_nms.argtypes = [vp, i, i, i, i, vp, i]
        L.orc_fast_atan2.restype = f
        L.orc_fast_atan2.argtypes = [f, f]
        L.orc_sincos.argtypes = [f, C.POINTER(f), C.POINTER(f)]
        L.orc_cv_round.argtypes = [f]
        L.orc_fast_atan2_n.argtypes = [vp, vp, i, vp]
        L.orc_sincos_n.argtypes = [vp, i, vp, vp]
        L.orc_hamming256.argtypes = [vp, vp]
        L.orc_stereo_match.argtypes = [vp, vp, i, vp, vp, i, vp, C.POINTER(StereoParams), vp, vp, vp, vp, vp, vp]
        L.orc_search_by_projection.argtypes = [vp, vp, vp, i, vp, i, C.POINTER(FrameBounds), vp, vp, i, f, f, vp, vp, vp]
        L.orc_features_in_area.argtypes = [vp, i, C.POINTER(FrameBounds), f, f, f, i, i, vp, i]
        L.orc_set_ocv_variant.argtypes = [i, i]
        L.orc_get_ocv_variant.argtypes = [i]
        L.orc_set_gauss_taps.argtypes = [vp]
        L.orc_get_gauss_taps.argtypes = [vp]
        _lib = L
        set_ocv_variants(**load_ocv_variants())      # oracle/ocv_variants.json: the committed [OCV] choices
    return _lib


def load_ocv_variants(path=None):
    import json
    j = json.load(open(path or _VARIANTS_PATH))
    return {k: j[k] for k in ("resize", "atan_fma", "blur_round", "gauss_taps")}


def set_ocv_variants(resize=None, atan_fma=None, blur_round=None, gauss_taps=None):
    """Switch the [OCV] variant table of the C oracle (process-wide).  None leaves a switch as it is."""
    L = lib() if _lib is None else _lib
    for name, v in (("resize", resize), ("atan_fma", atan_fma), ("blur_round", blur_round)):
        if v is not None and L.orc_set_ocv_variant(OCV_KEYS[name], int(v)) != 0:
            raise ValueError(name)
    if gauss_taps is not None:
        t = np.ascontiguousarray(gauss_taps, np.int32)
        assert t.shape == (7,)
        L.orc_set_gauss_taps(_p(t))


def get_ocv_variants():
    L = lib()
    t = np.zeros(7, np.int32)
    L.orc_get_gauss_taps(_p(t))
    out = {name: L.orc_get_ocv_variant(k) for name, k in OCV_KEYS.items()}
    out["gauss_taps"] = [int(x) for x in t]
    return out


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """Mirror of ORBextractor (include/ORBextractor.h:81-119) over the C oracle."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self._h = lib().orc_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self._h:
            raise ValueError("orc_create failed")
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def set_variant(self, trig=TRIG_SHARED, rot=ROT_UNFUSED):
        lib().orc_set_variant(self._h, trig, rot)

    def _farr(self, fn):
        return np.ctypeslib.as_array(fn(self._h), shape=(self.nlevels,)).copy()

    @property
    def scale_factors(self):
        return self._farr(lib().orc_scale_factors)

    @property
    def inv_scale_factors(self):
        return self._farr(lib().orc_inv_scale_factors)

    @property
    def level_sigma2(self):
        return self._farr(lib().orc_level_sigma2)

    @property
    def inv_level_sigma2(self):
        return self._farr(lib().orc_inv_level_sigma2)

    @property
    def features_per_level(self):
        return self._farr(lib().orc_features_per_level)

    @property
    def umax(self):
        return np.ctypeslib.as_array(lib().orc_umax(self._h), shape=(16,)).copy()

    def __call__(self, image, cap=None):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape
        cap = cap or (self.nfeatures * 2 + 64)
        kp = np.zeros(cap, dtype=KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = lib().orc_extract(self._h, _p(image), w, h, w, _p(kp), _p(desc), cap)
        if n > cap:
            return self.__call__(image, cap=n)
        return kp[:n].copy(), desc[:n].copy()

    def compute_pyramid(self, image):
        image = np.ascontiguousarray(image, dtype=np.uint8)
        h, w = image.shape
        lib().orc_compute_pyramid(self._h, _p(image), w, h, w)

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        if lib().orc_level_size(self._h, level, C.byref(w), C.byref(h)) != 0:
            raise IndexError(level)
        return w.value, h.value

    def level(self, level, padded=False, blurred=False):
        w, h = self.level_size(level)
        if padded:
            out = np.zeros((h + 38, w + 38), dtype=np.uint8)
            lib().orc_get_level_padded(self._h, level, _p(out), w + 38)
        elif blurred:
            out = np.zeros((h, w), dtype=np.uint8)
            lib().orc_get_blurred_level(self._h, level, _p(out), w)
        else:
            out = np.zeros((h, w), dtype=np.uint8)
            lib().orc_get_level(self._h, level, _p(out), w)
        return out

    def level_candidates(self, level):
        n = lib().orc_level_candidates(self._h, level, None, 0)
        out = np.zeros((max(n, 1), 3), dtype=np.int32)
        lib().orc_level_candidates(self._h, level, _p(out), n)
        return out[:n]

    def level_keypoint_count(self, level):
        return lib().orc_level_keypoint_count(self._h, level)


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw = src.shape
    dst = np.zeros((dh, dw), dtype=np.uint8)
    lib().orc_resize_linear_u8(_p(src), sw, sh, sw, _p(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    h, w = src.shape
    dst = np.zeros_like(src)
    lib().orc_gaussian_blur7_u8(_p(src), w, h, w, _p(dst), w)
    return dst


def fast9_nms(img, threshold):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((w * h + 1, 3), dtype=np.int32)
    n = lib().orc_fast9_nms(_p(img), w, h, w, threshold, _p(out), w * h)
    return out[:n].copy()


def hamming256(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return lib().orc_hamming256(_p(a), _p(b))


def fast_atan2(y, x):
    return lib().orc_fast_atan2(float(y), float(x))


def sincos(t):
    s, c = C.c_float(), C.c_float()
    lib().orc_sincos(float(t), C.byref(s), C.byref(c))
    return s.value, c.value


def fast_atan2_n(y, x):
    y = np.ascontiguousarray(y, np.float32); x = np.ascontiguousarray(x, np.float32)
    out = np.zeros(len(y), np.float32)
    lib().orc_fast_atan2_n(_p(y), _p(x), len(y), _p(out))
    return out


def sincos_n(t):
    t = np.ascontiguousarray(t, np.float32)
    s = np.zeros(len(t), np.float32); c = np.zeros(len(t), np.float32)
    lib().orc_sincos_n(_p(t), len(t), _p(s), _p(c))
    return s, c


def stereo_match(kl, dl, kr, dr, scale_factors, n_rows, mbf, mb, min_x=0.0, min_d=None, max_d=None, online=False):
    """Frame::ComputeStereoMatches_Undistorted(isOnline) on arrays; online=True: no outlier cut (Frame.cc:1290)"""
    kl = np.ascontiguousarray(kl, dtype=KEYPOINT_DTYPE)
    kr = np.ascontiguousarray(kr, dtype=KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(dl, dtype=np.uint8)
    dr = np.ascontiguousarray(dr, dtype=np.uint8)
    sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
    nl, nr = len(kl), len(kr)
    u_right = np.zeros(max(nl, 1), np.float32)
    depth = np.zeros(max(nl, 1), np.float32)
    best_dist = np.zeros(max(nl, 1), np.int32)
    best_idx = np.zeros(max(nl, 1), np.int32)
    p = StereoParams(n_rows, mbf, mb, min_x)
    if min_d is not None:
        min_d = np.ascontiguousarray(min_d, np.float32)
        max_d = np.ascontiguousarray(max_d, np.float32)
    lib().orc_set_stereo_online(1 if online else 0)
    try:
        nm = lib().orc_stereo_match(_p(kl), _p(dl), nl, _p(kr), _p(dr), nr, _p(sf), C.byref(p), _p(min_d), _p(max_d),
                                    _p(u_right), _p(depth), _p(best_dist), _p(best_idx))
    finally:
        lib().orc_set_stereo_online(0)
    return nm, u_right[:nl], depth[:nl], best_dist[:nl], best_idx[:nl]


class StereoFrame:
    """The stereo members of ONE Frame across calls (orc_stereo_frame, orb_oracle.c): mvuRight, mvDepth, mvStereoMatched, mvDistIdx
    and the row table, reset only by PrepareStereoCandidates (Frame.h:230-263), which ComputeStereoMatches_Undistorted runs only
    when mvRowIndices.size() != nRows (Frame.cc:1173-1176)."""

    def __init__(self, kl, dl, kr, dr, scale_factors, n_rows, mbf, mb, min_x=0.0, delayed=False):
        L = lib()
        vp, i = C.c_void_p, C.c_int
        L.orc_stereo_frame_new.restype = vp
        L.orc_stereo_frame_free.argtypes = [vp]
        L.orc_stereo_frame_prepare.argtypes = [vp, i, vp, i, vp, i]
        L.orc_stereo_frame_clear_matched.argtypes = [vp]
        L.orc_stereo_frame_match.argtypes = [vp, vp, vp, i, vp, vp, i, vp, vp, vp, vp, vp, i, i]
        for nm, rt in (("orc_stereo_frame_uright", C.POINTER(C.c_float)), ("orc_stereo_frame_depth", C.POINTER(C.c_float)),
                       ("orc_stereo_frame_matched", C.POINTER(C.c_ubyte))):
            getattr(L, nm).restype = rt
            getattr(L, nm).argtypes = [vp]
        L.orc_stereo_frame_dist_idx.argtypes = [vp, vp, i]
        self.kl = np.ascontiguousarray(kl, dtype=KEYPOINT_DTYPE)
        self.kr = np.ascontiguousarray(kr, dtype=KEYPOINT_DTYPE)
        self.dl = np.ascontiguousarray(dl, dtype=np.uint8)
        self.dr = np.ascontiguousarray(dr, dtype=np.uint8)
        self.sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        self.p = StereoParams(n_rows, mbf, mb, min_x)
        self.delayed = bool(delayed)
        self._h = L.orc_stereo_frame_new()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_stereo_frame_free(self._h)
            self._h = None

    def prepare(self):
        """Frame::PrepareStereoCandidates called by the caller itself (Tracking.cc:613,649,681)"""
        lib().orc_stereo_frame_prepare(self._h, len(self.kl), _p(self.kr), len(self.kr), _p(self.sf), self.p.n_rows)

    def clear_matched(self):
        """mvStereoMatched = vector<bool>(N,false) of the Frame constructor after its own call (Frame.cc:118)"""
        lib().orc_stereo_frame_clear_matched(self._h)

    def match(self, min_d=None, max_d=None, has_mp=None, online=False):
        """one ComputeStereoMatches_Undistorted(online) call; returns (nmatched, mvuRight, mvDepth, mvDistIdx as (k,2) int32)"""
        if min_d is not None:
            min_d = np.ascontiguousarray(min_d, np.float32)
            max_d = np.ascontiguousarray(max_d, np.float32)
        if has_mp is not None:
            has_mp = np.ascontiguousarray(has_mp, np.uint8)
        nl = len(self.kl)
        nm = lib().orc_stereo_frame_match(self._h, _p(self.kl), _p(self.dl), nl, _p(self.kr), _p(self.dr), len(self.kr), _p(self.sf),
                                          C.byref(self.p), _p(min_d), _p(max_d), _p(has_mp), 1 if online else 0, 1 if self.delayed else 0)
        return (nm,) + self.state()

    def state(self):
        L = lib()
        nl = len(self.kl)
        ur = np.ctypeslib.as_array(L.orc_stereo_frame_uright(self._h), (max(nl, 1),))[:nl].copy()
        dp = np.ctypeslib.as_array(L.orc_stereo_frame_depth(self._h), (max(nl, 1),))[:nl].copy()
        k = L.orc_stereo_frame_dist_idx(self._h, None, 0)
        di = np.zeros((max(k, 1), 2), np.int32)
        L.orc_stereo_frame_dist_idx(self._h, _p(di), k)
        return ur, dp, di[:k]

    def matched(self):
        nl = len(self.kl)
        return np.ctypeslib.as_array(lib().orc_stereo_frame_matched(self._h), (max(nl, 1),))[:nl].copy()


def stereo_match_sad(ext_l, ext_r, kl, dl, kr, dr, mbf, mb):
    """Frame::ComputeStereoMatches (SAD variant); ext_l / ext_r are OracleExtractors that extracted the two images."""
    kl = np.ascontiguousarray(kl, dtype=KEYPOINT_DTYPE); kr = np.ascontiguousarray(kr, dtype=KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(dl, dtype=np.uint8); dr = np.ascontiguousarray(dr, dtype=np.uint8)
    nl, nr = len(kl), len(kr)
    u = np.zeros(max(nl, 1), np.float32); dp = np.zeros(max(nl, 1), np.float32); bd = np.zeros(max(nl, 1), np.int32)
    L = lib()
    vp = C.c_void_p
    L.orc_stereo_match_sad.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_float, vp, vp, vp]
    n = L.orc_stereo_match_sad(ext_l._h, ext_r._h, _p(kl), _p(dl), nl, _p(kr), _p(dr), nr, mbf, mb, _p(u), _p(dp), _p(bd))
    return n, u[:nl], dp[:nl], bd[:nl]


def search_by_projection(kp_un, desc, u_right, scale_factors, bounds, mps, mp_desc, th, nn_ratio, kp_taken=None):
    kp_un = np.ascontiguousarray(kp_un, dtype=KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    mps = np.ascontiguousarray(mps, dtype=MAP_POINT_DTYPE)
    mp_desc = np.ascontiguousarray(mp_desc, dtype=np.uint8)
    sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
    n, m = len(kp_un), len(mps)
    if u_right is not None:
        u_right = np.ascontiguousarray(u_right, np.float32)
    if kp_taken is not None:
        kp_taken = np.ascontiguousarray(kp_taken, np.uint8)
    fb = FrameBounds(*bounds)
    out_mp = np.zeros(max(n, 1), np.int32)
    out_score = np.zeros(max(n, 1), np.int32)
    nm = lib().orc_search_by_projection(_p(kp_un), _p(desc), _p(u_right), n, _p(sf), len(sf), C.byref(fb), _p(mps), _p(mp_desc), m,
                                        th, nn_ratio, _p(kp_taken), _p(out_mp), _p(out_score))
    return nm, out_mp[:n], out_score[:n]


def search_by_projection_budget(kp_un, desc, u_right, scale_factors, bounds, mps, mp_desc, th, nn_ratio, kp_taken=None, clock_trip=0):
    """ORBmatcher::SearchByProjection_Budget (src/ORBmatcher.cc:45-153); clock_trip = k > 0: the k-th reading of the wall clock finds the
    budget spent.  Returns nmatches, out_mp, out_score, out_point (keypoint | distance << 16, -1 / -2 / -3, -4 = never reached), found."""
    kp_un = np.ascontiguousarray(kp_un, dtype=KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    mps = np.ascontiguousarray(mps, dtype=MAP_POINT_DTYPE)
    mp_desc = np.ascontiguousarray(mp_desc, dtype=np.uint8)
    sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
    n, m = len(kp_un), len(mps)
    if u_right is not None:
        u_right = np.ascontiguousarray(u_right, np.float32)
    if kp_taken is not None:
        kp_taken = np.ascontiguousarray(kp_taken, np.uint8)
    fb = FrameBounds(*bounds)
    out_mp = np.zeros(max(n, 1), np.int32); out_score = np.zeros(max(n, 1), np.int32)
    out_point = np.zeros(max(m, 1), np.int32); found = np.zeros(max(m, 1), np.int32)
    L = lib()
    L.orc_search_by_projection_budget.restype = C.c_int
    L.orc_search_by_projection_budget.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                  C.c_float, C.c_float, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    nm = L.orc_search_by_projection_budget(_p(kp_un), _p(desc), _p(u_right), n, _p(sf), len(sf), C.byref(fb), _p(mps), _p(mp_desc), m,
                                           th, nn_ratio, _p(kp_taken), clock_trip, _p(out_mp), _p(out_score), _p(out_point), _p(found))
    return nm, out_mp[:n], out_score[:n], out_point[:m], found[:m]


class ProjectionFrame:
    """What SearchByProjection_OnePoint / GetCandidates / MatchCandidates (include/ORBmatcher.h:71-250) read and write in a Frame, kept
    between calls: the grid, mvpMapPoints (label of the holder, whether it has observations) and mvpMatchScore."""

    def __init__(self, kp_un, desc, u_right, scale_factors, bounds, kp_taken=None):
        self.kp = np.ascontiguousarray(kp_un, dtype=KEYPOINT_DTYPE)
        self.desc = np.ascontiguousarray(desc, dtype=np.uint8)
        self.ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        self.sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
        self.taken = None if kp_taken is None else np.ascontiguousarray(kp_taken, np.uint8)
        self.n = len(self.kp)
        fb = FrameBounds(*bounds)
        L = lib()
        vp = C.c_void_p
        L.orc_proj_frame_new.restype = vp
        L.orc_proj_frame_new.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, vp, vp]
        L.orc_proj_frame_free.argtypes = [vp]
        L.orc_proj_frame_get.argtypes = [vp, vp, vp]
        L.orc_proj_frame_candidates.argtypes = [vp, vp, C.c_float, vp, C.c_int]
        L.orc_proj_frame_one_point.argtypes = [vp, vp, vp, C.c_float, C.c_float, C.c_int, vp]
        L.orc_proj_frame_match_candidates.argtypes = [vp, vp, vp, vp, C.c_int, C.c_float, C.c_float, C.c_int]
        self._h = L.orc_proj_frame_new(_p(self.kp), _p(self.desc), _p(self.ur), self.n, _p(self.sf), len(self.sf), C.byref(fb), _p(self.taken))
        assert self._h

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_proj_frame_free(self._h)
            self._h = None

    def candidates(self, mp, th):
        """GetCandidates: the keypoint indices in GetFeaturesInArea's order."""
        mp = np.ascontiguousarray(mp, dtype=MAP_POINT_DTYPE).reshape(1)
        out = np.zeros(max(self.n, 1), np.int32)
        k = lib().orc_proj_frame_candidates(self._h, _p(mp), th, _p(out), self.n)
        return out[:k].copy()

    def one_point(self, mp, mp_desc, th, nn_ratio, label):
        """SearchByProjection_OnePoint: returns (bestIdx or -1, why) with why 0 matched / 1 ratio / 2 nothing within TH_HIGH / 3 no candidates."""
        mp = np.ascontiguousarray(mp, dtype=MAP_POINT_DTYPE).reshape(1)
        d = np.ascontiguousarray(mp_desc, np.uint8)
        why = C.c_int(0)
        r = lib().orc_proj_frame_one_point(self._h, _p(mp), _p(d), th, nn_ratio, label, C.byref(why))
        return r, why.value

    def match_candidates(self, mp, mp_desc, cand, th, nn_ratio, label):
        mp = np.ascontiguousarray(mp, dtype=MAP_POINT_DTYPE).reshape(1)
        d = np.ascontiguousarray(mp_desc, np.uint8)
        cand = np.ascontiguousarray(cand, np.int32)
        return lib().orc_proj_frame_match_candidates(self._h, _p(mp), _p(d), _p(cand), len(cand), th, nn_ratio, label)

    def state(self):
        out_mp = np.zeros(max(self.n, 1), np.int32); out_score = np.zeros(max(self.n, 1), np.int32)
        lib().orc_proj_frame_get(self._h, _p(out_mp), _p(out_score))
        return out_mp[:self.n], out_score[:self.n]


class VocabularyC(C.Structure):
    _fields_ = [("first_child", C.c_void_p), ("n_children", C.c_void_p), ("descriptors", C.c_void_p), ("word_id", C.c_void_p),
                ("weight", C.c_void_p), ("n_nodes", C.c_int32), ("depth", C.c_int32), ("weight64", C.c_void_p)]


def make_vocabulary(k, depth, seed=0, p_stop=0.05, ragged=True):
    """A synthetic k-ary vocabulary tree of the given depth, flattened breadth-first (children contiguous).
    Node centres are random 256-bit strings refined towards their parent (so descents are not all ties);
    a fraction of the words carries weight 0 (stop words); with ragged=True some nodes have fewer children
    and some branches end early.  Returns dict of arrays + depth."""
    rng = np.random.default_rng(seed)
    first, nch, desc, level = [0], [0], [np.zeros(32, np.uint8)], [0]
    i = 0
    while i < len(first):
        if level[i] < depth and not (ragged and level[i] >= 1 and rng.random() < 0.05):
            kk = int(rng.integers(max(2, k - 3), k + 1)) if ragged else k
            first[i] = len(first)
            nch[i] = kk
            for _ in range(kk):
                d = desc[i].copy()
                nflip = 128 if level[i] == 0 else 128 >> level[i]
                bits = rng.choice(256, nflip, replace=False)
                for b in bits:
                    d[b >> 3] ^= np.uint8(1 << (b & 7))
                first.append(0); nch.append(0); desc.append(d); level.append(level[i] + 1)
        i += 1
    n = len(first)
    nch_a = np.array(nch, np.int32)
    word = np.full(n, -1, np.int32)
    leaves = np.nonzero(nch_a == 0)[0]
    word[leaves] = np.arange(len(leaves), dtype=np.int32)
    w = np.zeros(n, np.float32)
    w[leaves] = rng.uniform(0.1, 5.0, len(leaves)).astype(np.float32)
    w[leaves[rng.random(len(leaves)) < p_stop]] = 0.0
    # DBoW2 keeps Node::weight as a double (idf = log(N / Ni)): values a float cannot hold exactly
    w64 = np.zeros(n, np.float64)
    w64[leaves] = np.log(rng.uniform(1.1, 400.0, len(leaves)))
    w64[w == 0.0] = 0.0
    return {"first_child": np.array(first, np.int32), "n_children": nch_a, "descriptors": np.stack(desc).astype(np.uint8),
            "word_id": word, "weight": w, "weight64": w64, "depth": depth}


def make_vocabulary_full(k, depth, seed=0, p_stop=0.02):
    """The FULL k-ary tree of the given depth, breadth first, vectorised -- for vocabularies of the size the reference loads
    (ORBvoc: k = 10, L = 6, 1 111 111 nodes, 35.5 MB of centres; test/test_Stereo.cpp:87).  Node i's children are k i + 1 .. k i + k;
    a child's centre is its parent's with random bits flipped, sparser with every level (the AND of `level` random words), so that a
    descent is decided at every level and ties are rare but present.  Same dictionary as make_vocabulary."""
    rng = np.random.default_rng(seed)
    n_level = [k ** l for l in range(depth + 1)]
    n = sum(n_level)
    n_internal = n - n_level[-1]
    first = np.zeros(n, np.int32)
    nch = np.zeros(n, np.int32)
    first[:n_internal] = (np.arange(n_internal, dtype=np.int64) * k + 1).astype(np.int32)
    nch[:n_internal] = k
    desc = np.zeros((n, 4), np.uint64)
    start = 1
    prev = desc[0:1]
    for l in range(1, depth + 1):
        cnt = n_level[l]
        mask = rng.integers(0, 1 << 63, (cnt, 4), dtype=np.uint64) << np.uint64(1) | rng.integers(0, 2, (cnt, 4), dtype=np.uint64)
        for _ in range(l - 1):
            mask &= rng.integers(0, 1 << 63, (cnt, 4), dtype=np.uint64) << np.uint64(1) | rng.integers(0, 2, (cnt, 4), dtype=np.uint64)
        cur = np.repeat(prev, k, axis=0) ^ mask
        desc[start:start + cnt] = cur
        prev = cur
        start += cnt
    word = np.full(n, -1, np.int32)
    word[n_internal:] = np.arange(n_level[-1], dtype=np.int32)
    w = np.zeros(n, np.float32)
    w[n_internal:] = rng.uniform(0.1, 5.0, n_level[-1]).astype(np.float32)
    stop = rng.random(n_level[-1]) < p_stop
    w[n_internal:][stop] = 0.0
    w64 = np.zeros(n, np.float64)
    w64[n_internal:] = np.log(rng.uniform(1.1, 400.0, n_level[-1]))
    w64[w == 0.0] = 0.0
    w64[:n_internal] = 0.0
    return {"first_child": first, "n_children": nch, "descriptors": np.ascontiguousarray(desc).view(np.uint8).reshape(n, 32),
            "word_id": word, "weight": w, "weight64": w64, "depth": depth}


def bow_transform(voc, desc, levelsup=4):
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc)
    arrs = [np.ascontiguousarray(voc[k]) for k in ("first_child", "n_children", "descriptors", "word_id", "weight")]
    v = VocabularyC(*[a.ctypes.data for a in arrs], len(arrs[0]), voc["depth"], None)
    wid = np.zeros(max(n, 1), np.int32); wt = np.zeros(max(n, 1), np.float32); nid = np.zeros(max(n, 1), np.int32)
    L = lib()
    L.orc_bow_transform.argtypes = [C.POINTER(VocabularyC), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_bow_transform.restype = None
    L.orc_bow_transform(C.byref(v), _p(desc), n, levelsup, _p(wid), _p(wt), _p(nid))
    return wid[:n], wt[:n], nid[:n]


def compute_bow(voc, desc, levelsup=4, weighting=0, norm=1):
    """TemplatedVocabulary::transform(features, BowVector, FeatureVector, levelsup) with the fold, literally.
    Returns (bow_words, bow_values[float64], fv_node_ids, fv_start, fv_items)."""
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc)
    arrs = [np.ascontiguousarray(voc[k]) for k in ("first_child", "n_children", "descriptors", "word_id", "weight")]
    w64 = np.ascontiguousarray(voc["weight64"], np.float64) if "weight64" in voc else None
    v = VocabularyC(*[a.ctypes.data for a in arrs], len(arrs[0]), voc["depth"], None if w64 is None else w64.ctypes.data)
    m = max(n, 1)
    bw = np.zeros(m, np.uint32); bv = np.zeros(m, np.float64); fn = np.zeros(m, np.uint32); fs = np.zeros(m + 1, np.int32)
    fi = np.zeros(m, np.uint32); nf = C.c_int()
    L = lib()
    vp = C.c_void_p
    L.orc_compute_bow.argtypes = [C.POINTER(VocabularyC), vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_int)]
    nw = L.orc_compute_bow(C.byref(v), _p(desc), n, levelsup, weighting, norm, _p(bw), _p(bv), _p(fn), _p(fs), _p(fi), C.byref(nf))
    return bw[:nw].copy(), bv[:nw].copy(), fn[:nf.value].copy(), fs[:nf.value + 1].copy(), fi[:fs[nf.value]].copy()


def _voc_c(voc):
    arrs = [np.ascontiguousarray(voc[k]) for k in ("first_child", "n_children", "descriptors", "word_id", "weight")]
    w64 = np.ascontiguousarray(voc["weight64"], np.float64) if "weight64" in voc else None
    return VocabularyC(*[a.ctypes.data for a in arrs], len(arrs[0]), voc["depth"], None if w64 is None else w64.ctypes.data), (arrs, w64)


def bow_stream(voc, desc, levelsup=4):
    """The per-feature stream of the descent as DBoW2 holds it: (word u32, weight f64, node u32) -- the input of bow_fold."""
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc)
    v, keep = _voc_c(voc)
    m = max(n, 1)
    word = np.zeros(m, np.uint32); wt = np.zeros(m, np.float64); node = np.zeros(m, np.uint32)
    L = lib()
    L.orc_bow_stream.argtypes = [C.POINTER(VocabularyC), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_bow_stream.restype = None
    L.orc_bow_stream(C.byref(v), _p(desc), n, levelsup, _p(word), _p(wt), _p(node))
    return word[:n], wt[:n], node[:n]


def _fold(fn, word, weight, node, weighting, norm):
    word = np.ascontiguousarray(word, np.uint32); weight = np.ascontiguousarray(weight, np.float64); node = np.ascontiguousarray(node, np.uint32)
    n = len(word)
    m = max(n, 1)
    bw = np.zeros(m, np.uint32); bv = np.zeros(m, np.float64); fnn = np.zeros(m, np.uint32); fs = np.zeros(m + 1, np.int32)
    fi = np.zeros(m, np.uint32); nf = C.c_int()
    vp = C.c_void_p
    fn.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_int)]
    fn.restype = C.c_int
    nw = fn(_p(word), _p(weight), _p(node), n, weighting, norm, _p(bw), _p(bv), _p(fnn), _p(fs), _p(fi), C.byref(nf))
    return bw[:nw].copy(), bv[:nw].copy(), fnn[:nf.value].copy(), fs[:nf.value + 1].copy(), fi[:fs[nf.value]].copy()


def bow_fold(word, weight, node, weighting=0, norm=1):
    """orc_bow_fold: the fold of a (word, weight, node) stream into BowVector / FeatureVector (the oracle's restatement).
    weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; norm: 0 none, 1 L1, 2 L2."""
    return _fold(lib().orc_bow_fold, word, weight, node, weighting, norm)


_REF_DIR = os.path.join(_HERE, "_ref")
_REF_FOLD = os.path.join(_REF_DIR, "libdbow2_fold.so")
REFERENCE_ROOT = os.environ.get("GFO_REFERENCE_ROOT", "/root/reference")


def build_ref():
    """oracle/_ref/libdbow2_fold.so from the reference's own DBoW2 BowVector.cpp / FeatureVector.cpp, compiled unmodified where they
    lie (oracle/Makefile, target ref).  Only where the reference tree exists (the build container); returns the path or None."""
    if not os.path.exists(os.path.join(REFERENCE_ROOT, "Thirdparty", "DBoW2", "DBoW2", "BowVector.cpp")):
        return _REF_FOLD if os.path.exists(_REF_FOLD) else None
    subprocess.check_call(["make", "-C", _HERE, "-s", "ref", f"REFERENCE={REFERENCE_ROOT}"])
    return _REF_FOLD


_ref_lib = None


def ref_available():
    return os.path.exists(_REF_FOLD)


def ref_bow_fold(word, weight, node, weighting=0, norm=1):
    """The same fold through the reference's compiled std::map classes (oracle/_ref); raises FileNotFoundError without the build."""
    global _ref_lib
    if _ref_lib is None:
        if not os.path.exists(_REF_FOLD):
            raise FileNotFoundError(_REF_FOLD)
        _ref_lib = C.CDLL(_REF_FOLD)
    return _fold(_ref_lib.ref_bow_fold, word, weight, node, weighting, norm)


class ProjMode(C.Structure):
    _fields_ = [("use_ratio", C.c_int32), ("nn_ratio", C.c_float), ("th_dist", C.c_int32), ("check_orientation", C.c_int32)]


def search_by_projection_queries(kp_un, desc, u_right, kp_angle, bounds, queries, q_desc, use_ratio, nn_ratio, th_dist,
                                 check_orientation, kp_taken=None):
    kp_un = np.ascontiguousarray(kp_un, dtype=KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    queries = np.ascontiguousarray(queries, dtype=PROJ_QUERY_DTYPE)
    q_desc = np.ascontiguousarray(q_desc, dtype=np.uint8)
    n, m = len(kp_un), len(queries)
    u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    kp_angle = None if kp_angle is None else np.ascontiguousarray(kp_angle, np.float32)
    kp_taken = None if kp_taken is None else np.ascontiguousarray(kp_taken, np.uint8)
    fb = FrameBounds(*bounds)
    mode = ProjMode(1 if use_ratio else 0, nn_ratio, th_dist, 1 if check_orientation else 0)
    out_q = np.zeros(max(n, 1), np.int32)
    out_s = np.zeros(max(n, 1), np.int32)
    L = lib()
    vp = C.c_void_p
    L.orc_search_by_projection_queries.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(FrameBounds), vp, vp, C.c_int,
                                                   C.POINTER(ProjMode), vp, vp, vp]
    nm = L.orc_search_by_projection_queries(_p(kp_un), _p(desc), _p(u_right), _p(kp_angle), n, C.byref(fb), _p(queries),
                                            _p(q_desc), m, C.byref(mode), _p(kp_taken), _p(out_q), _p(out_s))
    return nm, out_q[:n], out_s[:n]


def search_by_projection_queries_points(kp_un, desc, u_right, kp_angle, bounds, queries, q_desc, use_ratio, nn_ratio, th_dist,
                                        check_orientation, kp_taken=None):
    """search_by_projection_queries + what every query did at its turn (keypoint | distance << 16, -1 / -2 / -3)."""
    kp_un = np.ascontiguousarray(kp_un, dtype=KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    queries = np.ascontiguousarray(queries, dtype=PROJ_QUERY_DTYPE)
    q_desc = np.ascontiguousarray(q_desc, dtype=np.uint8)
    n, m = len(kp_un), len(queries)
    u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    kp_angle = None if kp_angle is None else np.ascontiguousarray(kp_angle, np.float32)
    kp_taken = None if kp_taken is None else np.ascontiguousarray(kp_taken, np.uint8)
    fb = FrameBounds(*bounds)
    mode = ProjMode(1 if use_ratio else 0, nn_ratio, th_dist, 1 if check_orientation else 0)
    out_q = np.zeros(max(n, 1), np.int32); out_s = np.zeros(max(n, 1), np.int32); out_p = np.zeros(max(m, 1), np.int32)
    L = lib()
    vp = C.c_void_p
    L.orc_search_by_projection_queries_points.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(FrameBounds), vp, vp, C.c_int,
                                                          C.POINTER(ProjMode), vp, vp, vp, vp]
    nm = L.orc_search_by_projection_queries_points(_p(kp_un), _p(desc), _p(u_right), _p(kp_angle), n, C.byref(fb), _p(queries),
                                                   _p(q_desc), m, C.byref(mode), _p(kp_taken), _p(out_q), _p(out_s), _p(out_p))
    return nm, out_q[:n], out_s[:n], out_p[:m]


def search_for_fusion(kp_un, desc, u_right, bounds, inv_level_sigma2, queries, q_desc, th_dist=50):
    """The search of ORBmatcher::Fuse(KeyFrame*, MapPoints, th), ORBmatcher.cc:1000-1063, on pre-projected points.  Returns out_point[m]."""
    kp_un = np.ascontiguousarray(kp_un, dtype=KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    queries = np.ascontiguousarray(queries, dtype=PROJ_QUERY_DTYPE)
    q_desc = np.ascontiguousarray(q_desc, dtype=np.uint8)
    sig = np.ascontiguousarray(inv_level_sigma2, np.float32)
    u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    fb = FrameBounds(*bounds)
    out = np.full(max(len(queries), 1), -1, np.int32)
    L = lib()
    vp = C.c_void_p
    L.orc_search_for_fusion.argtypes = [vp, vp, vp, C.c_int, C.POINTER(FrameBounds), vp, vp, vp, C.c_int, C.c_int, vp]
    L.orc_search_for_fusion(_p(kp_un), _p(desc), _p(u_right), len(kp_un), C.byref(fb), _p(sig), _p(queries), _p(q_desc), len(queries), th_dist, _p(out))
    return out[:len(queries)]


def search_by_projection_kf(kp_un, desc, kp_angle, bounds, queries, q_desc, orb_dist, check_orientation, kp_set=None):
    """ORBmatcher::SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist), ORBmatcher.cc:1595-1721,
    on pre-projected map points (literal statement: any set keypoint is skipped)."""
    kp_un = np.ascontiguousarray(kp_un, dtype=KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    queries = np.ascontiguousarray(queries, dtype=PROJ_QUERY_DTYPE)
    q_desc = np.ascontiguousarray(q_desc, dtype=np.uint8)
    n, m = len(kp_un), len(queries)
    kp_angle = np.ascontiguousarray(kp_angle, np.float32)
    kp_set = None if kp_set is None else np.ascontiguousarray(kp_set, np.uint8)
    fb = FrameBounds(*bounds)
    out_q = np.zeros(max(n, 1), np.int32)
    out_s = np.zeros(max(n, 1), np.int32)
    L = lib()
    vp = C.c_void_p
    L.orc_search_by_projection_kf.argtypes = [vp, vp, vp, C.c_int, C.POINTER(FrameBounds), vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    nm = L.orc_search_by_projection_kf(_p(kp_un), _p(desc), _p(kp_angle), n, C.byref(fb), _p(queries), _p(q_desc), m, int(orb_dist),
                                       1 if check_orientation else 0, _p(kp_set), _p(out_q), _p(out_s))
    return nm, out_q[:n], out_s[:n]


class FeatureVectorC(C.Structure):
    _fields_ = [("node_ids", C.c_void_p), ("node_start", C.c_void_p), ("items", C.c_void_p), ("n_nodes", C.c_int32)]


def make_feature_vector(node_of_keypoint):
    """CSR of a DBoW2::FeatureVector from the node id of every keypoint (-1 = not in the vector):
    node ids ascending, keypoint indices ascending inside a node (as DBoW2 appends them)."""
    node_of_keypoint = np.asarray(node_of_keypoint)
    ids = np.unique(node_of_keypoint[node_of_keypoint >= 0]).astype(np.uint32)
    start = np.zeros(len(ids) + 1, np.int32)
    items = []
    for k, nid in enumerate(ids):
        idx = np.nonzero(node_of_keypoint == nid)[0]
        items.append(idx)
        start[k + 1] = start[k] + len(idx)
    items = np.concatenate(items).astype(np.uint32) if items else np.zeros(0, np.uint32)
    return ids, start, items


def _fv_struct(fv):
    ids, start, items = (np.ascontiguousarray(fv[0], np.uint32), np.ascontiguousarray(fv[1], np.int32),
                         np.ascontiguousarray(fv[2], np.uint32))
    return FeatureVectorC(ids.ctypes.data, start.ctypes.data, items.ctypes.data, len(ids)), (ids, start, items)


def search_by_bow(kf_desc, kf_angle, kf_mp_valid, kf_fv, f_desc, f_angle, f_fv, nn_ratio, check_orientation=True):
    kf_desc = np.ascontiguousarray(kf_desc, np.uint8)
    f_desc = np.ascontiguousarray(f_desc, np.uint8)
    kf_angle = np.ascontiguousarray(kf_angle, np.float32)
    f_angle = np.ascontiguousarray(f_angle, np.float32)
    kf_mp_valid = np.ascontiguousarray(kf_mp_valid, np.uint8)
    a, keep_a = _fv_struct(kf_fv)
    b, keep_b = _fv_struct(f_fv)
    out = np.full(max(len(f_desc), 1), -1, np.int32)
    L = lib()
    L.orc_search_by_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(FeatureVectorC), C.c_void_p, C.c_void_p,
                                    C.c_int, C.POINTER(FeatureVectorC), C.c_float, C.c_int, C.c_void_p]
    nm = L.orc_search_by_bow(_p(kf_desc), _p(kf_angle), _p(kf_mp_valid), len(kf_desc), C.byref(a), _p(f_desc), _p(f_angle),
                             len(f_desc), C.byref(b), nn_ratio, 1 if check_orientation else 0, _p(out))
    return nm, out[:len(f_desc)]


def search_by_bow_keyframes(desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2, nn_ratio, check_orientation=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12), ORBmatcher.cc:635-768.  Returns (nmatches, out12[n1])."""
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    angle1 = np.ascontiguousarray(angle1, np.float32); angle2 = np.ascontiguousarray(angle2, np.float32)
    valid1 = np.ascontiguousarray(valid1, np.uint8); valid2 = np.ascontiguousarray(valid2, np.uint8)
    a, keep_a = _fv_struct(fv1)
    b, keep_b = _fv_struct(fv2)
    out = np.full(max(len(desc1), 1), -1, np.int32)
    L = lib()
    vp = C.c_void_p
    L.orc_search_by_bow_keyframes.argtypes = [vp, vp, vp, C.c_int, C.POINTER(FeatureVectorC), vp, vp, vp, C.c_int, C.POINTER(FeatureVectorC),
                                              C.c_float, C.c_int, vp]
    nm = L.orc_search_by_bow_keyframes(_p(desc1), _p(angle1), _p(valid1), len(desc1), C.byref(a), _p(desc2), _p(angle2), _p(valid2), len(desc2),
                                       C.byref(b), nn_ratio, 1 if check_orientation else 0, _p(out))
    return nm, out[:len(desc1)]


def search_for_initialization(kp1, desc1, prev_matched, kp2, desc2, bounds, window_size=100, nn_ratio=0.9, check_orientation=True):
    """ORBmatcher::SearchForInitialization, ORBmatcher.cc:520-633.  prev_matched (n1, 2) float32 is updated in place.
    Returns (nmatches, vnMatches12[n1])."""
    kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE); kp2 = np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    assert prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous and prev_matched.shape == (len(kp1), 2)
    fb = FrameBounds(*bounds)
    out = np.full(max(len(kp1), 1), -1, np.int32)
    L = lib()
    vp = C.c_void_p
    L.orc_search_for_initialization.argtypes = [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(FrameBounds), C.c_int, C.c_float, C.c_int, vp]
    nm = L.orc_search_for_initialization(_p(kp1), _p(desc1), len(kp1), _p(prev_matched), _p(kp2), _p(desc2), len(kp2), C.byref(fb), int(window_size),
                                         nn_ratio, 1 if check_orientation else 0, _p(out))
    return nm, out[:len(kp1)]


def search_for_triangulation(kp1, desc1, has_mp1, u_right1, fv1, kp2, desc2, has_mp2, u_right2, fv2, scale_factors2, level_sigma2_2, f12, ex, ey,
                             only_stereo=False, check_orientation=True):
    """ORBmatcher::SearchForTriangulation, ORBmatcher.cc:770-935.  Returns (nmatches, out12[n1])."""
    kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE); kp2 = np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
    u1 = None if u_right1 is None else np.ascontiguousarray(u_right1, np.float32)
    u2 = None if u_right2 is None else np.ascontiguousarray(u_right2, np.float32)
    sf2 = np.ascontiguousarray(scale_factors2, np.float32); sg2 = np.ascontiguousarray(level_sigma2_2, np.float32)
    f = np.ascontiguousarray(f12, np.float32).reshape(9)
    a, keep_a = _fv_struct(fv1)
    b, keep_b = _fv_struct(fv2)
    out = np.full(max(len(kp1), 1), -1, np.int32)
    L = lib()
    vp = C.c_void_p
    L.orc_search_for_triangulation.argtypes = [vp, vp, vp, vp, C.c_int, C.POINTER(FeatureVectorC), vp, vp, vp, vp, C.c_int, C.POINTER(FeatureVectorC), vp, vp,
                                               vp, C.c_float, C.c_float, C.c_int, C.c_int, vp]
    nm = L.orc_search_for_triangulation(_p(kp1), _p(desc1), _p(h1), _p(u1), len(kp1), C.byref(a), _p(kp2), _p(desc2), _p(h2), _p(u2), len(kp2), C.byref(b),
                                        _p(sf2), _p(sg2), _p(f), ex, ey, 1 if only_stereo else 0, 1 if check_orientation else 0, _p(out))
    return nm, out[:len(kp1)]


def three_maxima(sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    i1, i2, i3 = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    L = lib()
    L.orc_three_maxima.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.orc_three_maxima(_p(sizes), len(sizes), C.byref(i1), C.byref(i2), C.byref(i3))
    return i1.value, i2.value, i3.value


def features_in_area(kp_un, bounds, x, y, r, min_level=-1, max_level=-1):
    kp_un = np.ascontiguousarray(kp_un, dtype=KEYPOINT_DTYPE)
    n = len(kp_un)
    fb = FrameBounds(*bounds)
    out = np.zeros(max(n, 1), np.int32)
    k = lib().orc_features_in_area(_p(kp_un), n, C.byref(fb), x, y, r, min_level, max_level, _p(out), n)
    return out[:k].copy()


class feature_budget:
    """with feature_budget(150): ... -- the matchers inside run as compiled with BUDGETING_FEATURE_MATCHING (ORBmatcher.h:36-37)"""

    def __init__(self, max_matches):
        self.k = int(max_matches)

    def __enter__(self):
        lib().orc_set_feature_budget(self.k)
        return self

    def __exit__(self, *exc):
        lib().orc_set_feature_budget(0)
        return False
