"""ORBmatcher + the stereo association of Frame: host-side mirror of the reference interfaces
(include/ORBmatcher.h:42-310, include/Frame.h:230-267) over the C ABI, on flattened arrays."""
import ctypes as C
from collections import namedtuple

import numpy as np

from ._lib import (FeatureVectorC, FrameBoundsC, GfoError, KEYPOINT_DTYPE, PROJ_QUERY_DTYPE, BowModeC, ProjModeC, ProjectionBatchC, VocabularyC,
                   MAP_POINT_DTYPE, StereoParamsC, check, load_library, ptr)

StereoParams = namedtuple("StereoParams", "n_rows mbf mb min_x")
FrameBounds = namedtuple("FrameBounds", "min_x min_y max_x max_y")


class ORBmatcher:
    """ORBmatcher(nnratio=0.6, checkOri=true) -- ORBmatcher.h:46."""
    TH_LOW = 50           # ORBmatcher.cc:58
    TH_HIGH = 100         # ORBmatcher.cc:57
    HISTO_LENGTH = 30     # ORBmatcher.cc:59

    def __init__(self, nnratio=0.6, checkOri=True, extractor=None):
        self._L = load_library()
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)
        if extractor is None:
            from .extractor import ORBextractor
            extractor = ORBextractor()
        # the matcher borrows the extractor's context (arena, stream): keep the OBJECT, not a copy of its handle,
        # so a closed extractor is seen as closed instead of leaving a dangling gfo_ctx pointer here
        self._ext = extractor

    @property
    def _ctx(self):
        h = self._ext.handle
        if not h or not h.value:
            raise GfoError(-5, "the extractor this matcher was created on has been closed")
        return h

    @staticmethod
    def DescriptorDistance(a, b):
        """ORBmatcher.h:49, ORBmatcher.cc:1768-1784."""
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        assert a.size == 32 and b.size == 32
        return load_library().gfo_hamming256(ptr(a), ptr(b))

    def ComputeStereoMatches(self, keys_l, desc_l, keys_r, desc_r, scale_factors, params, min_d=None, max_d=None):
        """Frame::ComputeStereoMatches_Undistorted(false) on flattened arrays (Frame.cc:1167-1316).
        Returns (nmatched, mvuRight, mvDepth, best_dist, best_idx_r)."""
        kl = np.ascontiguousarray(keys_l, KEYPOINT_DTYPE)
        kr = np.ascontiguousarray(keys_r, KEYPOINT_DTYPE)
        dl = np.ascontiguousarray(desc_l, np.uint8)
        dr = np.ascontiguousarray(desc_r, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        nl, nr = len(kl), len(kr)
        u = np.full(max(nl, 1), -1, np.float32)
        dp = np.full(max(nl, 1), -1, np.float32)
        bd = np.full(max(nl, 1), -1, np.int32)
        bi = np.full(max(nl, 1), -1, np.int32)
        nm = C.c_int()
        p = StereoParamsC(*params)
        if min_d is not None:
            min_d = np.ascontiguousarray(min_d, np.float32)
            max_d = np.ascontiguousarray(max_d, np.float32)
        check(self._L, self._ctx, self._L.gfo_stereo_match(self._ctx, ptr(kl), ptr(dl), nl, ptr(kr), ptr(dr), nr, ptr(sf), len(sf),
                                                           C.byref(p), ptr(min_d), ptr(max_d), ptr(u), ptr(dp), ptr(bd), ptr(bi), C.byref(nm)))
        return nm.value, u[:nl], dp[:nl], bd[:nl], bi[:nl]

    def stereo_match_batch(self, params):
        p = StereoParamsC(*params)
        check(self._L, self._ctx, self._L.gfo_stereo_match_batch(self._ctx, C.byref(p)))

    def stereo_match_sad_batch(self, mbf, mb):
        """Frame::ComputeStereoMatches (SAD variant, Frame.cc:889-1078) on the pairs of the last batch."""
        check(self._L, self._ctx, self._L.gfo_stereo_match_sad_batch(self._ctx, mbf, mb))

    def stereo_fetch(self, pair, cap):
        u = np.zeros(cap, np.float32); dp = np.zeros(cap, np.float32)
        bd = np.zeros(cap, np.int32); bi = np.zeros(cap, np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_stereo_fetch(self._ctx, pair, ptr(u), ptr(dp), ptr(bd), ptr(bi), cap, C.byref(nm)))
        return nm.value, u, dp, bd, bi

    def SearchByProjection(self, keys_un, desc, u_right, scale_factors, bounds, map_points, mp_desc, th=3.0, kp_taken=None):
        """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) -- ORBmatcher.h:64,
        ORBmatcher.cc:155-249.  Returns (nmatches, out_mp, out_score)."""
        kp = np.ascontiguousarray(keys_un, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        mps = np.ascontiguousarray(map_points, MAP_POINT_DTYPE)
        mpd = np.ascontiguousarray(mp_desc, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        n, m = len(kp), len(mps)
        if u_right is not None:
            u_right = np.ascontiguousarray(u_right, np.float32)
        if kp_taken is not None:
            kp_taken = np.ascontiguousarray(kp_taken, np.uint8)
        fb = FrameBoundsC(*bounds)
        out_mp = np.full(max(n, 1), -1, np.int32)
        out_sc = np.zeros(max(n, 1), np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_search_by_projection(self._ctx, ptr(kp), ptr(desc), ptr(u_right), n, ptr(sf), len(sf),
                                                                   C.byref(fb), ptr(mps), ptr(mpd), m, th, self.mfNNratio, ptr(kp_taken),
                                                                   ptr(out_mp), ptr(out_sc), C.byref(nm)))
        return nm.value, out_mp[:n], out_sc[:n]

    POINT_NONE, POINT_RATIO, POINT_FAR = -1, -2, -3   # include/gfo.h GFO_POINT_*

    def SearchByProjectionPoints(self, keys_un, desc, u_right, scale_factors, bounds, map_points, mp_desc, th=3.0, kp_taken=None):
        """The good-feature matchers' common core (gfo_search_by_projection_points): SearchByProjection above plus what every point
        did at its turn -- ORBmatcher::SearchByProjection_OnePoint's return value for the points taken in vector order
        (include/ORBmatcher.h:71-150), which is also the loop of SearchByProjection_Budget (src/ORBmatcher.cc:45-153).
        Returns (nmatches, out_mp, out_score, out_point); out_point[p] >= 0: keypoint | distance << 16, else POINT_*."""
        kp = np.ascontiguousarray(keys_un, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        mps = np.ascontiguousarray(map_points, MAP_POINT_DTYPE)
        mpd = np.ascontiguousarray(mp_desc, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        n, m = len(kp), len(mps)
        if u_right is not None:
            u_right = np.ascontiguousarray(u_right, np.float32)
        if kp_taken is not None:
            kp_taken = np.ascontiguousarray(kp_taken, np.uint8)
        fb = FrameBoundsC(*bounds)
        out_mp = np.full(max(n, 1), -1, np.int32)
        out_sc = np.zeros(max(n, 1), np.int32)
        out_pt = np.full(max(m, 1), -1, np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_search_by_projection_points(self._ctx, ptr(kp), ptr(desc), ptr(u_right), n, ptr(sf), len(sf),
                                                                          C.byref(fb), ptr(mps), ptr(mpd), m, th, self.mfNNratio,
                                                                          ptr(kp_taken), ptr(out_mp), ptr(out_sc), ptr(out_pt), C.byref(nm)))
        return nm.value, out_mp[:n], out_sc[:n], out_pt[:m]

    def points_prefix(self, out_point, prefix, n):
        """The frame after the first `prefix` points of a SearchByProjectionPoints call (gfo_projection_points_prefix): how every early
        exit of the reference's loops -- the clock of SearchByProjection_Budget, the match budget of runBaselineMapMatching -- reads
        the full answer.  Returns (nmatches, out_mp, out_score)."""
        op = np.ascontiguousarray(out_point, np.int32)
        out_mp = np.full(max(n, 1), -1, np.int32)
        out_sc = np.zeros(max(n, 1), np.int32)
        nm = C.c_int()
        rc = self._L.gfo_projection_points_prefix(ptr(op), len(op), int(prefix), n, ptr(out_mp), ptr(out_sc), C.byref(nm))
        if rc != 0:
            raise ValueError("gfo_projection_points_prefix: %d" % rc)
        return nm.value, out_mp[:n], out_sc[:n]

    def GetCandidates(self, keys_un, desc, u_right, scale_factors, bounds, map_points, mp_desc, th=1.0, cap=None):
        """ORBmatcher::GetCandidates for every map point (include/ORBmatcher.h:152-172) with each candidate's distance and mvuRight
        gate (gfo_projection_candidates).  Returns (cand_start[m + 1], cand[total]); an entry is keypoint | octave << 16 |
        distance << 20 | gated << 31, in GetFeaturesInArea's order."""
        kp = np.ascontiguousarray(keys_un, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        mps = np.ascontiguousarray(map_points, MAP_POINT_DTYPE)
        mpd = np.ascontiguousarray(mp_desc, np.uint8)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        n, m = len(kp), len(mps)
        if u_right is not None:
            u_right = np.ascontiguousarray(u_right, np.float32)
        fb = FrameBoundsC(*bounds)
        start = np.zeros(m + 1, np.int32)
        cap = 32 * m if cap is None else int(cap)
        tot = C.c_int()
        while True:
            cand = np.zeros(max(cap, 1), np.uint32)
            rc = self._L.gfo_projection_candidates(self._ctx, ptr(kp), ptr(desc), ptr(u_right), n, ptr(sf), len(sf), C.byref(fb), ptr(mps),
                                                   ptr(mpd), m, th, ptr(start), ptr(cand), cap, C.byref(tot))
            if rc == -3 and tot.value > cap:     # GFO_ERR_CAPACITY: the table is larger than the guess
                cap = tot.value
                continue
            check(self._L, self._ctx, rc)
            return start, cand[:tot.value]

    def MatchCandidates(self, cand, slot_taken):
        """ORBmatcher::MatchCandidates / the candidate loop of SearchByProjection_OnePoint on one point's entries (gfo_match_candidates).
        Returns (keypoint or POINT_*, best distance)."""
        cand = np.ascontiguousarray(cand, np.uint32)
        tk = None if slot_taken is None else np.ascontiguousarray(slot_taken, np.uint8)
        d = C.c_int()
        r = self._L.gfo_match_candidates(ptr(cand), len(cand), ptr(tk), 65536 if tk is None else len(tk), self.mfNNratio, C.byref(d))
        return r, d.value

    # ---- device-resident chain: extract_batch_device -> [stereo_match_batch] -> search_by_projection_batch ----
    def map_upload(self, mp_desc):
        """Descriptors of the local map (MapPoint::GetDescriptor(), vector order); resident until replaced."""
        mpd = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        check(self._L, self._ctx, self._L.gfo_map_upload(self._ctx, ptr(mpd), len(mpd)))
        self._map_m = len(mpd)

    def search_by_projection_batch(self, map_points, bounds, th=3.0, kp_taken=None, stereo=False, device_ptrs=False):
        """SearchByProjection(F, mvpLocalMapPoints, th) for every frame of the last device batch against the resident
        map.  map_points: [frames][m] MAP_POINT_DTYPE (host array), or a raw device address when device_ptrs."""
        if device_ptrs:
            mps_p, tk_p = C.c_void_p(int(map_points)), C.c_void_p(int(kp_taken)) if kp_taken else None
            keep = None
        else:
            mps = np.ascontiguousarray(map_points, MAP_POINT_DTYPE)
            tk = None if kp_taken is None else np.ascontiguousarray(kp_taken, np.uint8)
            keep = (mps, tk)
            mps_p, tk_p = ptr(mps), ptr(tk)
        a = ProjectionBatchC(mps_p, tk_p, 1 if device_ptrs else 0, 1 if stereo else 0, th, self.mfNNratio, FrameBoundsC(*bounds))
        check(self._L, self._ctx, self._L.gfo_search_by_projection_batch(self._ctx, C.byref(a)))
        if keep is not None:   # host arrays are staged with an asynchronous copy: hold them until the stream is done
            self._ext.synchronize()

    def projection_fetch(self, frame, cap):
        out_mp = np.full(max(cap, 1), -1, np.int32)
        out_sc = np.zeros(max(cap, 1), np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_projection_fetch(self._ctx, frame, ptr(out_mp), ptr(out_sc), cap, C.byref(nm)))
        return nm.value, out_mp, out_sc

    def SearchByBoW(self, kf_desc, kf_angle, kf_mp_valid, kf_fv, f_desc, f_angle, f_fv, max_matches=0):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) -- ORBmatcher.h:272, ORBmatcher.cc:270-404.
        kf_fv / f_fv = (node_ids, node_start, items): the CSR of each DBoW2::FeatureVector.
        Returns (nmatches, out_kf_idx)."""
        kf_desc = np.ascontiguousarray(kf_desc, np.uint8)
        f_desc = np.ascontiguousarray(f_desc, np.uint8)
        kf_angle = np.ascontiguousarray(kf_angle, np.float32)
        f_angle = np.ascontiguousarray(f_angle, np.float32)
        kf_mp_valid = np.ascontiguousarray(kf_mp_valid, np.uint8)
        keep = []

        def fv(t):
            ids, start, items = (np.ascontiguousarray(t[0], np.uint32), np.ascontiguousarray(t[1], np.int32),
                                 np.ascontiguousarray(t[2], np.uint32))
            keep.append((ids, start, items))
            return FeatureVectorC(ids.ctypes.data, start.ctypes.data, items.ctypes.data, len(ids))
        a, b = fv(kf_fv), fv(f_fv)
        n_f = len(f_desc)
        out = np.full(max(n_f, 1), -1, np.int32)
        nm = C.c_int()
        # max_matches > 0: the reference compiled with BUDGETING_FEATURE_MATCHING (ORBmatcher.h:36-37)
        check(self._L, self._ctx, self._L.gfo_search_by_bow_budget(self._ctx, ptr(kf_desc), ptr(kf_angle), ptr(kf_mp_valid), len(kf_desc),
                                                                   C.byref(a), ptr(f_desc), ptr(f_angle), n_f, C.byref(b), self.mfNNratio,
                                                                   1 if self.mbCheckOrientation else 0, int(max_matches), ptr(out), C.byref(nm)))
        return nm.value, out[:n_f]

    def SearchByBoWKeyFrames(self, desc1, angle1, mp_valid1, fv1, desc2, angle2, mp_valid2, fv2):
        """ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) -- ORBmatcher.h:273, ORBmatcher.cc:635-768 (loop closing).
        Returns (nmatches, out_idx2[n1]): the keypoint of the second keyframe whose map point lands in vpMatches12[i], -1 = NULL."""
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        angle1 = np.ascontiguousarray(angle1, np.float32); angle2 = np.ascontiguousarray(angle2, np.float32)
        v1 = np.ascontiguousarray(mp_valid1, np.uint8); v2 = np.ascontiguousarray(mp_valid2, np.uint8)
        keep = []

        def fv(t):
            ids, start, items = (np.ascontiguousarray(t[0], np.uint32), np.ascontiguousarray(t[1], np.int32),
                                 np.ascontiguousarray(t[2], np.uint32))
            keep.append((ids, start, items))
            return FeatureVectorC(ids.ctypes.data, start.ctypes.data, items.ctypes.data, len(ids))
        a, b = fv(fv1), fv(fv2)
        n1 = len(desc1)
        out = np.full(max(n1, 1), -1, np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_search_by_bow_keyframes(self._ctx, ptr(desc1), ptr(angle1), ptr(v1), n1, C.byref(a), ptr(desc2),
                                                                      ptr(angle2), ptr(v2), len(desc2), C.byref(b), self.mfNNratio,
                                                                      1 if self.mbCheckOrientation else 0, ptr(out), C.byref(nm)))
        return nm.value, out[:n1]

    def SearchForInitialization(self, kp1, desc1, prev_matched, kp2, desc2, bounds, window_size=100):
        """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) -- ORBmatcher.cc:520-633.
        prev_matched: (n1, 2) float32, updated IN PLACE where a keypoint matched (:626-629).  Returns (nmatches, vnMatches12[n1])."""
        kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE); kp2 = np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        if not (isinstance(prev_matched, np.ndarray) and prev_matched.dtype == np.float32 and prev_matched.flags.c_contiguous and
                prev_matched.shape == (len(kp1), 2)):
            raise ValueError("prev_matched: a C-contiguous float32 array of shape (n1, 2), updated in place")
        fb = FrameBoundsC(*bounds)
        n1 = len(kp1)
        out = np.full(max(n1, 1), -1, np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_search_for_initialization(self._ctx, ptr(kp1), ptr(desc1), n1, ptr(prev_matched), ptr(kp2), ptr(desc2), len(kp2),
                                                                        C.byref(fb), int(window_size), self.mfNNratio,
                                                                        1 if self.mbCheckOrientation else 0, ptr(out), C.byref(nm)))
        return nm.value, out[:n1]

    def SearchForTriangulation(self, kp1, desc1, has_mp1, u_right1, fv1, kp2, desc2, has_mp2, u_right2, fv2, scale_factors2, level_sigma2_2, f12, ex, ey,
                               only_stereo=False):
        """ORBmatcher::SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) -- ORBmatcher.cc:770-935.
        Returns (nmatches, out_idx2[n1])."""
        kp1 = np.ascontiguousarray(kp1, KEYPOINT_DTYPE); kp2 = np.ascontiguousarray(kp2, KEYPOINT_DTYPE)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
        u1 = None if u_right1 is None else np.ascontiguousarray(u_right1, np.float32)
        u2 = None if u_right2 is None else np.ascontiguousarray(u_right2, np.float32)
        sf2 = np.ascontiguousarray(scale_factors2, np.float32); sg2 = np.ascontiguousarray(level_sigma2_2, np.float32)
        f = np.ascontiguousarray(f12, np.float32).reshape(9)
        keep = []

        def fv(t):
            ids, start, items = (np.ascontiguousarray(t[0], np.uint32), np.ascontiguousarray(t[1], np.int32),
                                 np.ascontiguousarray(t[2], np.uint32))
            keep.append((ids, start, items))
            return FeatureVectorC(ids.ctypes.data, start.ctypes.data, items.ctypes.data, len(ids))
        a, b = fv(fv1), fv(fv2)
        n1 = len(kp1)
        out = np.full(max(n1, 1), -1, np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_search_for_triangulation(self._ctx, ptr(kp1), ptr(desc1), ptr(h1), ptr(u1), n1, C.byref(a), ptr(kp2), ptr(desc2),
                                                                       ptr(h2), ptr(u2), len(kp2), C.byref(b), ptr(sf2), ptr(sg2), len(sf2), ptr(f),
                                                                       float(ex), float(ey), 1 if only_stereo else 0,
                                                                       1 if self.mbCheckOrientation else 0, ptr(out), C.byref(nm)))
        return nm.value, out[:n1]

    def SearchByProjectionQueries(self, keys_un, desc, u_right, kp_angle, bounds, queries, q_desc, use_ratio=False,
                                  th_dist=None, kp_taken=None, max_matches=0):
        """The query form both projection overloads reduce to (gfo_search_by_projection_queries); with
        use_ratio=False and mbCheckOrientation it is ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, ...)
        (ORBmatcher.cc:1440-1593) once the caller has projected the last frame's map points.
        Returns (nmatches, out_query, out_score)."""
        kp = np.ascontiguousarray(keys_un, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        q = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE)
        qd = np.ascontiguousarray(q_desc, np.uint8)
        n, m = len(kp), len(q)
        u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        kp_angle = None if kp_angle is None else np.ascontiguousarray(kp_angle, np.float32)
        kp_taken = None if kp_taken is None else np.ascontiguousarray(kp_taken, np.uint8)
        fb = FrameBoundsC(*bounds)
        mode = ProjModeC(1 if use_ratio else 0, self.mfNNratio, self.TH_HIGH if th_dist is None else th_dist,
                         1 if self.mbCheckOrientation else 0, int(max_matches))
        out_q = np.full(max(n, 1), -1, np.int32)
        out_s = np.zeros(max(n, 1), np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_search_by_projection_queries(self._ctx, ptr(kp), ptr(desc), ptr(u_right), ptr(kp_angle), n,
                                                                           C.byref(fb), ptr(q), ptr(qd), m, C.byref(mode), ptr(kp_taken),
                                                                           ptr(out_q), ptr(out_s), C.byref(nm)))
        return nm.value, out_q[:n], out_s[:n]

    def SearchByProjectionQueriesPoints(self, keys_un, desc, u_right, kp_angle, bounds, queries, q_desc, use_ratio=False, th_dist=None,
                                        kp_taken=None):
        """SearchByProjectionQueries + what every query did at its turn (gfo_search_by_projection_queries_points).  With non-blocking
        queries: the independent best-match searches inside ORBmatcher::Fuse / SearchBySim3.
        Returns (nmatches, out_query, out_score, out_point)."""
        kp = np.ascontiguousarray(keys_un, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        q = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE)
        qd = np.ascontiguousarray(q_desc, np.uint8)
        n, m = len(kp), len(q)
        u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        kp_angle = None if kp_angle is None else np.ascontiguousarray(kp_angle, np.float32)
        kp_taken = None if kp_taken is None else np.ascontiguousarray(kp_taken, np.uint8)
        fb = FrameBoundsC(*bounds)
        mode = ProjModeC(1 if use_ratio else 0, self.mfNNratio, self.TH_HIGH if th_dist is None else th_dist,
                         1 if self.mbCheckOrientation else 0, 0)
        out_q = np.full(max(n, 1), -1, np.int32); out_s = np.zeros(max(n, 1), np.int32); out_p = np.full(max(m, 1), -1, np.int32)
        nm = C.c_int()
        check(self._L, self._ctx, self._L.gfo_search_by_projection_queries_points(self._ctx, ptr(kp), ptr(desc), ptr(u_right), ptr(kp_angle), n,
                                                                                  C.byref(fb), ptr(q), ptr(qd), m, C.byref(mode), ptr(kp_taken),
                                                                                  ptr(out_q), ptr(out_s), ptr(out_p), C.byref(nm)))
        return nm.value, out_q[:n], out_s[:n], out_p[:m]


    def SearchForFusion(self, keys_un, desc, u_right, bounds, inv_level_sigma2, queries, q_desc, th_dist=None):
        """The search of ORBmatcher::Fuse(KeyFrame*, MapPoints, th) (ORBmatcher.cc:937-1087) on pre-projected points
        (gfo_search_for_fusion): per point the best keypoint that passes the reprojection-error gate.  Returns out_point[m]."""
        kp = np.ascontiguousarray(keys_un, KEYPOINT_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        q = np.ascontiguousarray(queries, PROJ_QUERY_DTYPE)
        qd = np.ascontiguousarray(q_desc, np.uint8)
        sig = np.ascontiguousarray(inv_level_sigma2, np.float32)
        u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        fb = FrameBoundsC(*bounds)
        out = np.full(max(len(q), 1), -1, np.int32)
        check(self._L, self._ctx, self._L.gfo_search_for_fusion(self._ctx, ptr(kp), ptr(desc), ptr(u_right), len(kp), C.byref(fb), ptr(sig), len(sig),
                                                                ptr(q), ptr(qd), len(q), self.TH_LOW if th_dist is None else th_dist, ptr(out)))
        return out[:len(q)]


class ORBVocabulary:
    """DBoW2 TemplatedVocabulary<FORB> as far as Frame::ComputeBoW needs it (Frame.cc:661-668): a flattened tree
    resident on the device and transform(descriptors, levelsup) -> (BowVector, FeatureVector)."""

    def __init__(self, tree, extractor):
        self._L = load_library()
        self._ext = extractor
        arrs = [np.ascontiguousarray(tree[k]) for k in ("first_child", "n_children", "descriptors", "word_id", "weight")]
        w64 = np.ascontiguousarray(tree["weight64"], np.float64) if "weight64" in tree else None
        v = VocabularyC(*[a.ctypes.data for a in arrs], len(arrs[0]), int(tree["depth"]), None if w64 is None else w64.ctypes.data)
        check(self._L, self._ctx, self._L.gfo_vocabulary_upload(self._ctx, C.byref(v)))

    @property
    def _ctx(self):
        h = self._ext.handle
        if not h or not h.value:
            raise GfoError(-5, "the extractor this vocabulary was uploaded to has been closed")
        return h

    def transform_raw(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        wid = np.zeros(max(n, 1), np.int32); wt = np.zeros(max(n, 1), np.float32); nid = np.zeros(max(n, 1), np.int32)
        check(self._L, self._ctx, self._L.gfo_bow_transform(self._ctx, ptr(desc), n, levelsup, ptr(wid), ptr(wt), ptr(nid)))
        return wid[:n], wt[:n], nid[:n]

    WEIGHTING = {"TF_IDF": 0, "TF": 1, "IDF": 2, "BINARY": 3}    # DBoW2::WeightingType
    NORM = {None: 0, "L1": 1, "L2": 2}                              # DBoW2::LNorm when the scoring normalises

    def compute_bow(self, desc, levelsup=4, weighting="TF_IDF", norm="L1"):
        """Frame::ComputeBoW in full on the device (gfo_compute_bow): returns the two maps flattened in std::map order,
        (bow_words, bow_values[float64]) and the FeatureVector CSR (node_ids, start, items)."""
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        m = max(n, 1)
        bw = np.zeros(m, np.uint32); bv = np.zeros(m, np.float64); fn = np.zeros(m, np.uint32); fs = np.zeros(m + 1, np.int32)
        fi = np.zeros(m, np.uint32)
        nw, nf = C.c_int(), C.c_int()
        mode = BowModeC(self.WEIGHTING[weighting], self.NORM[norm])
        check(self._L, self._ctx, self._L.gfo_compute_bow(self._ctx, ptr(desc), n, levelsup, C.byref(mode), ptr(bw), ptr(bv), C.byref(nw),
                                                          ptr(fn), ptr(fs), ptr(fi), C.byref(nf)))
        return (bw[:nw.value].copy(), bv[:nw.value].copy()), (fn[:nf.value].copy(), fs[:nf.value + 1].copy(), fi[:fs[nf.value]].copy())

    def transform(self, desc, levelsup=4):
        """Host-side fold of transform_raw (float weights; kept for comparison with compute_bow).
        (BowVector as {word: summed weight, L1-normalised}, FeatureVector as CSR (node_ids, node_start, items)) --
        TemplatedVocabulary.h:1140-1212 for TF_IDF weighting with L1 scoring."""
        wid, wt, nid = self.transform_raw(desc, levelsup)
        keep = wt > 0
        bow = {}
        for w_, v_ in zip(wid[keep].tolist(), wt[keep].tolist()):
            bow[w_] = bow.get(w_, 0.0) + v_
        norm = sum(abs(v_) for v_ in bow.values())
        if norm > 0:
            bow = {k: v_ / norm for k, v_ in bow.items()}
        node = np.where(keep, nid, -1)
        ids = np.unique(node[node >= 0]).astype(np.uint32)
        start = np.zeros(len(ids) + 1, np.int32)
        items = []
        for k, n_ in enumerate(ids):
            idx = np.nonzero(node == n_)[0]
            items.append(idx)
            start[k + 1] = start[k] + len(idx)
        items = np.concatenate(items).astype(np.uint32) if items else np.zeros(0, np.uint32)
        return bow, (ids, start, items)
