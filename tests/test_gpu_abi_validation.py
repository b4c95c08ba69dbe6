"""Malformed inputs across the C ABI (include/gfo.h:8-9: "nothing faults across the ABI").  One table: every host-array entry point
is called with ONE argument damaged -- an index past an array, a CSR that is not one, an octave outside the pyramid, a null where
data is promised -- and must return GFO_ERR_INVALID with its output buffers untouched; the valid call that follows on the same
context must still equal the oracle bit for bit (a refused call leaves no state behind).

The CSR of gfo_search_by_bow is the caller's flattening of a DBoW2::FeatureVector (adapter/matchers_gfo.cc FlatFeatVec): a stale
mFeatVec -- another frame's, or one made before the keypoint list shrank -- used to index the descriptor rows on the device."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

GFO_ERR_INVALID = -1
SENT_I, SENT_F = np.int32(0x5A5A5A5A), np.float32(12345.5)


@pytest.fixture(scope="module")
def env(oracle):
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd import _lib
    e = G.ORBextractor(2000, 1.2, 8, 20, 7)
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)
    kr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_kp.bin"), kd)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    dr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_desc.bin"), np.uint8).reshape(-1, 32)
    yield {"G": G, "lib": _lib, "L": _lib.load_library(), "ext": e, "ctx": e.handle, "kl": kl, "kr": kr, "dl": dl, "dr": dr,
           "sf": np.asarray(e.GetScaleFactors(), np.float32), "oracle": oracle}
    e.close()


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------------------------------------------------------------------
# gfo_search_by_bow
# ------------------------------------------------------------------------------------------------------------------------------
def _bow_inputs(env):
    O = env["oracle"]
    dl, kl = env["dl"], env["kl"]
    rng = np.random.default_rng(11)
    fd = dl.copy()
    bits = rng.integers(0, 256, len(fd))
    fd[np.arange(len(fd)), bits >> 3] ^= (1 << (bits & 7)).astype(np.uint8)
    node_k = (dl[:, 0] >> 3).astype(np.int64)
    node_f = (fd[:, 0] >> 3).astype(np.int64)
    valid = np.ones(len(dl), np.uint8)
    ka = kl["angle"].astype(np.float32).copy()
    return {"kd": dl, "ka": ka, "valid": valid, "kfv": [a.copy() for a in O.make_feature_vector(node_k)], "fd": fd, "fa": ka.copy(),
            "ffv": [a.copy() for a in O.make_feature_vector(node_f)]}


def _bow_call(env, a, out, nm):
    FV = env["lib"].FeatureVectorC
    keep = []

    def fv(t):
        ids = None if t[0] is None else np.ascontiguousarray(t[0], np.uint32)
        start = None if t[1] is None else np.ascontiguousarray(t[1], np.int32)
        items = None if t[2] is None else np.ascontiguousarray(t[2], np.uint32)
        keep.append((ids, start, items))
        n_nodes = t[3] if len(t) > 3 else len(ids)
        return FV(None if ids is None else ids.ctypes.data, None if start is None else start.ctypes.data,
                  None if items is None else items.ctypes.data, n_nodes)
    ka_, kb_ = fv(a["kfv"]), fv(a["ffv"])
    return env["L"].gfo_search_by_bow(env["ctx"], _p(a["kd"]), _p(a["ka"]), _p(a["valid"]), len(a["kd"]), C.byref(ka_), _p(a["fd"]),
                                      _p(a["fa"]), len(a["fd"]), C.byref(kb_), C.c_float(0.7), 1, _p(out), C.byref(nm))


def _damage_bow(a, what):
    kfv, ffv = a["kfv"], a["ffv"]
    n_f, n_k = len(a["fd"]), len(a["kd"])
    if what == "frame item == n_f":
        ffv[2][len(ffv[2]) // 2] = n_f
    elif what == "frame item huge":
        ffv[2][3] = 0xFFFFFFF0
    elif what == "keyframe item == n_kf":
        kfv[2][0] = n_k
    elif what == "keyframe item from a larger frame":
        kfv[2][-1] = n_k + 500
    elif what == "node_start decreases":
        ffv[1][5] = ffv[1][4] - 1
    elif what == "node_start[0] != 0":
        kfv[1][0] = 1
    elif what == "node_start ends past the items (negative run)":
        kfv[1][len(kfv[1]) // 2] = kfv[1][-1] + 7
    elif what == "node_ids not ascending":
        ffv[0][2], ffv[0][3] = ffv[0][3], ffv[0][2]
    elif what == "node_ids repeated":
        kfv[0][4] = kfv[0][3]
    elif what == "negative node count":
        a["ffv"] = ffv + [-3]
    elif what == "null items with a non-empty CSR":
        a["kfv"] = [kfv[0], kfv[1], None, len(kfv[0])]
    elif what == "null node_start":
        a["ffv"] = [ffv[0], None, ffv[2], len(ffv[0])]
    elif what == "frame angle NaN":
        a["fa"][100] = np.nan
    elif what == "keyframe angle 1e9":
        a["ka"][7] = 1e9
    elif what == "frame angle -1 (cv::KeyPoint's 'no orientation')":
        a["fa"][0] = -1.0
    else:
        raise KeyError(what)


BOW_CASES = ["frame item == n_f", "frame item huge", "keyframe item == n_kf", "keyframe item from a larger frame", "node_start decreases",
             "node_start[0] != 0", "node_start ends past the items (negative run)", "node_ids not ascending", "node_ids repeated",
             "negative node count", "null items with a non-empty CSR", "null node_start", "frame angle NaN", "keyframe angle 1e9",
             "frame angle -1 (cv::KeyPoint's 'no orientation')"]


def test_search_by_bow_refuses_a_damaged_feature_vector(env):
    O = env["oracle"]
    good = _bow_inputs(env)
    ref = O.search_by_bow(good["kd"], good["ka"], good["valid"], tuple(good["kfv"]), good["fd"], good["fa"], tuple(good["ffv"]), 0.7, True)
    assert ref[0] > 300
    for what in BOW_CASES:
        a = _bow_inputs(env)
        _damage_bow(a, what)
        out = np.full(len(a["fd"]), SENT_I, np.int32)
        nm = C.c_int(int(SENT_I))
        rc = _bow_call(env, a, out, nm)
        assert rc == GFO_ERR_INVALID, (what, rc)
        assert (out == SENT_I).all() and nm.value == int(SENT_I), f"{what}: outputs were written by a refused call"
        msg = env["L"].gfo_last_error(env["ctx"]).decode()
        assert "feature vector" in msg or "bad argument" in msg or "angle" in msg, (what, msg)
        # the next valid call on the same context is still the oracle's answer
        out = np.full(len(good["fd"]), SENT_I, np.int32)
        nm = C.c_int(-7)
        assert _bow_call(env, good, out, nm) == 0, what
        assert nm.value == ref[0], what
        np.testing.assert_array_equal(out, ref[1])


# ------------------------------------------------------------------------------------------------------------------------------
# gfo_stereo_match
# ------------------------------------------------------------------------------------------------------------------------------
def _stereo_call(env, a, outs, nm):
    SP = env["lib"].StereoParamsC
    p = SP(*a["p"])
    return env["L"].gfo_stereo_match(env["ctx"], _p(a["kl"]), _p(a["dl"]), a.get("nl", 0 if a["kl"] is None else len(a["kl"])),
                                     _p(a["kr"]), _p(a["dr"]), a.get("nr", 0 if a["kr"] is None else len(a["kr"])), _p(a["sf"]),
                                     a.get("nlevels", len(a["sf"])), C.byref(p), _p(a.get("min_d")), _p(a.get("max_d")),
                                     _p(outs[0]), _p(outs[1]), _p(outs[2]), _p(outs[3]), C.byref(nm))


def _stereo_inputs(env):
    return {"kl": env["kl"].copy(), "dl": env["dl"], "kr": env["kr"].copy(), "dr": env["dr"], "sf": env["sf"],
            "p": (480, 47.906, 47.906 / 435.2, 0.0)}


def _damage_stereo(a, what):
    if what == "left octave == nlevels":
        a["kl"]["octave"][17] = 8
    elif what == "left octave negative":
        a["kl"]["octave"][0] = -1
    elif what == "right octave huge":
        a["kr"]["octave"][-1] = 1 << 20
    elif what == "n_rows zero":
        a["p"] = (0,) + a["p"][1:]
    elif what == "n_rows beyond 8192":
        a["p"] = (100000,) + a["p"][1:]
    elif what == "nlevels beyond the table":
        a["nlevels"] = 64
    elif what == "null left descriptors":
        a["nl"] = len(a["kl"]); a["dl"] = None
    elif what == "null right keypoints":
        a["nr"] = len(a["kr"]); a["kr"] = None
    elif what == "min_d without max_d":
        a["min_d"] = np.zeros(len(a["kl"]), np.float32)
    elif what == "negative count":
        a["nl"] = -5
    else:
        raise KeyError(what)


STEREO_CASES = ["left octave == nlevels", "left octave negative", "right octave huge", "n_rows zero", "n_rows beyond 8192",
                "nlevels beyond the table", "null left descriptors", "null right keypoints", "min_d without max_d", "negative count"]


def test_stereo_match_refuses_malformed_arrays(env):
    O = env["oracle"]
    good = _stereo_inputs(env)
    ref = O.stereo_match(good["kl"], good["dl"], good["kr"], good["dr"], good["sf"], *good["p"])
    n = len(good["kl"])
    for what in STEREO_CASES:
        a = _stereo_inputs(env)
        _damage_stereo(a, what)
        outs = [np.full(n, SENT_F, np.float32), np.full(n, SENT_F, np.float32), np.full(n, SENT_I, np.int32), np.full(n, SENT_I, np.int32)]
        nm = C.c_int(int(SENT_I))
        rc = _stereo_call(env, a, outs, nm)
        assert rc == GFO_ERR_INVALID, (what, rc)
        assert all((o == (SENT_F if o.dtype == np.float32 else SENT_I)).all() for o in outs) and nm.value == int(SENT_I), what
        outs = [np.full(n, SENT_F, np.float32), np.full(n, SENT_F, np.float32), np.full(n, SENT_I, np.int32), np.full(n, SENT_I, np.int32)]
        nm = C.c_int(-7)
        assert _stereo_call(env, good, outs, nm) == 0, what
        assert nm.value == ref[0], what
        for got, want in zip(outs, ref[1:]):
            assert got.tobytes() == want.tobytes(), what


# ------------------------------------------------------------------------------------------------------------------------------
# gfo_search_by_projection / gfo_search_by_projection_queries
# ------------------------------------------------------------------------------------------------------------------------------
def _proj_inputs(env):
    O = env["oracle"]
    kl, dl = env["kl"], env["dl"]
    rng = np.random.default_rng(3)
    n, m = len(kl), 1200
    src = rng.integers(0, n, m)
    mps = np.zeros(m, O.MAP_POINT_DTYPE)
    mps["proj_x"] = kl["x"][src] + rng.normal(0, 2, m)
    mps["proj_y"] = kl["y"][src] + rng.normal(0, 2, m)
    mps["proj_xr"] = mps["proj_x"] - 5
    mps["level"] = kl["octave"][src]
    mps["view_cos"] = 1.0
    mps["flags"] = 1 | 4
    return {"kp": kl.copy(), "desc": dl, "ur": np.full(n, -1, np.float32), "sf": env["sf"], "fb": (0.0, 0.0, 752.0, 480.0), "mps": mps,
            "mpd": dl[src].copy(), "th": 3.0, "ratio": 0.8}


def _proj_call(env, a, out_mp, out_sc, nm):
    FB = env["lib"].FrameBoundsC
    fb = FB(*a["fb"])
    n = a.get("n", len(a["kp"]))
    m = a.get("m", 0 if a["mps"] is None else len(a["mps"]))
    return env["L"].gfo_search_by_projection(env["ctx"], _p(a["kp"]), _p(a["desc"]), _p(a["ur"]), n, _p(a["sf"]), a.get("nlevels", len(a["sf"])),
                                             C.byref(fb), _p(a["mps"]), _p(a["mpd"]), m, C.c_float(a["th"]), C.c_float(a["ratio"]), None,
                                             _p(out_mp), _p(out_sc), C.byref(nm))


def _damage_proj(a, what):
    if what == "keypoint octave 16":
        a["kp"]["octave"][5] = 16
    elif what == "keypoint octave negative":
        a["kp"]["octave"][-1] = -2
    elif what == "empty frame bounds":
        a["fb"] = (0.0, 0.0, 0.0, 480.0)
    elif what == "inverted frame bounds":
        a["fb"] = (0.0, 480.0, 752.0, 0.0)
    elif what == "NaN frame bounds":
        a["fb"] = (0.0, 0.0, float("nan"), 480.0)
    elif what == "null map descriptors":
        a["mpd"] = None
    elif what == "null map points":
        a["m"] = len(a["mps"]); a["mps"] = None
    elif what == "nlevels 0":
        a["nlevels"] = 0
    elif what == "more than 65535 keypoints":
        a["n"] = 70000
    elif what == "negative map size":
        a["m"] = -1
    else:
        raise KeyError(what)


PROJ_CASES = ["keypoint octave 16", "keypoint octave negative", "empty frame bounds", "inverted frame bounds", "NaN frame bounds",
              "null map descriptors", "null map points", "nlevels 0", "more than 65535 keypoints", "negative map size"]


def test_search_by_projection_refuses_malformed_arrays(env):
    O = env["oracle"]
    good = _proj_inputs(env)
    ref = O.search_by_projection(good["kp"], good["desc"], good["ur"], good["sf"], good["fb"], good["mps"], good["mpd"], good["th"], good["ratio"])
    assert ref[0] > 400
    n = len(good["kp"])
    for what in PROJ_CASES:
        a = _proj_inputs(env)
        _damage_proj(a, what)
        out_mp, out_sc, nm = np.full(n, SENT_I, np.int32), np.full(n, SENT_I, np.int32), C.c_int(int(SENT_I))
        rc = _proj_call(env, a, out_mp, out_sc, nm)
        assert rc == GFO_ERR_INVALID, (what, rc)
        assert (out_mp == SENT_I).all() and (out_sc == SENT_I).all() and nm.value == int(SENT_I), what
        out_mp, out_sc, nm = np.full(n, SENT_I, np.int32), np.full(n, SENT_I, np.int32), C.c_int(-7)
        assert _proj_call(env, good, out_mp, out_sc, nm) == 0, what
        assert nm.value == ref[0], what
        np.testing.assert_array_equal(out_mp, ref[1])
        np.testing.assert_array_equal(out_sc[ref[1] >= 0], ref[2][ref[1] >= 0])


def test_good_feature_entry_points_refuse_the_same_malformed_arrays(env):
    """gfo_search_by_projection_points and gfo_projection_candidates take the arrays of gfo_search_by_projection: the same damaged
    inputs, refused the same way -- nothing written, the valid call afterwards equal to the oracle."""
    O = env["oracle"]
    L, FB = env["L"], env["lib"].FrameBoundsC
    good = _proj_inputs(env)
    ref = O.search_by_projection_budget(good["kp"], good["desc"], good["ur"], good["sf"], good["fb"], good["mps"], good["mpd"], good["th"], good["ratio"])
    n, m = len(good["kp"]), len(good["mps"])

    def points(a, out_mp, out_sc, out_pt, nm):
        fb = FB(*a["fb"])
        return L.gfo_search_by_projection_points(env["ctx"], _p(a["kp"]), _p(a["desc"]), _p(a["ur"]), a.get("n", len(a["kp"])), _p(a["sf"]),
                                                 a.get("nlevels", len(a["sf"])), C.byref(fb), _p(a["mps"]), _p(a["mpd"]),
                                                 a.get("m", 0 if a["mps"] is None else len(a["mps"])), C.c_float(a["th"]), C.c_float(a["ratio"]),
                                                 None, _p(out_mp), _p(out_sc), _p(out_pt), C.byref(nm))

    def table(a, start, cand, cap, tot):
        fb = FB(*a["fb"])
        return L.gfo_projection_candidates(env["ctx"], _p(a["kp"]), _p(a["desc"]), _p(a["ur"]), a.get("n", len(a["kp"])), _p(a["sf"]),
                                           a.get("nlevels", len(a["sf"])), C.byref(fb), _p(a["mps"]), _p(a["mpd"]),
                                           a.get("m", 0 if a["mps"] is None else len(a["mps"])), C.c_float(a["th"]), _p(start), _p(cand), cap, C.byref(tot))

    cap = 64 * m
    for what in PROJ_CASES + ["null out_point", "negative capacity", "null table"]:
        a = _proj_inputs(env)
        if what in PROJ_CASES:
            _damage_proj(a, what)
        out_mp, out_sc, out_pt, nm = np.full(n, SENT_I, np.int32), np.full(n, SENT_I, np.int32), np.full(m, SENT_I, np.int32), C.c_int(int(SENT_I))
        if what not in ("negative capacity", "null table"):
            rc = points(a, out_mp, out_sc, None if what == "null out_point" else out_pt, nm)
            assert rc == GFO_ERR_INVALID, (what, rc)
            assert (out_mp == SENT_I).all() and (out_sc == SENT_I).all() and (out_pt == SENT_I).all() and nm.value == int(SENT_I), what
        start, cand, tot = np.full(m + 1, SENT_I, np.int32), np.full(cap, 0x5A5A5A5A, np.uint32), C.c_int(int(SENT_I))
        if what != "null out_point":
            rc = table(a, start, None if what == "null table" else cand, -1 if what == "negative capacity" else cap, tot)
            assert rc == GFO_ERR_INVALID, (what, rc)
            assert (start == SENT_I).all() and (cand == 0x5A5A5A5A).all() and tot.value == int(SENT_I), what
        # a refused call leaves nothing behind
        assert points(good, out_mp, out_sc, out_pt, nm) == 0, what
        assert nm.value == ref[0]
        np.testing.assert_array_equal(out_pt, ref[3]); np.testing.assert_array_equal(out_mp, ref[1])
        assert table(good, start, cand, cap, tot) == 0, what
        assert start[0] == 0 and start[m] == tot.value and 0 < tot.value <= cap


def test_projection_points_outside_the_scale_table_are_skipped_not_indexed(env):
    """a map point whose predicted level lies outside the frame's scale table (ORBmatcher.cc:177-180 indexes mvScaleFactors unchecked):
    library and oracle skip it -- whatever the integer is, nothing is indexed with it"""
    O = env["oracle"]
    a = _proj_inputs(env)
    a["mps"]["level"][::7] = [(-1, 8, 1 << 30, -(1 << 31))[i % 4] for i in range(len(a["mps"][::7]))]
    ref = O.search_by_projection(a["kp"], a["desc"], a["ur"], a["sf"], a["fb"], a["mps"], a["mpd"], a["th"], a["ratio"])
    n = len(a["kp"])
    out_mp, out_sc, nm = np.full(n, SENT_I, np.int32), np.full(n, SENT_I, np.int32), C.c_int(-7)
    assert _proj_call(env, a, out_mp, out_sc, nm) == 0
    assert nm.value == ref[0]
    np.testing.assert_array_equal(out_mp, ref[1])
    assert not np.isin(out_mp[out_mp >= 0] % 7, [0]).any() or True     # (skipped points can still be other points' indices)
    assert not np.isin(out_mp[out_mp >= 0], np.arange(0, len(a["mps"]), 7)).any()


def test_projection_queries_refuse_a_bad_mode_and_nonfinite_queries_fault_nothing(env):
    """th_dist outside 0..255 is refused; NaN / infinite query positions and radii select no cell (the window arithmetic saturates)
    and the call answers like the oracle"""
    O, G = env["oracle"], env["G"]
    kl, dl = env["kl"], env["dl"]
    n = len(kl)
    rng = np.random.default_rng(8)
    m = 600
    src = rng.integers(0, n, m)
    q = np.zeros(m, O.PROJ_QUERY_DTYPE)
    q["u"] = kl["x"][src] + rng.normal(0, 1.5, m); q["v"] = kl["y"][src] + rng.normal(0, 1.5, m); q["ur"] = q["u"] - 4
    q["radius"] = (np.float32(7.0) * env["sf"][kl["octave"][src]]).astype(np.float32)
    q["min_level"] = kl["octave"][src] - 1; q["max_level"] = kl["octave"][src] + 1
    q["angle"] = kl["angle"][src]; q["flags"] = 1 | 4
    q["u"][0] = np.nan; q["v"][1] = np.inf; q["radius"][2] = np.inf; q["radius"][3] = np.nan; q["u"][4] = -np.inf; q["radius"][5] = -3.0
    q["min_level"][6] = -(1 << 31); q["max_level"][6] = (1 << 31) - 1
    qd = dl[src].copy()
    M = G.ORBmatcher(0.9, True, extractor=env["ext"])
    for th in (-1, 256, 100000):
        with pytest.raises(G.GfoError) as e:
            M.SearchByProjectionQueries(kl, dl, None, kl["angle"], (0.0, 0.0, 752.0, 480.0), q, qd, th_dist=th)
        assert e.value.code == GFO_ERR_INVALID
    # the rotation histogram is indexed with the angle difference (ORBmatcher.cc:1557-1565 asserts the bin): angles outside 0..360
    for where, idx, val in (("kp", 9, -1.0), ("kp", 0, np.nan), ("q", 50, 1e9), ("q", 7, np.inf), ("kp", 3, 360.5)):
        ang, q2 = kl["angle"].copy(), q.copy()
        if where == "kp":
            ang[idx] = val
        else:
            q2["angle"][idx] = val
        with pytest.raises(G.GfoError) as e:
            M.SearchByProjectionQueries(kl, dl, None, ang, (0.0, 0.0, 752.0, 480.0), q2, qd)
        assert e.value.code == GFO_ERR_INVALID and "angle" in str(e.value)
    ref = O.search_by_projection_queries(kl, dl, None, kl["angle"], (0.0, 0.0, 752.0, 480.0), q, qd, False, 0.9, 100, True)
    got = M.SearchByProjectionQueries(kl, dl, None, kl["angle"], (0.0, 0.0, 752.0, 480.0), q, qd)
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1])


# ------------------------------------------------------------------------------------------------------------------------------
# gfo_vocabulary_upload / gfo_compute_bow / gfo_bow_transform
# ------------------------------------------------------------------------------------------------------------------------------
def test_vocabulary_upload_refuses_a_tree_that_is_not_one(env):
    """children must follow their parent as one contiguous range inside the node array: anything else would let the descent walk
    out of the arrays (or around in circles); a refused upload keeps the vocabulary that was resident"""
    O, G = env["oracle"], env["G"]
    tree = O.make_vocabulary(6, 3, seed=4)
    voc = G.ORBVocabulary(tree, env["ext"])
    dl = env["dl"][:700]
    ref = O.compute_bow(tree, dl, 2, 0, 1)
    n = len(tree["first_child"])

    def damaged(what):
        t = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in tree.items()}
        if what == "child range past the end":
            t["n_children"][n - 1] = 3; t["first_child"][n - 1] = n - 1
        elif what == "first_child past the end":
            t["first_child"][0] = n + 5
        elif what == "child before its parent (a cycle)":
            i = int(np.nonzero(t["n_children"] > 0)[0][-1])
            t["first_child"][i] = 0
        elif what == "negative child count":
            t["n_children"][1] = -2
        elif what == "child count beyond 65535":
            t["n_children"][0] = 70000
        elif what == "no nodes":
            t = {k: (v[:0] if isinstance(v, np.ndarray) else v) for k, v in t.items()}
        return t

    for what in ("child range past the end", "first_child past the end", "child before its parent (a cycle)", "negative child count",
                 "child count beyond 65535", "no nodes"):
        with pytest.raises(G.GfoError) as e:
            G.ORBVocabulary(damaged(what), env["ext"])
        assert e.value.code == GFO_ERR_INVALID, what
        (bw, bv), (fn, fs, fi) = voc.compute_bow(dl, 2, "TF_IDF", "L1")          # the resident vocabulary still answers
        np.testing.assert_array_equal(bw, ref[0]); assert bv.tobytes() == ref[1].tobytes(), what
        np.testing.assert_array_equal(fn, ref[2]); np.testing.assert_array_equal(fs, ref[3]); np.testing.assert_array_equal(fi, ref[4])


def test_compute_bow_refuses_bad_modes_and_leaves_outputs_alone(env):
    O, G, L = env["oracle"], env["G"], env["L"]
    tree = O.make_vocabulary(5, 3, seed=2)
    voc = G.ORBVocabulary(tree, env["ext"])
    dl = env["dl"][:300]
    n = len(dl)
    BM = env["lib"].BowModeC
    for weighting, norm, nn in ((4, 1, n), (-1, 0, n), (0, 3, n), (0, -1, n), (0, 1, -4)):
        bw = np.full(n, 0xAAAAAAAA, np.uint32); bv = np.full(n, 7.25, np.float64); fn = np.full(n, 0xAAAAAAAA, np.uint32)
        fs = np.full(n + 1, SENT_I, np.int32); fi = np.full(n, 0xAAAAAAAA, np.uint32)
        nw, nf = C.c_int(int(SENT_I)), C.c_int(int(SENT_I))
        mode = BM(weighting, norm)
        rc = L.gfo_compute_bow(env["ctx"], _p(dl), nn, 2, C.byref(mode), _p(bw), _p(bv), C.byref(nw), _p(fn), _p(fs), _p(fi), C.byref(nf))
        assert rc == GFO_ERR_INVALID, (weighting, norm, nn)
        assert (bw == 0xAAAAAAAA).all() and (bv == 7.25).all() and (fn == 0xAAAAAAAA).all() and (fs == SENT_I).all() and (fi == 0xAAAAAAAA).all()
        assert nw.value == int(SENT_I) and nf.value == int(SENT_I)
    ref = O.compute_bow(tree, dl, 2, 0, 1)
    (bw, bv), (fn, fs, fi) = voc.compute_bow(dl, 2, "TF_IDF", "L1")
    np.testing.assert_array_equal(bw, ref[0]); assert bv.tobytes() == ref[1].tobytes()
    np.testing.assert_array_equal(fi, ref[4])


# ------------------------------------------------------------------------------------------------------------------------------
# gfo_search_for_triangulation, gfo_search_for_initialization
# ------------------------------------------------------------------------------------------------------------------------------
def _tri_inputs(env):
    import gf_cases
    O = env["oracle"]
    c = gf_cases.triangulation_case(O, env["kl"], env["dl"], np.random.default_rng(3))
    sf = env["sf"]
    return {"kp1": c["kp1"].copy(), "d1": c["desc1"], "h1": c["has1"], "u1": c["ur1"], "fv1": [a.copy() for a in c["fv1"]], "kp2": c["kp2"].copy(),
            "d2": c["desc2"], "h2": c["has2"], "u2": c["ur2"], "fv2": [a.copy() for a in c["fv2"]], "sf": sf, "sg": (sf * sf).astype(np.float32),
            "f12": c["f12"].reshape(9).copy(), "ex": float(c["ex"]), "ey": float(c["ey"]), "nlevels": len(sf)}


def _tri_call(env, a, out, nm):
    FV = env["lib"].FeatureVectorC
    keep = []

    def fv(t):
        ids, start, items = (None if x is None else np.ascontiguousarray(x, dt) for x, dt in zip(t[:3], (np.uint32, np.int32, np.uint32)))
        keep.append((ids, start, items))
        return FV(None if ids is None else ids.ctypes.data, None if start is None else start.ctypes.data, None if items is None else items.ctypes.data,
                  t[3] if len(t) > 3 else len(ids))
    f1, f2 = fv(a["fv1"]), fv(a["fv2"])
    return env["L"].gfo_search_for_triangulation(env["ctx"], _p(a["kp1"]), _p(a["d1"]), _p(a["h1"]), _p(a["u1"]), a.get("n1", len(a["kp1"])), C.byref(f1),
                                                 _p(a["kp2"]), _p(a["d2"]), _p(a["h2"]), _p(a["u2"]), a.get("n2", len(a["kp2"])), C.byref(f2), _p(a["sf"]),
                                                 _p(a["sg"]), a["nlevels"], _p(a["f12"]), C.c_float(a["ex"]), C.c_float(a["ey"]), 0, 1, _p(out), C.byref(nm))


def _damage_tri(a, what):
    if what == "second item == n2":
        a["fv2"][2][5] = len(a["kp2"])
    elif what == "first item huge":
        a["fv1"][2][0] = 0xFFFFFFF0
    elif what == "node_start decreases":
        a["fv1"][1][3] = a["fv1"][1][2] - 1
    elif what == "node_ids not ascending":
        a["fv2"][0][1], a["fv2"][0][2] = a["fv2"][0][2], a["fv2"][0][1]
    elif what == "second octave == nlevels":
        a["kp2"]["octave"][9] = a["nlevels"]
    elif what == "second octave negative":
        a["kp2"]["octave"][0] = -2
    elif what == "nlevels beyond the table":
        a["nlevels"] = 64
    elif what == "null fundamental matrix":
        a["f12"] = None
    elif what == "null map-point flags":
        a["h2"] = None
    elif what == "null sigma table":
        a["sg"] = None
    elif what == "negative count":
        a["n1"] = -1
    elif what == "first angle NaN":
        a["kp1"]["angle"][4] = np.nan
    else:
        raise KeyError(what)


TRI_CASES = ["second item == n2", "first item huge", "node_start decreases", "node_ids not ascending", "second octave == nlevels", "second octave negative",
             "nlevels beyond the table", "null fundamental matrix", "null map-point flags", "null sigma table", "negative count", "first angle NaN"]


def test_search_for_triangulation_refuses_malformed_arrays(env):
    O = env["oracle"]
    good = _tri_inputs(env)
    ref = O.search_for_triangulation(good["kp1"], good["d1"], good["h1"], good["u1"], tuple(good["fv1"]), good["kp2"], good["d2"], good["h2"], good["u2"],
                                     tuple(good["fv2"]), good["sf"], good["sg"], good["f12"], good["ex"], good["ey"], False, True)
    assert ref[0] > 300
    for what in TRI_CASES:
        a = _tri_inputs(env)
        _damage_tri(a, what)
        out = np.full(len(a["kp1"]), SENT_I, np.int32)
        nm = C.c_int(int(SENT_I))
        rc = _tri_call(env, a, out, nm)
        assert rc == GFO_ERR_INVALID, (what, rc)
        assert (out == SENT_I).all() and nm.value == int(SENT_I), f"{what}: outputs were written by a refused call"
        assert "gfo_search_for_triangulation" in env["L"].gfo_last_error(env["ctx"]).decode(), what
        out = np.full(len(good["kp1"]), SENT_I, np.int32)
        nm = C.c_int(-7)
        assert _tri_call(env, good, out, nm) == 0, what
        assert nm.value == ref[0], what
        np.testing.assert_array_equal(out, ref[1])


def _init_inputs(env):
    import gf_cases
    kp2, d2, prev = gf_cases.initialization_case(env["oracle"], env["kl"], env["dl"], np.random.default_rng(5))
    return {"kp1": env["kl"].copy(), "d1": env["dl"], "prev": prev, "kp2": kp2, "d2": d2, "fb": (0.0, 0.0, 752.0, 480.0), "win": 100}


def _init_call(env, a, out, nm):
    fb = None if a["fb"] is None else env["lib"].FrameBoundsC(*a["fb"])
    return env["L"].gfo_search_for_initialization(env["ctx"], _p(a["kp1"]), _p(a["d1"]), a.get("n1", len(a["kp1"])), _p(a["prev"]), _p(a["kp2"]), _p(a["d2"]),
                                                  a.get("n2", len(a["kp2"])), None if fb is None else C.byref(fb), a["win"], C.c_float(0.9), 1, _p(out),
                                                  C.byref(nm))


def _damage_init(a, what):
    if what == "second octave 16":
        a["kp2"]["octave"][3] = 16
    elif what == "second octave negative":
        a["kp2"]["octave"][-1] = -1
    elif what == "empty frame bounds":
        a["fb"] = (0.0, 0.0, 0.0, 480.0)
    elif what == "NaN frame bounds":
        a["fb"] = (0.0, 0.0, float("nan"), 480.0)
    elif what == "null frame bounds":
        a["fb"] = None
    elif what == "negative window":
        a["win"] = -3
    elif what == "null vbPrevMatched":
        a["prev_keep"], a["prev"] = a["prev"], None
    elif what == "null second descriptors":
        a["d2"] = None
    elif what == "negative count":
        a["n2"] = -4
    elif what == "more than 65535 keypoints in F2":
        k = np.zeros(70000, a["kp2"].dtype)
        k["x"] = 100; k["y"] = 100
        a["kp2"], a["d2"] = k, np.zeros((70000, 32), np.uint8)
    else:
        raise KeyError(what)


INIT_CASES = ["second octave 16", "second octave negative", "empty frame bounds", "NaN frame bounds", "null frame bounds", "negative window",
              "null vbPrevMatched", "null second descriptors", "negative count", "more than 65535 keypoints in F2"]


def test_search_for_initialization_refuses_malformed_arrays(env):
    O = env["oracle"]
    good = _init_inputs(env)
    p_ref = good["prev"].copy()
    ref = O.search_for_initialization(good["kp1"], good["d1"], p_ref, good["kp2"], good["d2"], good["fb"], 100, 0.9, True)
    assert ref[0] > 100
    for what in INIT_CASES:
        a = _init_inputs(env)
        _damage_init(a, what)
        before = None if a["prev"] is None else a["prev"].copy()
        out = np.full(len(a["kp1"]), SENT_I, np.int32)
        nm = C.c_int(int(SENT_I))
        rc = _init_call(env, a, out, nm)
        assert rc == GFO_ERR_INVALID, (what, rc)
        assert (out == SENT_I).all() and nm.value == int(SENT_I), f"{what}: outputs were written by a refused call"
        assert before is None or a["prev"].tobytes() == before.tobytes(), f"{what}: vbPrevMatched was written by a refused call"
        g = _init_inputs(env)
        out = np.full(len(g["kp1"]), SENT_I, np.int32)
        nm = C.c_int(-7)
        assert _init_call(env, g, out, nm) == 0, what
        assert nm.value == ref[0], what
        np.testing.assert_array_equal(out, ref[1])
        assert g["prev"].tobytes() == p_ref.tobytes(), what


# ------------------------------------------------------------------------------------------------------------------------------
# gfo_search_by_bow_keyframes (the damaged CSRs of the table above on either side), gfo_search_for_fusion,
# gfo_search_by_projection_queries_points
# ------------------------------------------------------------------------------------------------------------------------------
def _bow_kf_call(env, a, out, nm):
    FV = env["lib"].FeatureVectorC
    keep = []

    def fv(t):
        ids, start, items = (None if x is None else np.ascontiguousarray(x, dt) for x, dt in zip(t[:3], (np.uint32, np.int32, np.uint32)))
        keep.append((ids, start, items))
        return FV(None if ids is None else ids.ctypes.data, None if start is None else start.ctypes.data, None if items is None else items.ctypes.data,
                  t[3] if len(t) > 3 else len(ids))
    f1, f2 = fv(a["kfv"]), fv(a["ffv"])
    return env["L"].gfo_search_by_bow_keyframes(env["ctx"], _p(a["kd"]), _p(a["ka"]), _p(a["valid"]), len(a["kd"]), C.byref(f1), _p(a["fd"]), _p(a["fa"]),
                                                _p(a.get("valid2", a["valid"])), len(a["fd"]), C.byref(f2), C.c_float(0.75), 1, _p(out), C.byref(nm))


def test_search_by_bow_between_keyframes_refuses_the_same_damage(env):
    O = env["oracle"]
    good = _bow_inputs(env)
    ref = O.search_by_bow_keyframes(good["kd"], good["ka"], good["valid"], tuple(good["kfv"]), good["fd"], good["fa"], good["valid"], tuple(good["ffv"]), 0.75,
                                    True)
    assert ref[0] > 300
    for what in BOW_CASES + ["null second mask"]:
        a = _bow_inputs(env)
        if what == "null second mask":
            a["valid2"] = None
        else:
            _damage_bow(a, what)
        out = np.full(len(a["kd"]), SENT_I, np.int32)
        nm = C.c_int(int(SENT_I))
        rc = _bow_kf_call(env, a, out, nm)
        assert rc == GFO_ERR_INVALID, (what, rc)
        assert (out == SENT_I).all() and nm.value == int(SENT_I), f"{what}: outputs were written by a refused call"
        out = np.full(len(good["kd"]), SENT_I, np.int32)
        nm = C.c_int(-7)
        assert _bow_kf_call(env, good, out, nm) == 0, what
        assert nm.value == ref[0], what
        np.testing.assert_array_equal(out, ref[1])


def _fuse_inputs(env):
    G = env["G"]
    kl, dl = env["kl"], env["dl"]
    rng = np.random.default_rng(17)
    m = 1500
    src = rng.integers(0, len(kl), m)
    q = np.zeros(m, G.PROJ_QUERY_DTYPE)
    q["u"] = kl["x"][src] + rng.normal(0, 1.5, m); q["v"] = kl["y"][src] + rng.normal(0, 1.5, m)
    q["ur"] = -1
    q["radius"] = 3.0 * env["sf"][kl["octave"][src]]
    q["min_level"] = kl["octave"][src] - 1; q["max_level"] = kl["octave"][src]
    q["flags"] = 1
    qd = dl[src].copy()
    bits = rng.integers(0, 256, m)
    qd[np.arange(m), bits >> 3] ^= (1 << (bits & 7)).astype(np.uint8)
    sf = env["sf"]
    return {"kp": kl.copy(), "desc": dl, "fb": (0.0, 0.0, 752.0, 480.0), "inv": (1.0 / (sf * sf)).astype(np.float32), "nlevels": len(sf), "q": q, "qd": qd,
            "th": 50}


def _fuse_call(env, a, out):
    fb = None if a["fb"] is None else env["lib"].FrameBoundsC(*a["fb"])
    return env["L"].gfo_search_for_fusion(env["ctx"], _p(a["kp"]), _p(a["desc"]), None, a["n"] if "n" in a else len(a["kp"]), None if fb is None else C.byref(fb),
                                          _p(a["inv"]), a["nlevels"], _p(a["q"]), _p(a["qd"]), a.get("m", len(a["q"])), a["th"], _p(out))


FUSE_CASES = {"keypoint octave == nlevels": lambda a: a["kp"]["octave"].__setitem__(5, a["nlevels"]),
              "keypoint octave negative": lambda a: a["kp"]["octave"].__setitem__(0, -1),
              "null sigma table": lambda a: a.__setitem__("inv", None),
              "nlevels zero": lambda a: a.__setitem__("nlevels", 0),
              "nlevels beyond the table": lambda a: a.__setitem__("nlevels", 17),
              "empty frame bounds": lambda a: a.__setitem__("fb", (10.0, 0.0, 10.0, 480.0)),
              "null frame bounds": lambda a: a.__setitem__("fb", None),
              "null query descriptors": lambda a: a.__setitem__("qd", None),
              "null keypoints": lambda a: (a.__setitem__("n", len(a["kp"])), a.__setitem__("kp", None)),
              "threshold 256": lambda a: a.__setitem__("th", 256),
              "negative keypoint count": lambda a: a.__setitem__("n", -1),
              "negative point count": lambda a: a.__setitem__("m", -2)}


def test_search_for_fusion_refuses_malformed_arrays(env):
    O = env["oracle"]
    good = _fuse_inputs(env)
    ref = O.search_for_fusion(good["kp"], good["desc"], None, good["fb"], good["inv"], good["q"], good["qd"], good["th"])
    assert (ref >= 0).sum() > 800
    for what, damage in FUSE_CASES.items():
        a = _fuse_inputs(env)
        damage(a)
        out = np.full(len(good["q"]), SENT_I, np.int32)
        rc = _fuse_call(env, a, out)
        assert rc == GFO_ERR_INVALID, (what, rc)
        assert (out == SENT_I).all(), f"{what}: outputs were written by a refused call"
        out = np.full(len(good["q"]), SENT_I, np.int32)
        assert _fuse_call(env, good, out) == 0, what
        np.testing.assert_array_equal(out, ref, err_msg=what)
