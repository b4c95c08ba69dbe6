"""GPU parity tests of ORBmatcher::SearchByProjection(Frame&, MapPoints, th) through the C ABI vs
the CPU oracle (serial, order-dependent reference semantics).  Indices bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["wavefront_per_point", "wavefront_per_point_small_pool", "thread_per_point"], autouse=True)
def round0_form(request, monkeypatch):
    """Round 0 has two forms: k_proj_round0_wave when a call brings at most 16 384 points (the per-frame calls of Tracking), the
    thread-per-point kernels otherwise.  Every case of this module runs under both (GFO_PROJ_WAVE is read on every call), and a third
    time with the wavefront form's pool of full candidate lists cut to 3000 keys, so that some contended points find their list there
    and the others fall back to the grid rescan inside one resolve."""
    monkeypatch.setenv("GFO_PROJ_WAVE", "0" if request.param == "thread_per_point" else "1")
    if request.param == "wavefront_per_point_small_pool":
        monkeypatch.setenv("GFO_PROJ_SPILL_CAP", "3000")
    return request.param


@pytest.fixture(scope="module")
def ext():
    import gf_orb_slam2_amd as G
    e = G.ORBextractor(2000, 1.2, 8, 20, 7)
    yield e
    e.close()


def _frame(oracle):
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    st = np.load(os.path.join(GOLDEN, "EuRoC_stereo.npz"))
    return kl, dl, st["u_right"]


def test_golden_projection_case(ext, oracle):
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    g = np.load(os.path.join(GOLDEN, "EuRoC_projection.npz"))
    m = G.ORBmatcher(0.8, True, extractor=ext)
    sf = ext.GetScaleFactors()
    nm, out_mp, out_sc = m.SearchByProjection(kl, dl, u, sf, (0.0, 0.0, 752.0, 480.0), g["mps"], g["mp_desc"], 3.0, g["taken"])
    assert nm == int(g["nmatches"])
    np.testing.assert_array_equal(out_mp, g["out_mp"])
    np.testing.assert_array_equal(out_sc, g["out_score"])


@pytest.mark.parametrize("seed,m,th,ratio", [(1, 3000, 3.0, 0.8), (2, 8000, 1.0, 0.6), (3, 20000, 5.0, 0.9)])
def test_contended_maps(ext, oracle, seed, m, th, ratio):
    """Many map points compete for the same keypoints (near-duplicate descriptors, clustered
    projections), with blocking and non-blocking points mixed: exercises the ordered resolution."""
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    rng = np.random.default_rng(seed)
    n = len(kl)
    mps = np.zeros(m, oracle.MAP_POINT_DTYPE)
    src = rng.integers(0, n, m)                       # every map point imitates some keypoint
    mpd = dl[src].copy()
    flips = rng.integers(0, 256, (m, 6))
    for j in range(6):                                # up to 6 flipped bits: lots of ties and near ties
        sel = rng.random(m) < 0.5
        mpd[sel, flips[sel, j] >> 3] ^= (1 << (flips[sel, j] & 7)).astype(np.uint8)
    mps["proj_x"] = kl["x"][src] + rng.normal(0, 3, m)
    mps["proj_y"] = kl["y"][src] + rng.normal(0, 3, m)
    mps["proj_xr"] = mps["proj_x"] - rng.uniform(0, 30, m)
    mps["level"] = np.clip(kl["octave"][src] + rng.integers(-1, 2, m), 0, 7)
    mps["view_cos"] = rng.choice([1.0, 0.9985, 0.99], m)
    fl = np.full(m, 1 | 4, np.int32)
    fl[rng.random(m) < 0.05] = 4
    fl[rng.random(m) < 0.05] |= 2
    fl[rng.random(m) < 0.3] &= ~4
    mps["flags"] = fl
    taken = (rng.random(n) < 0.2).astype(np.uint8)
    bounds = (0.0, 0.0, 752.0, 480.0)
    sf = ext.GetScaleFactors()
    ref = oracle.search_by_projection(kl, dl, u, sf, bounds, mps, mpd, th, ratio, taken)
    got = G.ORBmatcher(ratio, True, extractor=ext).SearchByProjection(kl, dl, u, sf, bounds, mps, mpd, th, taken)
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    assert ref[0] > 100


@pytest.mark.parametrize("seed,m,th", [(1, 3000, 3.0), (3, 10000, 3.0), (4, 2000, 5.0), (5, 16000, 1.0)])
def test_every_point_blocking_with_wide_windows(ext, oracle, seed, m, th):
    """The scenario of tools/matcher_call_latency.py: every map point has observations (so every accepted match blocks), imitates a
    random keypoint and projects within a few pixels of it, several points per keypoint, th = 3 / 5 windows.  Hundreds to thousands of
    points find their seven cached candidates all claimed by lower points (690 of 3000 at th 3) and go to their full candidate list --
    the spill list k_proj_round0_wave leaves, or the grid rescan when there is none (the other two forms of this module's fixture)."""
    import gf_orb_slam2_amd as G
    kl, dl, _ = _frame(oracle)
    rng = np.random.default_rng(seed)
    n = len(kl)
    mps = np.zeros(m, oracle.MAP_POINT_DTYPE)
    src = rng.integers(0, n, m)
    mpd = dl[src].copy()
    fl = rng.integers(0, 256, (m, 8))
    for j in range(8):
        mpd[np.arange(m), fl[:, j] >> 3] ^= (1 << (fl[:, j] & 7)).astype(np.uint8)
    mps["proj_x"] = kl["x"][src] + rng.normal(0, 2, m)
    mps["proj_y"] = kl["y"][src] + rng.normal(0, 2, m)
    mps["proj_xr"] = mps["proj_x"] - 5
    mps["level"] = kl["octave"][src]
    mps["view_cos"] = 1.0
    mps["flags"] = 1 | 4
    u = np.full(n, -1, np.float32)
    bounds = (0.0, 0.0, 752.0, 480.0)
    sf = ext.GetScaleFactors()
    ref = oracle.search_by_projection(kl, dl, u, sf, bounds, mps, mpd, th, 0.8)
    got = G.ORBmatcher(0.8, True, extractor=ext).SearchByProjection(kl, dl, u, sf, bounds, mps, mpd, th, None)
    assert got[0] == ref[0] and ref[0] > 1000
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])


def test_projection_edge_cases(ext, oracle):
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    sf = ext.GetScaleFactors()
    b = (0.0, 0.0, 752.0, 480.0)
    z = np.zeros(0, oracle.MAP_POINT_DTYPE)
    nm, out_mp, _ = m.SearchByProjection(kl, dl, None, sf, b, z, np.zeros((0, 32), np.uint8), 3.0, None)
    assert nm == 0 and (out_mp == -1).all()
    # projections outside the image / all filtered / no u_right / no taken
    mps = np.zeros(50, oracle.MAP_POINT_DTYPE)
    mps["proj_x"] = np.linspace(-500, 1500, 50); mps["proj_y"] = np.linspace(-300, 900, 50)
    mps["level"] = 3; mps["view_cos"] = 1.0; mps["flags"] = 5
    mpd = np.tile(dl[:1], (50, 1))
    ref = oracle.search_by_projection(kl, dl, None, sf, b, mps, mpd, 3.0, 0.8, None)
    got = m.SearchByProjection(kl, dl, None, sf, b, mps, mpd, 3.0, None)
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1]); np.testing.assert_array_equal(got[2], ref[2])


def _frame_queries(oracle, kl, dl, seed, th, forward=False, backward=False):
    """Flattening an adapter does for SearchByProjection(CurrentFrame, LastFrame, th, bMono): the last
    frame's keypoints with a map point, projected into the current frame (ORBmatcher.cc:1465-1510)."""
    rng = np.random.default_rng(seed)
    n = len(kl)
    m = n
    q = np.zeros(m, oracle.PROJ_QUERY_DTYPE)
    sf = oracle.OracleExtractor().scale_factors
    q["u"] = kl["x"] + rng.normal(0, 4, m); q["v"] = kl["y"] + rng.normal(0, 4, m)
    q["ur"] = q["u"] - rng.uniform(0, 40, m).astype(np.float32)
    octv = kl["octave"]
    q["radius"] = (np.float32(th) * sf[octv]).astype(np.float32)
    if forward:
        q["min_level"] = octv; q["max_level"] = -1
    elif backward:
        q["min_level"] = 0; q["max_level"] = octv
    else:
        q["min_level"] = octv - 1; q["max_level"] = octv + 1
    q["angle"] = ((kl["angle"] + rng.normal(0, 25, m)) % 360).astype(np.float32)
    fl = np.full(m, 1 | 4, np.int32)
    fl[rng.random(m) < 0.25] = 0          # no map point / outlier / behind the camera / outside the image
    fl[rng.random(m) < 0.2] &= ~4
    q["flags"] = fl
    qd = dl.copy()
    for _ in range(12):
        sel = rng.random(m) < 0.5
        bits = rng.integers(0, 256, m)
        qd[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    perm = rng.permutation(m)             # the last frame's keypoint order is unrelated to the current one's
    return q[perm], qd[perm]


@pytest.mark.parametrize("seed,th,fwd,bwd,ori", [(1, 7.0, False, False, True), (2, 15.0, True, False, True),
                                                 (3, 15.0, False, True, False), (4, 3.0, False, False, True)])
def test_frame_to_frame_projection(ext, oracle, seed, th, fwd, bwd, ori):
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    q, qd = _frame_queries(oracle, kl, dl, seed, th, fwd, bwd)
    taken = (np.random.default_rng(seed).random(len(kl)) < 0.1).astype(np.uint8)
    b = (0.0, 0.0, 752.0, 480.0)
    ref = oracle.search_by_projection_queries(kl, dl, u, kl["angle"], b, q, qd, False, 0.9, 100, ori, taken)
    got = G.ORBmatcher(0.9, ori, extractor=ext).SearchByProjectionQueries(kl, dl, u, kl["angle"], b, q, qd, kp_taken=taken)
    assert got[0] == ref[0] and ref[0] > 300
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2][got[1] >= 0], ref[2][ref[1] >= 0])


@pytest.mark.parametrize("seed,th,fwd,ori,K", [(1, 7.0, False, True, 150), (2, 15.0, True, True, 1), (3, 15.0, False, False, 400), (4, 3.0, False, True, 37),
                                               (5, 7.0, False, True, 100000)])
def test_frame_to_frame_projection_with_a_feature_budget(ext, oracle, seed, th, fwd, ori, K):
    """gfo_proj_mode::max_matches = SearchByProjection(Cur, Last) compiled with BUDGETING_FEATURE_MATCHING (ORBmatcher.h:36-37): the loop over
    the last frame's points ends with the K-th match, which never enters the rotation histogram (ORBmatcher.cc:1547-1552); queries in
    the shuffled order of the last frame's keypoints, some of their map points unobserved (later queries may take their keypoint)"""
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    q, qd = _frame_queries(oracle, kl, dl, seed, th, fwd, False)
    taken = (np.random.default_rng(seed).random(len(kl)) < 0.1).astype(np.uint8)
    b = (0.0, 0.0, 752.0, 480.0)
    with oracle.feature_budget(K):
        ref = oracle.search_by_projection_queries(kl, dl, u, kl["angle"], b, q, qd, False, 0.9, 100, ori, taken)
    got = G.ORBmatcher(0.9, ori, extractor=ext).SearchByProjectionQueries(kl, dl, u, kl["angle"], b, q, qd, kp_taken=taken, max_matches=K)
    assert got[0] == ref[0] and 0 < ref[0] <= K
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2][got[1] >= 0], ref[2][ref[1] >= 0])
    full = oracle.search_by_projection_queries(kl, dl, u, kl["angle"], b, q, qd, False, 0.9, 100, ori, taken)
    if K < 1000:
        assert full[0] > ref[0] or not ori


@pytest.mark.parametrize("seed,th,orb_dist,ori", [(1, 10.0, 100, True), (2, 3.0, 64, True), (3, 10.0, 100, False)])
def test_keyframe_projection_overload(ext, oracle, seed, th, orb_dist, ori):
    """SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1595-1721): the
    adapter's flattening -- every query blocks the slot it takes, a slot with ANY map point is taken on entry, no
    mvuRight gate, level window [l-1, l+1], threshold ORBdist -- through the query form, against the oracle's literal
    statement of that overload."""
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    q, qd = _frame_queries(oracle, kl, dl, 100 + seed, th)
    active = (q["flags"] & 1) != 0
    q["flags"] = np.where(active, 1 | 4, 0)          # adapter: flags = 1 | 4 for every map point that reaches the search
    q["ur"] = 0
    kp_set = (np.random.default_rng(seed).random(len(kl)) < 0.3).astype(np.uint8)   # mvpMapPoints[i] != NULL on entry
    b = (0.0, 0.0, 752.0, 480.0)
    ref = oracle.search_by_projection_kf(kl, dl, kl["angle"], b, q, qd, orb_dist, ori, kp_set)
    got = G.ORBmatcher(0.9, ori, extractor=ext).SearchByProjectionQueries(kl, dl, None, kl["angle"], b, q, qd, use_ratio=False,
                                                                           th_dist=orb_dist, kp_taken=kp_set)
    assert got[0] == ref[0] and ref[0] > 200
    np.testing.assert_array_equal(got[1], ref[1])
    assert (got[1] == -2).any() == bool(ori)          # the rotation check really cleared something when it ran
    np.testing.assert_array_equal(got[2][got[1] >= 0], ref[2][ref[1] >= 0])


def test_query_form_reproduces_map_point_overload(oracle):
    """the two oracle statements (literal map-point overload vs query form) agree: pins the flattening"""
    kl, dl, u = _frame(oracle)
    g = np.load(os.path.join(GOLDEN, "EuRoC_projection.npz"))
    mps = g["mps"]
    sf = oracle.OracleExtractor().scale_factors
    q = np.zeros(len(mps), oracle.PROJ_QUERY_DTYPE)
    r = np.where(mps["view_cos"].astype(np.float64) > 0.998, np.float32(2.5), np.float32(4.0)) * np.float32(3.0)
    q["u"], q["v"], q["ur"] = mps["proj_x"], mps["proj_y"], mps["proj_xr"]
    q["radius"] = (r.astype(np.float32) * sf[mps["level"]]).astype(np.float32)
    q["min_level"] = mps["level"] - 1; q["max_level"] = mps["level"]
    q["flags"] = (((mps["flags"] & 1) != 0) & ((mps["flags"] & 2) == 0)).astype(np.int32) | (mps["flags"] & 4)
    got = oracle.search_by_projection_queries(kl, dl, u, None, (0.0, 0.0, 752.0, 480.0), q, g["mp_desc"], True, 0.8, 100, False, g["taken"])
    assert got[0] == int(g["nmatches"])
    np.testing.assert_array_equal(got[1], g["out_mp"])


def test_more_keypoints_than_fit_the_lds_tables(ext, oracle):
    """20 000 keypoints: the claim / owner tables of the resolve kernel (8 B per keypoint) and the grid copy of round 0
    no longer fit LDS, so both run their HBM variants (tables read past the L1, grid gathered in place) -- same result as
    the serial oracle, ratio test, taken keypoints, mvuRight gate and the rotation check included"""
    import gf_orb_slam2_amd as G
    rng = np.random.default_rng(9)
    n, m = 20000, 30000
    kp = np.zeros(n, oracle.KEYPOINT_DTYPE)
    kp["x"] = rng.uniform(20, 1900, n).astype(np.float32); kp["y"] = rng.uniform(20, 1060, n).astype(np.float32)
    kp["octave"] = rng.integers(0, 8, n); kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    u = np.where(rng.random(n) < 0.5, kp["x"] - rng.uniform(1, 40, n), -1).astype(np.float32)
    sf = ext.GetScaleFactors()
    src = rng.integers(0, n, m)
    q = np.zeros(m, oracle.PROJ_QUERY_DTYPE)
    q["u"] = kp["x"][src] + rng.normal(0, 3, m); q["v"] = kp["y"][src] + rng.normal(0, 3, m)
    q["ur"] = q["u"] - rng.uniform(0, 40, m).astype(np.float32)
    q["radius"] = (np.float32(7.0) * sf[kp["octave"][src]]).astype(np.float32)
    q["min_level"] = kp["octave"][src] - 1; q["max_level"] = kp["octave"][src] + 1
    q["angle"] = ((kp["angle"][src] + rng.normal(0, 20, m)) % 360).astype(np.float32)
    fl = np.full(m, 5, np.int32); fl[rng.random(m) < 0.2] = 0; fl[rng.random(m) < 0.3] &= ~4
    q["flags"] = fl
    qd = desc[src].copy()
    for _ in range(10):
        sel = rng.random(m) < 0.5
        bits = rng.integers(0, 256, m)
        qd[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    taken = (rng.random(n) < 0.1).astype(np.uint8)
    b = (0.0, 0.0, 1920.0, 1080.0)
    for use_ratio, ori in ((True, False), (False, True)):
        ref = oracle.search_by_projection_queries(kp, desc, u, kp["angle"], b, q, qd, use_ratio, 0.8, 100, ori, taken)
        got = G.ORBmatcher(0.8, ori, extractor=ext).SearchByProjectionQueries(kp, desc, u, kp["angle"], b, q, qd, use_ratio=use_ratio,
                                                                              kp_taken=taken)
        assert got[0] == ref[0] and ref[0] > 5000
        np.testing.assert_array_equal(got[1], ref[1])
        np.testing.assert_array_equal(got[2][got[1] >= 0], ref[2][ref[1] >= 0])


def test_candidate_at_distance_256_is_no_candidate(ext, oracle):
    """ADVICE r2 asked about k_project.hip's key_entry: a candidate whose descriptor differs in all 256 bits packs as
    dist << 23 with bit 31 set and reads as "none".  That IS the reference: bestDist / bestDist2 start at 256 and are
    replaced on strict `<` only (ORBmatcher.cc:186-224), so such a candidate is never second-best and the ratio test
    (:228-231) does not see it.  Two keypoints of one level inside the window, the query 200 bits from the first;
    threshold 255, ratio 0.6: with the second 256 bits away the match is ACCEPTED (no second candidate), with the second
    255 bits away it is rejected (200 > 0.6 * 255)."""
    import gf_orb_slam2_amd as G
    kp = np.zeros(2, oracle.KEYPOINT_DTYPE)
    kp["x"] = [100.0, 103.0]; kp["y"] = [100.0, 101.0]; kp["octave"] = 0; kp["size"] = 31.0; kp["angle"] = 0.0
    qd = np.zeros((1, 32), np.uint8)
    d0 = np.zeros(32, np.uint8); d0[:25] = 0xFF                  # 200 bits from the query
    for second_bits, accepted in ((256, True), (255, False)):
        d1 = np.full(32, 0xFF, np.uint8)
        if second_bits == 255:
            d1[31] = 0x7F
        desc = np.stack([d0, d1])
        q = np.zeros(1, oracle.PROJ_QUERY_DTYPE)
        q["u"] = 101.0; q["v"] = 100.0; q["radius"] = 10.0; q["min_level"] = -1; q["max_level"] = -1; q["flags"] = 1 | 4
        b = (0.0, 0.0, 752.0, 480.0)
        ref = oracle.search_by_projection_queries(kp, desc, None, kp["angle"], b, q, qd, True, 0.6, 255, False, None)
        got = G.ORBmatcher(0.6, False, extractor=ext).SearchByProjectionQueries(kp, desc, None, kp["angle"], b, q, qd, use_ratio=True,
                                                                               th_dist=255)
        assert ref[0] == (1 if accepted else 0), "the oracle itself must follow :186-231"
        assert got[0] == ref[0]
        np.testing.assert_array_equal(got[1], ref[1]); np.testing.assert_array_equal(got[2], ref[2])


@pytest.mark.parametrize("seed,th,fwd,bwd,ori,ratio", [(1, 7.0, False, False, True, False), (2, 15.0, True, False, True, True), (3, 15.0, False, True, False, True)])
def test_query_form_reports_what_every_query_did(ext, oracle, seed, th, fwd, bwd, ori, ratio):
    """gfo_search_by_projection_queries_points: the frame-to-frame cases again, with every query's own outcome (the keypoint and distance it took
    at its turn, or why it took none) -- before the rotation check, which only touches the slots."""
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    q, qd = _frame_queries(oracle, kl, dl, seed, th, fwd, bwd)
    taken = (np.random.default_rng(seed).random(len(kl)) < 0.1).astype(np.uint8)
    b = (0.0, 0.0, 752.0, 480.0)
    ref = oracle.search_by_projection_queries_points(kl, dl, u, kl["angle"], b, q, qd, ratio, 0.9, 100, ori, taken)
    got = G.ORBmatcher(0.9, ori, extractor=ext).SearchByProjectionQueriesPoints(kl, dl, u, kl["angle"], b, q, qd, use_ratio=ratio, kp_taken=taken)
    assert got[0] == ref[0] and ref[0] > 300
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[3], ref[3])
    assert (ref[3] == -1).any() and (ref[3] == -3).any() and ((ref[3] == -2).any() == ratio or not ratio)


def test_non_blocking_queries_are_independent_searches(ext, oracle):
    """The search inside ORBmatcher::Fuse(KF, Scw, ...) (ORBmatcher.cc:1089-1212): every map point looks for its best keypoint of the two predicted levels in
    a window, TH_LOW, no ratio test, and NOTHING it finds hides a keypoint from the next point.  As queries that block nothing: each query's
    outcome is what it would be alone, whatever the order -- checked by shuffling the queries and against a single-query call per sample."""
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    q, qd = _frame_queries(oracle, kl, dl, 9, 3.0)
    q["flags"] &= ~4                                  # nothing blocks
    lev = kl["octave"][np.random.default_rng(9).integers(0, len(kl), len(q))]
    q["min_level"], q["max_level"] = lev - 1, lev     # kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel: skipped (:1179)
    b = (0.0, 0.0, 752.0, 480.0)
    mt = G.ORBmatcher(0.9, False, extractor=ext)
    ref = oracle.search_by_projection_queries_points(kl, dl, None, kl["angle"], b, q, qd, False, 0.9, 50, False, None)
    got = mt.SearchByProjectionQueriesPoints(kl, dl, None, kl["angle"], b, q, qd, use_ratio=False, th_dist=50)
    np.testing.assert_array_equal(got[3], ref[3])
    assert (got[3] >= 0).sum() > 150
    perm = np.random.default_rng(1).permutation(len(q))
    shuffled = mt.SearchByProjectionQueriesPoints(kl, dl, None, kl["angle"], b, q[perm], qd[perm], use_ratio=False, th_dist=50)
    np.testing.assert_array_equal(shuffled[3], got[3][perm])
    for i in (0, 17, 400, 1234):
        alone = mt.SearchByProjectionQueriesPoints(kl, dl, None, kl["angle"], b, q[i:i + 1], qd[i:i + 1], use_ratio=False, th_dist=50)
        assert alone[3][0] == got[3][i]
    # several points may name the same keypoint; the slot table keeps the last of them (Fuse decides per point what to do with it)
    k = got[3][got[3] >= 0] & 0xFFFF
    assert len(np.unique(k)) < len(k)


@pytest.mark.parametrize("seed,m,jitter,th", [(1, 2000, 1.0, 3.0), (2, 6000, 2.5, 4.0), (3, 20000, 1.5, 3.0)])
def test_fusion_search_with_the_reprojection_error_gate(ext, oracle, seed, m, jitter, th):
    """The search of ORBmatcher::Fuse(KeyFrame*, MapPoints, th) (ORBmatcher.cc:1000-1063; gfo_search_for_fusion): candidates are gated by
    e2 * mvInvLevelSigma2[keypoint level] against 5.99 / 7.8 instead of the mvuRight window, nothing blocks, TH_LOW.  20 000 points go to
    the device in two pieces (they are independent); the third form of this module's fixture asks for the thread-per-point kernels, which a
    fusion search never takes."""
    import gf_orb_slam2_amd as G
    kl, dl, u = _frame(oracle)
    rng = np.random.default_rng(seed)
    sf = oracle.OracleExtractor().scale_factors
    inv_sigma2 = (1.0 / (sf.astype(np.float32) ** 2)).astype(np.float32)
    src = rng.integers(0, len(kl), m)
    q = np.zeros(m, oracle.PROJ_QUERY_DTYPE)
    q["u"] = kl["x"][src] + rng.normal(0, jitter, m); q["v"] = kl["y"][src] + rng.normal(0, jitter, m)
    q["ur"] = np.where(u[src] >= 0, u[src] + rng.normal(0, jitter, m), q["u"] - 5).astype(np.float32)
    lev = np.clip(kl["octave"][src] + rng.integers(0, 2, m), 0, 7)
    q["radius"] = (np.float32(th) * sf[lev]).astype(np.float32)
    q["min_level"], q["max_level"] = lev - 1, lev
    q["flags"] = np.where(rng.random(m) < 0.95, 1 | 4, 0)
    qd = dl[src].copy()
    for _ in range(10):
        sel = rng.random(m) < 0.5
        bits = rng.integers(0, 256, m)
        qd[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    uu = u.copy()
    uu[rng.random(len(uu)) < 0.05] = 0.0          # mvuRight == 0 counts as a right-image coordinate here (>= 0), unlike the tracking overloads (> 0)
    b = (0.0, 0.0, 752.0, 480.0)
    ref = oracle.search_for_fusion(kl, dl, uu, b, inv_sigma2, q, qd, 50)
    got = G.ORBmatcher(0.8, True, extractor=ext).SearchForFusion(kl, dl, uu, b, inv_sigma2, q, qd)
    np.testing.assert_array_equal(got, ref)
    assert (ref >= 0).sum() > m // 4 and (ref == -3).any() and (ref == -1).any()


@pytest.mark.parametrize("seed,flips,sigma,window,ratio,ori", [(0, 8, 15.0, 100, 0.9, True), (1, 16, 5.0, 30, 0.9, False), (2, 4, 40.0, 100, 0.7, True),
                                                               (3, 10, 10.0, 1000, 0.9, True), (4, 8, 15.0, 0, 0.9, True)])
def test_search_for_initialization(oracle, seed, flips, sigma, window, ratio, ori):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cc:520-633), the monocular bootstrap: level-0 keypoints only, windows around
    vbPrevMatched, a later keypoint takes a keypoint of F2 from an earlier one when strictly closer, vbPrevMatched updated -- and the call
    repeated on the updated vector, as Tracking::MonocularInitialization does frame after frame.  Window 1000 covers the frame (a table
    larger than the first guess: the entry asks again); window 0 finds only keypoints at the very position."""
    import gf_orb_slam2_amd as G
    import gf_cases
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), oracle.KEYPOINT_DTYPE)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    rng = np.random.default_rng(seed)
    kp2, d2, prev = gf_cases.initialization_case(oracle, kl, dl, rng, flips=flips, sigma=sigma)
    bounds = (0.0, 0.0, 752.0, 480.0)
    e = G.ORBextractor(2000, 1.2, 8, 20, 7)
    try:
        M = G.ORBmatcher(ratio, ori, extractor=e)
        p_ref, p_got = prev.copy(), prev.copy()
        for rep in range(2):
            ref = oracle.search_for_initialization(kl, dl, p_ref, kp2, d2, bounds, window, ratio, ori)
            got = M.SearchForInitialization(kl, dl, p_got, kp2, d2, bounds, window)
            assert got[0] == ref[0], rep
            np.testing.assert_array_equal(got[1], ref[1])
            assert p_got.tobytes() == p_ref.tobytes()
            m = ref[1] >= 0
            assert (kl["octave"][m] == 0).all() and (kp2["octave"][ref[1][m]] == 0).all()
            assert len(set(ref[1][m].tolist())) == int(m.sum())            # a keypoint of F2 belongs to one keypoint of F1 at the end
            if window == 100:
                assert ref[0] > 100, ref[0]
    finally:
        e.close()


def test_search_for_initialization_edge_cases(oracle):
    import gf_orb_slam2_amd as G
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), oracle.KEYPOINT_DTYPE)[:400]
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)[:400]
    bounds = (0.0, 0.0, 752.0, 480.0)
    e = G.ORBextractor(2000, 1.2, 8, 20, 7)
    try:
        M = G.ORBmatcher(0.9, True, extractor=e)
        prev = np.stack([kl["x"], kl["y"]], 1).astype(np.float32)
        # a frame against itself: every level-0 keypoint finds itself at distance 0 (unless a twin descriptor breaks the ratio test)
        n, out = M.SearchForInitialization(kl, dl, prev.copy(), kl, dl, bounds, 100)
        ref = oracle.search_for_initialization(kl, dl, prev.copy(), kl, dl, bounds, 100, 0.9, True)
        assert n == ref[0] and (out == ref[1]).all() and n > 0.8 * int((kl["octave"] == 0).sum())
        assert (out[out >= 0] == np.flatnonzero(out >= 0)).all()
        # no level-0 keypoint in F1: nothing searches
        up = kl.copy(); up["octave"] = np.maximum(up["octave"], 1)
        n, out = M.SearchForInitialization(up, dl, prev.copy(), kl, dl, bounds, 100)
        assert n == 0 and (out == -1).all()
        # empty sides
        n, out = M.SearchForInitialization(kl[:0], dl[:0], np.zeros((0, 2), np.float32), kl, dl, bounds, 100)
        assert n == 0 and len(out) == 0
        p = prev.copy()
        n, out = M.SearchForInitialization(kl, dl, p, kl[:0], dl[:0], bounds, 100)
        assert n == 0 and (out == -1).all() and p.tobytes() == prev.tobytes()
        with pytest.raises(G.GfoError):
            M.SearchForInitialization(kl, dl, prev.copy(), kl, dl, bounds, -1)
        with pytest.raises(ValueError):
            M.SearchForInitialization(kl, dl, prev.astype(np.float64), kl, dl, bounds, 100)
    finally:
        e.close()
