"""GPU parity tests of the good-feature matchers through the C ABI: gfo_search_by_projection_points against the oracle's literal
statements of ORBmatcher::SearchByProjection_Budget (src/ORBmatcher.cc:45-153) and SearchByProjection_OnePoint
(include/ORBmatcher.h:71-150).  Indices, distances and the three ways a point leaves the loop body: bit-exact."""
import numpy as np
import pytest

import gf_cases as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["wavefront_per_point", "wavefront_per_point_small_pool", "thread_per_point"], autouse=True)
def round0_form(request, monkeypatch):
    """Both forms of round 0 write the points' first outcome (tests/test_gpu_projection.py has the details of the three settings)."""
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


@pytest.mark.parametrize("seed,m,th,ratio,sigma", [(1, 1500, 1.0, 0.8, 3.0), (2, 4000, 0.5, 0.8, 2.0), (3, 3000, 3.0, 0.9, 3.0),
                                                   (4, 20000, 1.0, 0.8, 3.0), (5, 9000, 5.0, 0.7, 6.0)])
def test_every_point_reports_what_it_did(ext, oracle, seed, m, th, ratio, sigma):
    import gf_orb_slam2_amd as G
    kl, dl, u, _ = gc.frame(oracle)
    sf = ext.GetScaleFactors()
    mps, mpd, taken = gc.contended_map(oracle, kl, dl, seed, m, sigma)
    ref = oracle.search_by_projection_budget(kl, dl, u, sf, gc.BOUNDS, mps, mpd, th, ratio, taken, 0)
    mt = G.ORBmatcher(ratio, True, extractor=ext)
    nm, out_mp, out_sc, out_pt = mt.SearchByProjectionPoints(kl, dl, u, sf, gc.BOUNDS, mps, mpd, th, taken)
    assert nm == ref[0] and nm > 100
    np.testing.assert_array_equal(out_pt, ref[3])
    np.testing.assert_array_equal(out_mp, ref[1])
    np.testing.assert_array_equal(out_sc, ref[2])
    for code in (mt.POINT_NONE, mt.POINT_FAR) + ((mt.POINT_RATIO,) if th >= 1.0 else ()):   # (half-size windows rarely hold two close candidates)
        assert (out_pt == code).any()
    # the plain entry point is unchanged by the extra output
    plain = mt.SearchByProjection(kl, dl, u, sf, gc.BOUNDS, mps, mpd, th, taken)
    assert plain[0] == nm
    np.testing.assert_array_equal(plain[1], out_mp)
    np.testing.assert_array_equal(plain[2], out_sc)


def test_the_clock_of_the_budget_matcher_cuts_a_prefix(ext, oracle):
    """SearchByProjection_Budget with a clock that trips at its k-th reading = the first `cut` points of ONE device call."""
    import gf_orb_slam2_amd as G
    kl, dl, u, _ = gc.frame(oracle)
    sf = ext.GetScaleFactors()
    mps, mpd, taken = gc.contended_map(oracle, kl, dl, 5, 2500)
    mt = G.ORBmatcher(0.8, True, extractor=ext)
    _, _, _, out_pt = mt.SearchByProjectionPoints(kl, dl, u, sf, gc.BOUNDS, mps, mpd, 1.0, taken)
    for k in (1, 2, 17, 300, 100000):
        ref = oracle.search_by_projection_budget(kl, dl, u, sf, gc.BOUNDS, mps, mpd, 1.0, 0.8, taken, k)
        cut = gc.clock_cut(out_pt, k)
        nm, pm, ps = mt.points_prefix(out_pt, cut, len(kl))
        assert nm == ref[0]
        np.testing.assert_array_equal(pm, ref[1])
        np.testing.assert_array_equal(ps, ref[2])
        np.testing.assert_array_equal((out_pt[:cut] >= 0).astype(np.int32), ref[4][:cut])     # IncreaseFound() per match
        assert not ref[4][cut:].any()


@pytest.mark.parametrize("num_to_match", [30, 200, 1000])
def test_baseline_map_matching_is_a_weighted_prefix(ext, oracle, num_to_match):
    """Observability::runBaselineMapMatching (src/Observability.cc:1233-1262): the in-view points sorted by life, SearchByProjection_OnePoint
    one after the other, `nMatched += 2` per match and `+ 1` when the matched keypoint has a depth, until nMatched >= num_to_match.  The
    oracle walks that loop literally; the device answers all points in the sorted order at once and the loop's exit becomes a prefix."""
    import gf_orb_slam2_amd as G
    kl, dl, u, depth = gc.frame(oracle)
    sf = ext.GetScaleFactors()
    mps, mpd, taken = gc.contended_map(oracle, kl, dl, 11, 3000)
    rng = np.random.default_rng(3)
    n_visible = rng.integers(1, 50, len(mps))
    valid = np.flatnonzero((mps["flags"] & 1).astype(bool) & ~(mps["flags"] & 2).astype(bool))      # :1203-1211
    order = valid[np.argsort(-n_visible[valid], kind="stable")]                                    # BASELINE_LONGLIVE: sort by life
    pf = oracle.ProjectionFrame(kl, dl, u, sf, gc.BOUNDS, taken)
    n_matched, walked = 0, 0
    for p in order:
        if n_matched >= num_to_match:
            break
        best, _ = pf.one_point(mps[p], mpd[p], 1.0, 0.8, walked)
        walked += 1
        if best >= 0:
            n_matched += 2
            if depth[best] >= 0:
                n_matched += 1
    ref_mp, ref_sc = pf.state()
    mt = G.ORBmatcher(0.8, True, extractor=ext)
    _, _, _, out_pt = mt.SearchByProjectionPoints(kl, dl, u, sf, gc.BOUNDS, mps[order], mpd[order], 1.0, taken)
    got, cut = 0, 0
    while cut < len(order) and got < num_to_match:
        v = int(out_pt[cut])
        cut += 1
        if v >= 0:
            got += 2 + (1 if depth[v & 0xFFFF] >= 0 else 0)
    assert cut == walked and got == n_matched and cut < len(order)
    _, pm, ps = mt.points_prefix(out_pt, cut, len(kl))
    np.testing.assert_array_equal(pm, ref_mp)
    np.testing.assert_array_equal(ps, ref_sc)


def test_points_edge_cases(ext, oracle):
    import gf_orb_slam2_amd as G
    kl, dl, u, _ = gc.frame(oracle)
    sf = ext.GetScaleFactors()
    mt = G.ORBmatcher(0.8, True, extractor=ext)
    z = np.zeros(0, oracle.MAP_POINT_DTYPE)
    nm, out_mp, _, out_pt = mt.SearchByProjectionPoints(kl, dl, None, sf, gc.BOUNDS, z, np.zeros((0, 32), np.uint8), 1.0, None)
    assert nm == 0 and (out_mp == -1).all() and len(out_pt) == 0
    mps, mpd, _ = gc.contended_map(oracle, kl, dl, 2, 64)
    nm, out_mp, _, out_pt = mt.SearchByProjectionPoints(kl[:0], dl[:0], None, sf, gc.BOUNDS, mps, mpd, 1.0, None)
    assert nm == 0 and (out_pt == -1).all()
    # every point out of view / bad
    mps["flags"] = 2 | 4
    nm, _, _, out_pt = mt.SearchByProjectionPoints(kl, dl, u, sf, gc.BOUNDS, mps, mpd, 1.0, None)
    assert nm == 0 and (out_pt == -1).all()
    # the prefix helper refuses what cannot be a result of the call
    with pytest.raises(ValueError):
        mt.points_prefix(np.array([5 | (3 << 16)], np.int32), 1, 4)
    assert mt.points_prefix(np.array([2 | (3 << 16), -3, 2 | (9 << 16)], np.int32), 3, 4)[1].tolist() == [-1, -1, 2, -1]


def _entries(cand):
    return (cand & 0xFFFF).astype(np.int64), ((cand >> 16) & 0xF).astype(np.int64), ((cand >> 20) & 0x1FF).astype(np.int64), (cand >> 31).astype(bool)


def _check_table(oracle, kl, dl, u, sf, mps, mpd, th, start, cand):
    """Every list against ORBmatcher::GetCandidates as the oracle states it (order included); every entry's octave, distance and gate
    against their definitions (numpy)."""
    m = len(mps)
    assert start[0] == 0 and (np.diff(start) >= 0).all() and start[m] == len(cand)
    pf = oracle.ProjectionFrame(kl, dl, u, sf, gc.BOUNDS, None)
    idx, octv, dist, gated = _entries(cand)
    for p in range(m):
        np.testing.assert_array_equal(idx[start[p]:start[p + 1]], pf.candidates(mps[p], th), err_msg=f"point {p}")
    owner = np.repeat(np.arange(m), np.diff(start))
    np.testing.assert_array_equal(octv, kl["octave"][idx])
    x = np.bitwise_xor(mpd[owner], dl[idx])
    np.testing.assert_array_equal(dist, np.unpackbits(x, axis=1).sum(axis=1))
    r = np.where(mps["view_cos"].astype(np.float64) > 0.998, np.float32(2.5), np.float32(4.0)).astype(np.float32)
    if th != 1.0:
        r = (r * np.float32(th)).astype(np.float32)
    rs = (r * sf[np.clip(mps["level"], 0, len(sf) - 1)]).astype(np.float32)
    if u is None:
        assert not gated.any()
    else:
        ur = u[idx]
        want = (ur > 0) & (np.abs((mps["proj_xr"][owner] - ur).astype(np.float32)) > rs[owner])
        np.testing.assert_array_equal(gated, want)


@pytest.mark.parametrize("seed,m,th,sigma", [(1, 1500, 1.0, 3.0), (2, 4000, 0.5, 2.0), (3, 3000, 3.0, 3.0), (5, 6000, 5.0, 6.0)])
def test_candidate_table_is_get_candidates_for_every_point(ext, oracle, seed, m, th, sigma):
    import gf_orb_slam2_amd as G
    kl, dl, u, _ = gc.frame(oracle)
    sf = ext.GetScaleFactors()
    mps, mpd, _ = gc.contended_map(oracle, kl, dl, seed, m, sigma)
    mt = G.ORBmatcher(0.8, True, extractor=ext)
    start, cand = mt.GetCandidates(kl, dl, u, sf, gc.BOUNDS, mps, mpd, th)
    assert len(cand) > m // 4
    _check_table(oracle, kl, dl, u, sf, mps, mpd, th, start, cand)


def test_candidate_lists_longer_than_a_wavefront(ext, oracle):
    """Seven hundred keypoints inside one window: lists ranked in LDS in several rounds (65..256 entries) and the ones that go through
    device memory instead (beyond 256)."""
    import gf_orb_slam2_amd as G
    kl, dl, u, _ = gc.frame(oracle)
    kl = kl.copy()
    rng = np.random.default_rng(8)
    crowd = rng.choice(len(kl), 700, replace=False)
    kl["x"][crowd] = 300 + rng.uniform(-12, 12, 700).astype(np.float32)
    kl["y"][crowd] = 200 + rng.uniform(-12, 12, 700).astype(np.float32)
    kl["octave"][crowd] = rng.integers(0, 2, 700)
    sf = ext.GetScaleFactors()
    mps, mpd, _ = gc.contended_map(oracle, kl, dl, 4, 600)
    mps["proj_x"][:200] = 300 + rng.uniform(-6, 6, 200); mps["proj_y"][:200] = 200 + rng.uniform(-6, 6, 200)
    mps["level"][:200] = rng.integers(0, 3, 200); mps["flags"][:200] = 5
    mt = G.ORBmatcher(0.8, True, extractor=ext)
    start, cand = mt.GetCandidates(kl, dl, u, sf, gc.BOUNDS, mps, mpd, 5.0)
    sizes = np.diff(start)
    assert sizes.max() > 256 and ((sizes > 64) & (sizes <= 256)).any(), np.sort(sizes)[-10:]
    _check_table(oracle, kl, dl, u, sf, mps, mpd, 5.0, start, cand)


@pytest.mark.parametrize("seed,m,th,ratio", [(1, 1500, 1.0, 0.8), (7, 5000, 3.0, 0.9)])
def test_table_then_match_equals_one_point_in_any_order(ext, oracle, seed, m, th, ratio):
    """The use the table is for: SearchByProjection_OnePoint called in an order nobody knows in advance (a shuffle stands for the greedy
    selection of Observability::runActiveMapMatching).  Oracle: the literal function on a frame that carries its slots from call to
    call.  Here: one device call for the table, then gfo_match_candidates per point with the caller keeping the slots."""
    import gf_orb_slam2_amd as G
    kl, dl, u, _ = gc.frame(oracle)
    sf = ext.GetScaleFactors()
    mps, mpd, taken = gc.contended_map(oracle, kl, dl, seed, m)
    mt = G.ORBmatcher(ratio, True, extractor=ext)
    start, cand = mt.GetCandidates(kl, dl, u, sf, gc.BOUNDS, mps, mpd, th)
    pf = oracle.ProjectionFrame(kl, dl, u, sf, gc.BOUNDS, taken)
    slot_taken = taken.copy()
    slot_mp = np.full(len(kl), -1, np.int32); slot_sc = np.zeros(len(kl), np.int32)
    why_code = {1: mt.POINT_RATIO, 2: mt.POINT_FAR, 3: mt.POINT_NONE}
    nmatch = 0
    for p in np.random.default_rng(seed).permutation(m):
        ref, why = pf.one_point(mps[p], mpd[p], th, ratio, int(p))
        got, d = mt.MatchCandidates(cand[start[p]:start[p + 1]], slot_taken)
        if ref >= 0:
            assert got == ref
            slot_mp[got] = p; slot_sc[got] = d
            slot_taken[got] = 1 if mps["flags"][p] & 4 else 0       # F.mvpMapPoints[bestIdx] = pMP: later points ask ITS Observations()
            nmatch += 1
        else:
            assert got == why_code[why], (p, got, why)
    assert nmatch > 200
    st = pf.state()
    np.testing.assert_array_equal(slot_mp, st[0])
    np.testing.assert_array_equal(slot_sc[slot_mp >= 0], st[1][slot_mp >= 0])


def test_candidate_table_capacity_and_edges(ext, oracle):
    import ctypes as C
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd._lib import ptr, FrameBoundsC
    kl, dl, u, _ = gc.frame(oracle)
    sf = ext.GetScaleFactors()
    mps, mpd, _ = gc.contended_map(oracle, kl, dl, 1, 500)
    mt = G.ORBmatcher(0.8, True, extractor=ext)
    start, cand = mt.GetCandidates(kl, dl, u, sf, gc.BOUNDS, mps, mpd, 1.0)
    start1, cand1 = mt.GetCandidates(kl, dl, u, sf, gc.BOUNDS, mps, mpd, 1.0, cap=1)      # first answer: GFO_ERR_CAPACITY and the size
    np.testing.assert_array_equal(start, start1); np.testing.assert_array_equal(cand, cand1)
    # the raw call: too small an array is refused with the size, offsets valid
    fb = FrameBoundsC(*gc.BOUNDS)
    st = np.zeros(len(mps) + 1, np.int32); small = np.zeros(8, np.uint32); tot = C.c_int()
    kp = np.ascontiguousarray(kl); sfa = np.ascontiguousarray(sf, np.float32)
    rc = mt._L.gfo_projection_candidates(mt._ctx, ptr(kp), ptr(dl), ptr(u), len(kl), ptr(sfa), len(sfa), C.byref(fb), ptr(mps), ptr(mpd),
                                         len(mps), 1.0, ptr(st), ptr(small), 8, C.byref(tot))
    assert rc == -3 and tot.value == len(cand)
    np.testing.assert_array_equal(st, start)
    # nothing to do
    z = np.zeros(0, oracle.MAP_POINT_DTYPE)
    s0, c0 = mt.GetCandidates(kl, dl, u, sf, gc.BOUNDS, z, np.zeros((0, 32), np.uint8), 1.0)
    assert s0.tolist() == [0] and len(c0) == 0
    s1, c1 = mt.GetCandidates(kl[:0], dl[:0], None, sf, gc.BOUNDS, mps, mpd, 1.0)
    assert not s1.any() and len(c1) == 0
    assert mt.MatchCandidates(np.zeros(0, np.uint32), None)[0] == mt.POINT_NONE
    # an entry that names a keypoint the frame does not have (another frame's table) is ignored, not read
    foreign = np.array([40000 | (3 << 20), 7 | (9 << 20)], np.uint32)
    assert mt.MatchCandidates(foreign, np.zeros(100, np.uint8)) == (7, 9)
    assert mt._L.gfo_match_candidates(None, 3, None, 10, C.c_float(0.8), None) < mt.POINT_FAR


def test_golden_good_feature_vectors(ext, oracle):
    """The committed vectors (tests/golden/EuRoC_gf_matchers.npz) through the C ABI: per-point outcomes at th 0.5 and 1, the clock's
    prefix, the candidate table's lists, SearchByBoW between the two golden extractions as keyframes."""
    import os
    import gf_orb_slam2_amd as G
    from conftest import GOLDEN
    kl, dl, u, _ = gc.frame(oracle)
    kr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_kp.bin"), oracle.KEYPOINT_DTYPE)
    dr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_desc.bin"), np.uint8).reshape(-1, 32)
    p = np.load(os.path.join(GOLDEN, "EuRoC_projection.npz"))
    g = np.load(os.path.join(GOLDEN, "EuRoC_gf_matchers.npz"))
    sf = ext.GetScaleFactors()
    mt = G.ORBmatcher(0.8, True, extractor=ext)
    for tag, th in (("th05", 0.5), ("th1", 1.0)):
        nm, out_mp, out_sc, out_pt = mt.SearchByProjectionPoints(kl, dl, u, sf, gc.BOUNDS, p["mps"], p["mp_desc"], th, p["taken"])
        assert nm == int(g[f"{tag}_nmatches"])
        np.testing.assert_array_equal(out_pt, g[f"{tag}_out_point"])
        np.testing.assert_array_equal(out_mp, g[f"{tag}_out_mp"])
        np.testing.assert_array_equal(out_sc, g[f"{tag}_out_score"])
        np.testing.assert_array_equal((out_pt >= 0).astype(np.int32), g[f"{tag}_found"])
    nm5, mp5, sc5 = mt.points_prefix(out_pt, gc.clock_cut(out_pt, 5), len(kl))
    assert nm5 == int(g["th1_trip5_nmatches"])
    np.testing.assert_array_equal(mp5, g["th1_trip5_out_mp"]); np.testing.assert_array_equal(sc5, g["th1_trip5_out_score"])
    start, cand = mt.GetCandidates(kl, dl, u, sf, gc.BOUNDS, p["mps"], p["mp_desc"], 1.0)
    np.testing.assert_array_equal(start, g["th1_cand_start"])
    np.testing.assert_array_equal((cand & 0xFFFF).astype(np.int32), g["th1_cand_idx"])
    n1 = (dl[:, 0] >> 2).astype(np.int64); n2 = (dr[:, 0] >> 2).astype(np.int64)
    for ori in (0, 1):
        nmk, o12 = G.ORBmatcher(0.75, bool(ori), extractor=ext).SearchByBoWKeyFrames(dl, kl["angle"], g["bowkf_valid1"], oracle.make_feature_vector(n1), dr,
                                                                                     kr["angle"], g["bowkf_valid2"], oracle.make_feature_vector(n2))
        assert nmk == int(g[f"bowkf_ori{ori}_nmatches"])
        np.testing.assert_array_equal(o12, g[f"bowkf_ori{ori}_out12"])
