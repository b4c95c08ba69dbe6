"""GPU parity tests of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) through the C ABI vs the CPU
oracle.  Indices bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    import gf_orb_slam2_amd as G
    e = G.ORBextractor(2000, 1.2, 8, 20, 7)
    yield e
    e.close()


def _case(oracle, seed, flips, node_shift, angle_sigma, p_invalid=0.1):
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    rng = np.random.default_rng(seed)
    fd = dl.copy()
    for _ in range(flips):
        sel = rng.random(len(fd)) < 0.5
        bits = rng.integers(0, 256, len(fd))
        fd[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    perm = rng.permutation(len(fd))
    fd = fd[perm]
    fa = ((kl["angle"][perm] + rng.normal(0, angle_sigma, len(fd))) % 360).astype(np.float32)
    # a toy vocabulary: the node of a descriptor is its leading bits (a real one is a k-means tree)
    node_k = (dl[:, 0] >> node_shift).astype(np.int64)
    node_f = (fd[:, 0] >> node_shift).astype(np.int64)
    node_k[rng.random(len(node_k)) < 0.02] = -1          # keypoints missing from the feature vector
    valid = (rng.random(len(dl)) >= p_invalid).astype(np.uint8)
    return dl, kl["angle"].copy(), valid, oracle.make_feature_vector(node_k), fd, fa, oracle.make_feature_vector(node_f)


@pytest.mark.parametrize("seed,flips,shift,sigma,ratio,ori", [(0, 10, 2, 5.0, 0.7, True), (1, 4, 4, 40.0, 0.9, True),
                                                              (2, 20, 0, 2.0, 0.6, False), (3, 0, 6, 0.0, 0.75, True)])
def test_bow_matches_oracle(ext, oracle, seed, flips, shift, sigma, ratio, ori):
    import gf_orb_slam2_amd as G
    kd_, ka, valid, kfv, fd, fa, ffv = _case(oracle, seed, flips, shift, sigma)
    ref = oracle.search_by_bow(kd_, ka, valid, kfv, fd, fa, ffv, ratio, ori)
    got = G.ORBmatcher(ratio, ori, extractor=ext).SearchByBoW(kd_, ka, valid, kfv, fd, fa, ffv)
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1])
    if flips <= 10:
        assert ref[0] > 200


@pytest.mark.parametrize("K", [1, 40, 150, 600, 100000])
def test_bow_with_a_feature_budget(ext, oracle, K):
    """gfo_search_by_bow_budget = SearchByBoW compiled with BUDGETING_FEATURE_MATCHING (ORBmatcher.h:36-37, ORBmatcher.cc:360-365): the
    break leaves ONE node's keyframe loop, so every common node after the one that reaches the budget still adds its first match;
    nodes of a few and of hundreds of keypoints (both k_bow_match paths), with and without the rotation check"""
    import gf_orb_slam2_amd as G
    for seed, shift, ori in ((0, 2, True), (1, 4, True), (2, 6, False), (3, 0, True)):
        kd_, ka, valid, kfv, fd, fa, ffv = _case(oracle, seed, 6, shift, 5.0)
        with oracle.feature_budget(K):
            ref = oracle.search_by_bow(kd_, ka, valid, kfv, fd, fa, ffv, 0.75, ori)
        got = G.ORBmatcher(0.75, ori, extractor=ext).SearchByBoW(kd_, ka, valid, kfv, fd, fa, ffv, max_matches=K)
        assert got[0] == ref[0], (K, seed)
        np.testing.assert_array_equal(got[1], ref[1])
    full = oracle.search_by_bow(kd_, ka, valid, kfv, fd, fa, ffv, 0.75, ori)
    assert (ref[0] == full[0]) == (K >= full[0])


def test_bow_edge_cases(ext, oracle):
    import gf_orb_slam2_amd as G
    m = G.ORBmatcher(0.7, True, extractor=ext)
    kd_, ka, valid, kfv, fd, fa, ffv = _case(oracle, 5, 6, 3, 3.0)
    empty = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.uint32))
    assert m.SearchByBoW(kd_, ka, valid, empty, fd, fa, ffv)[0] == 0
    assert m.SearchByBoW(kd_, ka, valid, kfv, fd, fa, empty)[0] == 0
    none_valid = np.zeros_like(valid)
    nm, out = m.SearchByBoW(kd_, ka, none_valid, kfv, fd, fa, ffv)
    assert nm == 0 and (out == -1).all()
    # disjoint vocabularies
    kfv2 = (kfv[0] + 1000, kfv[1], kfv[2])
    assert m.SearchByBoW(kd_, ka, valid, kfv2, fd, fa, ffv)[0] == 0


@pytest.mark.parametrize("k,depth,levelsup,seed", [(10, 4, 2, 0), (6, 5, 4, 1), (10, 3, 4, 2), (4, 6, 3, 3)])
def test_vocabulary_transform_matches_oracle(ext, oracle, k, depth, levelsup, seed):
    """Frame::ComputeBoW's tree descent (DBoW2 transform) on a synthetic vocabulary: word, weight and the node
    at level L - levelsup, first minimum on ties."""
    import gf_orb_slam2_amd as G
    voc = oracle.make_vocabulary(k, depth, seed)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    # some descriptors equal to node centres and some equidistant ones, to hit exact ties
    extra = voc["descriptors"][np.random.default_rng(seed).integers(1, len(voc["descriptors"]), 64)]
    desc = np.concatenate([dl, extra, np.zeros((3, 32), np.uint8), np.full((3, 32), 255, np.uint8)])
    ref = oracle.bow_transform(voc, desc, levelsup)
    v = G.ORBVocabulary(voc, ext)
    got = v.transform_raw(desc, levelsup)
    for a, b in zip(got, ref):
        assert a.tobytes() == b.tobytes()
    # and the chained use: transform -> SearchByBoW agrees with the oracle run on the oracle's transform
    bow, fv = v.transform(dl, levelsup)
    assert abs(sum(bow.values()) - 1.0) < 1e-5
    wid, wt, nid = ref[0][:len(dl)], ref[1][:len(dl)], ref[2][:len(dl)]
    fv_ref = oracle.make_feature_vector(np.where(wt > 0, nid, -1))
    for a, b in zip(fv, fv_ref):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("weighting,norm", [("TF_IDF", "L1"), ("TF_IDF", None), ("TF", "L2"), ("IDF", "L1"), ("BINARY", None)])
def test_compute_bow_folds_both_maps_on_the_device(oracle, weighting, norm):
    """Frame::ComputeBoW in full (gfo_compute_bow): descent + the fold into BowVector / FeatureVector, against the oracle's
    literal std::map statement (addWeight / addIfNotExist / normalize in the reference's summation order, double
    weights): word ids, node ids, index lists identical, WordValues bit for bit."""
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(500, 1.2, 8, 20, 7)
    W = {"TF_IDF": 0, "TF": 1, "IDF": 2, "BINARY": 3}[weighting]
    Nn = {None: 0, "L1": 1, "L2": 2}[norm]
    for seed, k, depth, n in ((0, 10, 3, 2000), (1, 4, 5, 3500), (2, 6, 2, 8192), (3, 10, 4, 1), (4, 10, 4, 77), (5, 8, 3, 8193), (6, 10, 3, 21000)):
        # (more than 8192 descriptors: the fold sorts in device memory instead of LDS -- round 5, no frame is refused)
        voc = oracle.make_vocabulary(k, depth, seed=seed, p_stop=0.1)
        rng = np.random.default_rng(100 + seed)
        leaves = voc["descriptors"][voc["n_children"] == 0]
        desc = leaves[rng.integers(0, len(leaves), n)].copy()          # near the words: many features share a word
        flips = rng.integers(0, 256, (n, 5))
        for j in range(5):
            desc[np.arange(n), flips[:, j] >> 3] ^= (1 << (flips[:, j] & 7)).astype(np.uint8)
        V = G.ORBVocabulary(voc, ext)
        for levelsup in (1, depth + 1):
            (bw, bv), (fn, fs, fi) = V.compute_bow(desc, levelsup, weighting, norm)
            rw, rv, rn, rs, ri = oracle.compute_bow(voc, desc, levelsup, W, Nn)
            np.testing.assert_array_equal(bw, rw)
            assert bv.tobytes() == rv.tobytes()
            np.testing.assert_array_equal(fn, rn)
            np.testing.assert_array_equal(fs, rs)
            np.testing.assert_array_equal(fi, ri)
            assert len(bw) > 0 and (np.diff(bw.astype(np.int64)) > 0).all() and (np.diff(fn.astype(np.int64)) > 0).all()
    ext.close()


def test_compute_bow_fold_against_the_reference_compiled_fixtures(oracle):
    """gfo_compute_bow against tests/golden/bow_fold.npz -- the outputs of the reference's own BowVector.cpp / FeatureVector.cpp
    (oracle/_ref, compiled unmodified) on the stream of each stored descriptor set: the device descent must reproduce the stored
    (word, weight, node) stream and the device fold the reference's maps, WordValues bit for bit, for 4 weightings x 3 norms."""
    import gf_orb_slam2_amd as G
    fx = np.load(os.path.join(GOLDEN, "bow_fold.npz"))
    ext = G.ORBextractor(500, 1.2, 8, 20, 7)
    cases = 0
    for name in sorted({k.split(".")[0] for k in fx.files if k.startswith("voc")}):
        seed, k, depth, n, levelsup = (int(x) for x in fx[f"{name}.params"])
        voc = oracle.make_vocabulary(k, depth, seed=seed, p_stop=0.1)
        desc = fx[f"{name}.desc"]
        V = G.ORBVocabulary(voc, ext)
        wid, wt, nid = V.transform_raw(desc, levelsup)
        np.testing.assert_array_equal(wid.astype(np.uint32), fx[f"{name}.word"])
        np.testing.assert_array_equal(nid.astype(np.uint32), fx[f"{name}.node"])
        for wn in ("TF_IDF", "TF", "IDF", "BINARY"):
            for nn in (None, "L1", "L2"):
                (bw, bv), (fn, fs, fi) = V.compute_bow(desc, levelsup, wn, nn)
                pre = f"{name}.{wn}.{nn or 'none'}."
                np.testing.assert_array_equal(bw, fx[pre + "bow_words"])
                assert bv.tobytes() == fx[pre + "bow_values"].tobytes(), pre
                np.testing.assert_array_equal(fn, fx[f"{name}.fv_nodes"])
                np.testing.assert_array_equal(fs, fx[f"{name}.fv_start"])
                np.testing.assert_array_equal(fi, fx[f"{name}.fv_items"])
                cases += 1
    assert cases == 60
    ext.close()


def test_vocabulary_at_the_size_the_reference_loads(oracle):
    """ORBvoc (test/test_Stereo.cpp:87, Tracking's mpORBVocabulary): k = 10, L = 6 -- 1 111 111 nodes, 35.5 MB of centres, a million
    words.  A synthetic full tree of that shape through gfo_vocabulary_upload -> gfo_compute_bow (Frame::ComputeBoW, levelsup 4 as
    Frame.cc:666 passes it) -> gfo_search_by_bow between a keyframe's and a frame's vectors, everything against the oracle; what the
    sizes could break are 32-bit offsets (node index x 32 bytes, the 8-byte weights behind 50 MB of other arrays) and the per-call cost."""
    import json
    import time
    import gf_orb_slam2_amd as G
    from conftest import ROOT
    voc = oracle.make_vocabulary_full(10, 6, seed=11)
    assert len(voc["first_child"]) == 1111111 and int((voc["n_children"] == 0).sum()) == 10 ** 6
    ext = G.ORBextractor(2000, 1.2, 8, 20, 7)
    t0 = time.perf_counter()
    V = G.ORBVocabulary(voc, ext)
    t_up = time.perf_counter() - t0
    assert V._L.gfo_vocabulary_nodes(ext.handle) == 1111111
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)
    dk = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    rng = np.random.default_rng(2)
    # the frame: the keyframe's descriptors with a few bits flipped, shuffled; plus descriptors that ARE deep nodes' centres (the last
    # nodes of the array: the largest offsets) and their near neighbours
    fd = dk.copy()
    for _ in range(6):
        sel = rng.random(len(fd)) < 0.5
        bits = rng.integers(0, 256, len(fd))
        fd[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    perm = rng.permutation(len(fd))
    fd = fd[perm]
    deep = voc["descriptors"][-300:].copy()
    deep[100:, 5] ^= 0x10
    fd = np.concatenate([fd, deep])
    fa = np.concatenate([(kl["angle"][perm] + rng.normal(0, 4, len(perm))) % 360, rng.uniform(0, 360, len(deep))]).astype(np.float32)
    # ComputeBoW of both, all four weightings once
    got_k = V.compute_bow(dk, 4, "TF_IDF", "L1")
    ref_k = oracle.compute_bow(voc, dk, 4, 0, 1)
    for (wn, nn), (W, Nn) in zip((("TF_IDF", "L1"), ("TF", "L2"), ("IDF", None), ("BINARY", "L1")), ((0, 1), (1, 2), (2, 0), (3, 1))):
        (bw, bv), (fn, fs, fi) = V.compute_bow(fd, 4, wn, nn)
        rw, rv, rn, rs, ri = oracle.compute_bow(voc, fd, 4, W, Nn)
        np.testing.assert_array_equal(bw, rw)
        assert bv.tobytes() == rv.tobytes(), (wn, nn)
        np.testing.assert_array_equal(fn, rn); np.testing.assert_array_equal(fs, rs); np.testing.assert_array_equal(fi, ri)
    assert int(rw.max()) > 990000 and int(rn.max()) > 100          # words from the far end of the table; level-2 nodes are 11 .. 110
    (bwk, bvk), fvk = got_k
    np.testing.assert_array_equal(bwk, ref_k[0]); assert bvk.tobytes() == ref_k[1].tobytes()
    # the descent alone, levelsup 1 .. 6: node ids at every level of the tree
    for levelsup in (1, 2, 6):
        for a, b in zip(V.transform_raw(fd, levelsup), oracle.bow_transform(voc, fd, levelsup)):
            assert a.tobytes() == b.tobytes(), levelsup
    # SearchByBoW(KF, F) over the two feature vectors
    fvf = (fn, fs, fi)
    valid = (rng.random(len(dk)) > 0.1).astype(np.uint8)
    M = G.ORBmatcher(0.7, True, extractor=ext)
    got = M.SearchByBoW(dk, kl["angle"], valid, fvk, fd, fa, fvf)
    ref = oracle.search_by_bow(dk, kl["angle"], valid, (ref_k[2], ref_k[3], ref_k[4]), fd, fa, (rn, rs, ri), 0.7, True)
    assert got[0] == ref[0] and ref[0] > 500
    np.testing.assert_array_equal(got[1], ref[1])
    # per-call cost with the big vocabulary resident (median of 30)
    ts = []
    for _ in range(33):
        t0 = time.perf_counter()
        V.compute_bow(dk, 4, "TF_IDF", "L1")
        ts.append(time.perf_counter() - t0)
    ts = sorted(ts[3:])
    rec = {"nodes": 1111111, "k": 10, "L": 6, "upload_ms": round(t_up * 1e3, 2), "compute_bow_ms_median": round(ts[len(ts) // 2] * 1e3, 4),
           "descriptors": int(len(dk)), "words": int(len(bwk)), "feature_vector_nodes": int(len(fvk[0])), "search_by_bow_matches": int(got[0])}
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, "bigvoc.json"), "w") as fh:
            json.dump(rec, fh)
    assert rec["compute_bow_ms_median"] < 1.0, rec              # a call stays a call: the tree's size is not in it
    ext.close()


@pytest.mark.parametrize("seed,flips,shift,sigma,ratio,ori", [(0, 10, 2, 5.0, 0.7, True), (1, 4, 4, 40.0, 0.9, True),
                                                              (2, 20, 0, 2.0, 0.6, False), (3, 0, 6, 0.0, 0.75, True), (4, 30, 3, 5.0, 0.95, True)])
def test_bow_between_two_keyframes(ext, oracle, seed, flips, shift, sigma, ratio, ori):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12) (ORBmatcher.cc:635-768, loop closing): a map-point mask on both sides,
    the strict `bestDist1 < TH_LOW`, the answer indexed by the first keyframe.  Seed 4 flips enough bits for distances of exactly
    TH_LOW to occur (the one place the two overloads' thresholds differ)."""
    import gf_orb_slam2_amd as G
    d1, a1, v1, fv1, d2, a2, fv2 = _case(oracle, seed, flips, shift, sigma)
    v2 = (np.random.default_rng(seed + 100).random(len(d2)) >= 0.15).astype(np.uint8)
    ref = oracle.search_by_bow_keyframes(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori)
    got = G.ORBmatcher(ratio, ori, extractor=ext).SearchByBoWKeyFrames(d1, a1, v1, fv1, d2, a2, v2, fv2)
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1])
    m = ref[1] >= 0
    assert v1[m].all() and v2[ref[1][m]].all() and len(set(ref[1][m].tolist())) == int(m.sum())
    if flips <= 10:
        assert ref[0] > 150
    # nothing usable on the second side: nothing matches; the plain overload is untouched by the mask machinery
    none = G.ORBmatcher(ratio, ori, extractor=ext).SearchByBoWKeyFrames(d1, a1, v1, fv1, d2, a2, np.zeros_like(v2), fv2)
    assert none[0] == 0 and (none[1] == -1).all()
    plain = G.ORBmatcher(ratio, ori, extractor=ext).SearchByBoW(d1, a1, v1, fv1, d2, a2, fv2)
    refp = oracle.search_by_bow(d1, a1, v1, fv1, d2, a2, fv2, ratio, ori)
    assert plain[0] == refp[0]
    np.testing.assert_array_equal(plain[1], refp[1])


@pytest.mark.parametrize("seed,flips,shift,noise,only_stereo,ori,mono,forward", [(0, 6, 3, 0.6, False, True, False, False), (1, 2, 5, 0.3, True, True, False, False),
                                                                                 (2, 12, 1, 1.5, False, False, False, True), (3, 0, 0, 0.0, False, True, True, False),
                                                                                 (4, 8, 8, 0.8, False, True, False, False), (5, 4, 2, 0.5, False, True, True, True)])
def test_search_for_triangulation(ext, oracle, seed, flips, shift, noise, only_stereo, ori, mono, forward):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:770-935, local mapping): keypoints WITHOUT map points of two keyframes, the
    epipolar gate of CheckDistEpipolarLine in floats, the epipole gate for mono pairs, the last candidate on a distance tie, no blocking.
    shift 8 puts every keypoint in one node (a 2000 x 2000 sweep); mono runs without mvuRight at all."""
    import gf_orb_slam2_amd as G
    import gf_cases
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), oracle.KEYPOINT_DTYPE)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    c = gf_cases.triangulation_case(oracle, kl, dl, np.random.default_rng(seed), flips=flips, noise=noise, node_shift=shift, forward=forward)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    sg = (sf * sf).astype(np.float32)
    ur1, ur2 = (None, None) if mono else (c["ur1"], c["ur2"])
    ref = oracle.search_for_triangulation(c["kp1"], c["desc1"], c["has1"], ur1, c["fv1"], c["kp2"], c["desc2"], c["has2"], ur2, c["fv2"], sf, sg,
                                          c["f12"], c["ex"], c["ey"], only_stereo, ori)
    got = G.ORBmatcher(0.6, ori, extractor=ext).SearchForTriangulation(c["kp1"], c["desc1"], c["has1"], ur1, c["fv1"], c["kp2"], c["desc2"], c["has2"], ur2,
                                                                       c["fv2"], sf, sg, c["f12"], c["ex"], c["ey"], only_stereo)
    assert got[0] == ref[0]
    np.testing.assert_array_equal(got[1], ref[1])
    m = ref[1] >= 0
    assert not c["has1"][m].any() and not c["has2"][ref[1][m]].any()
    if only_stereo and not mono:
        assert (c["ur1"][m] >= 0).all() and (c["ur2"][ref[1][m]] >= 0).all()
    if mono and not forward:
        assert ref[0] > 300          # the geometry is exact and the descriptors equal: most unmatched keypoints pair up
    # the epipolar gate does reject: without it (a sigma table of 1e9) strictly more pairs
    loose = oracle.search_for_triangulation(c["kp1"], c["desc1"], c["has1"], ur1, c["fv1"], c["kp2"], c["desc2"], c["has2"], ur2, c["fv2"], sf,
                                            np.full(8, 1e9, np.float32), c["f12"], c["ex"], c["ey"], only_stereo, False)
    ref0 = oracle.search_for_triangulation(c["kp1"], c["desc1"], c["has1"], ur1, c["fv1"], c["kp2"], c["desc2"], c["has2"], ur2, c["fv2"], sf, sg, c["f12"],
                                           c["ex"], c["ey"], only_stereo, False)
    if forward:                      # the epipole is inside the image and its gate does fire: with the epipole moved away, more pairs
        far = oracle.search_for_triangulation(c["kp1"], c["desc1"], c["has1"], ur1, c["fv1"], c["kp2"], c["desc2"], c["has2"], ur2, c["fv2"], sf, sg, c["f12"],
                                              np.float32(-1e6), np.float32(-1e6), only_stereo, False)
        assert 0 <= c["ex"] < 752 and 0 <= c["ey"] < 480
        assert far[0] >= ref0[0] + (1 if mono else 0)
    if noise > 0:
        assert loose[0] > ref0[0]


def test_search_for_triangulation_edge_cases(ext, oracle):
    import gf_orb_slam2_amd as G
    import gf_cases
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), oracle.KEYPOINT_DTYPE)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    c = gf_cases.triangulation_case(oracle, kl[:300], dl[:300], np.random.default_rng(9))
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    sg = (sf * sf).astype(np.float32)
    M = G.ORBmatcher(0.6, True, extractor=ext)
    args = lambda **kw: [{**c, **kw}[k] for k in ("kp1", "desc1", "has1", "ur1", "fv1", "kp2", "desc2", "has2", "ur2", "fv2")]
    # every keypoint of one side already has a map point: nothing to do
    n, out = M.SearchForTriangulation(*args(has1=np.ones(300, np.uint8)), sf, sg, c["f12"], c["ex"], c["ey"])
    assert n == 0 and (out == -1).all()
    n, out = M.SearchForTriangulation(*args(has2=np.ones(300, np.uint8)), sf, sg, c["f12"], c["ex"], c["ey"])
    assert n == 0 and (out == -1).all()
    # a degenerate fundamental matrix: den == 0 everywhere, CheckDistEpipolarLine answers false (:262-263)
    n, out = M.SearchForTriangulation(*args(), sf, sg, np.zeros(9, np.float32), c["ex"], c["ey"])
    assert n == 0
    # no common node
    empty = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.uint32))
    n, out = M.SearchForTriangulation(*args(fv2=empty), sf, sg, c["f12"], c["ex"], c["ey"])
    assert n == 0
    # an octave past the scale table is refused, not read
    bad = c["kp2"].copy()
    bad["octave"][5] = 8
    with pytest.raises(G.GfoError):
        M.SearchForTriangulation(*args(kp2=bad), sf, sg, c["f12"], c["ex"], c["ey"])
