"""GPU parity at the other sizes BASELINE.json names and at awkward ones: 1920x1080 @4000 features,
a 50k-point local map, odd widths (unaligned rows), tiny images (empty levels), 1 level, small quotas,
re-planning one context across sizes, ragged batches.  Bit-exact vs the oracle."""
import numpy as np
import pytest

from conftest import synth_frame

pytestmark = pytest.mark.gpu


def _same(ext, oe, img):
    gk, gd = ext(img)
    ok, od = oe(img)
    assert len(gk) == len(ok)
    assert gk.tobytes() == ok.tobytes()
    np.testing.assert_array_equal(gd, od)
    return len(gk)


def test_1080p_4000_features(oracle):
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(4000, 1.2, 8, 20, 7, max_batch=2)
    oe = oracle.OracleExtractor(4000, 1.2, 8, 20, 7)
    n = _same(ext, oe, synth_frame(1920, 1080, 11))
    assert n >= 3500
    assert ext.mnFeaturesPerLevel.tolist() == [869, 724, 603, 503, 419, 349, 291, 242]
    ext.close()


@pytest.mark.parametrize("w,h", [(1241, 376), (640, 480), (333, 217), (130, 100), (64, 48), (37, 300)])
def test_awkward_sizes_one_context(oracle, w, h):
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(1000, 1.2, 8, 20, 7)
    oe = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    _same(ext, oe, synth_frame(w, h, w + h))
    _same(ext, oe, synth_frame(752, 480, 1))       # re-plan on a size change
    ext.close()


@pytest.mark.parametrize("nf,sf,nl,ini,mn", [(500, 1.2, 8, 20, 7), (3000, 1.1, 12, 12, 5), (50, 1.5, 4, 40, 10), (1000, 2.0, 3, 20, 7),
                                             (1200, 1.2, 1, 20, 7), (7, 1.2, 8, 20, 7)])
def test_extractor_parameters(oracle, nf, sf, nl, ini, mn):
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(nf, sf, nl, ini, mn)
    oe = oracle.OracleExtractor(nf, sf, nl, ini, mn)
    np.testing.assert_array_equal(ext.GetScaleFactors().view(np.uint32), oe.scale_factors.view(np.uint32))
    np.testing.assert_array_equal(ext.GetInverseScaleSigmaSquares().view(np.uint32), oe.inv_level_sigma2.view(np.uint32))
    np.testing.assert_array_equal(ext.mnFeaturesPerLevel, oe.features_per_level)
    _same(ext, oe, synth_frame(752, 480, 31))
    ext.close()


def test_flat_and_saturated_images(oracle):
    """no corners at all / only the minThFAST fallback fires / saturated blocks"""
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(1000, 1.2, 8, 20, 7)
    oe = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    assert _same(ext, oe, np.full((480, 752), 128, np.uint8)) == 0
    faint = (128 + 6 * ((np.indices((480, 752)).sum(0) // 23) % 2)).astype(np.uint8)      # contrast 12 < iniTh
    faint[::37, ::41] += 9
    _same(ext, oe, faint)
    blocks = np.where((np.indices((480, 752)) // 16).sum(0) % 2 == 0, 255, 0).astype(np.uint8)
    _same(ext, oe, blocks)
    assert ext(np.zeros((0, 0), np.uint8))[0].size == 0          # empty image: untouched outputs
    ext.close()


def test_ragged_batches_and_device_pitch(oracle):
    import torch
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(800, 1.2, 8, 20, 7, max_batch=2)
    oe = oracle.OracleExtractor(800, 1.2, 8, 20, 7)
    imgs = [synth_frame(501, 397, i) for i in range(5)]           # odd width, batch larger than max_batch
    kps, descs = ext.extract_batch(imgs)
    for im, k, d in zip(imgs, kps, descs):
        ok, od = oe(im)
        assert k.tobytes() == ok.tobytes() and (d == od).all()
    # device-resident input with pitch == width (unaligned rows) and a padded pitch
    for pitch in (501, 560):
        buf = np.zeros((3, 397, pitch), np.uint8)
        for i in range(3):
            buf[i, :, :501] = imgs[i]
        t = torch.from_numpy(buf).cuda()
        ext.set_stream(torch.cuda.current_stream().cuda_stream)
        ext.extract_batch_device(t.data_ptr(), 3, 501, 397, pitch=pitch, img_stride=pitch * 397)
        torch.cuda.synchronize()
        for i in range(3):
            k, d = ext.batch_fetch(i)
            ok, od = oe(imgs[i])
            assert k.tobytes() == ok.tobytes() and (d == od).all()
        ext.set_stream(0)
    ext.close()


def test_pyramid_with_reference_border(oracle, euroc_l):
    """mvImagePyramid with its 19-px BORDER_REFLECT_101 frame (ORBextractor.cc:1182-1197)"""
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(2000, 1.2, 8, 20, 7)
    oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    ext.ComputePyramid(euroc_l)
    oe.compute_pyramid(euroc_l)
    for l in range(8):
        np.testing.assert_array_equal(ext.pyramid_level(l, border=19), oe.level(l, padded=True))
        np.testing.assert_array_equal(ext.pyramid_level(l), oe.level(l))
    ext.close()


def test_projection_against_50k_map(oracle):
    """config 4: 1920x1080 @4000 features, SearchByProjection against a 50 000-point synthetic local map"""
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(4000, 1.2, 8, 20, 7)
    img = synth_frame(1920, 1080, 3)
    kp, desc = ext(img)
    rng = np.random.default_rng(7)
    n, m = len(kp), 50000
    mps = np.zeros(m, oracle.MAP_POINT_DTYPE)
    mpd = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    nv = min(n, 3500)
    vis = rng.choice(m, nv, replace=False)
    src = rng.choice(n, nv, replace=False)
    d = desc[src].copy()
    for j in range(60):
        sel = rng.random(nv) < rng.random(nv)
        bits = rng.integers(0, 256, nv)
        d[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    mpd[vis] = d
    mps["proj_x"] = rng.uniform(0, 1920, m); mps["proj_y"] = rng.uniform(0, 1080, m)
    mps["proj_x"][vis] = kp["x"][src] + rng.normal(0, 2, nv)
    mps["proj_y"][vis] = kp["y"][src] + rng.normal(0, 2, nv)
    mps["level"] = rng.integers(0, 8, m); mps["level"][vis] = kp["octave"][src]
    mps["proj_xr"] = mps["proj_x"] - 10
    mps["view_cos"] = 1.0
    mps["flags"] = 5
    bounds = (0.0, 0.0, 1920.0, 1080.0)
    sf = ext.GetScaleFactors()
    ref = oracle.search_by_projection(kp, desc, None, sf, bounds, mps, mpd, 3.0, 0.8, None)
    got = G.ORBmatcher(0.8, True, extractor=ext).SearchByProjection(kp, desc, None, sf, bounds, mps, mpd, 3.0, None)
    assert got[0] == ref[0] and ref[0] > 1500
    np.testing.assert_array_equal(got[1], ref[1])
    np.testing.assert_array_equal(got[2], ref[2])
    ext.close()


@pytest.mark.parametrize("w,h,sf,nl,lds_kb,threads", [(752, 480, 1.2, 8, 64, 1024), (752, 480, 1.2, 8, 16, 256), (752, 480, 1.2, 8, 150, 512),
                                                       (1241, 376, 1.2, 8, 64, 1024), (333, 217, 1.2, 8, 8, 1024), (1920, 1080, 1.2, 8, 64, 1024),
                                                       (640, 480, 1.1, 12, 32, 1024), (640, 480, 2.0, 3, 64, 1024), (640, 480, 1.5, 4, 24, 512),
                                                       (130, 100, 1.2, 8, 64, 1024), (37, 300, 1.2, 6, 4, 1024)])
def test_banded_pyramid_equals_oracle(oracle, monkeypatch, w, h, sf, nl, lds_kb, threads):
    """The one-launch banded pyramid (every level through LDS, the path large batches take) forced on a single
    image: every level bit-exact against the oracle's cv::resize chain, then the whole extraction."""
    import gf_orb_slam2_amd as G
    monkeypatch.setenv("GFO_PYR_BAND_MIN_WG", "1")
    monkeypatch.setenv("GFO_PYR_MAX_W", "100000")          # also the wide images that default to the per-level path
    monkeypatch.setenv("GFO_PYR_MAX_OVERHEAD", "100")
    monkeypatch.setenv("GFO_PYR_GROUP", str(2 + (w + lds_kb) % 3))   # groups of up to 2, 3 or 4 levels
    monkeypatch.setenv("GFO_PYR_LDS_KB", str(lds_kb))
    monkeypatch.setenv("GFO_PYR_THREADS", str(threads))
    img = synth_frame(w, h, 3 * w + h)
    ext = G.ORBextractor(800, sf, nl, 20, 7)
    oe = oracle.OracleExtractor(800, sf, nl, 20, 7)
    ext.ComputePyramid(img)
    oe.compute_pyramid(img)
    for l in range(nl):
        np.testing.assert_array_equal(ext.pyramid_level(l), oe.level(l), err_msg=f"level {l}")
    _same(ext, oe, img)
    ext.close()


@pytest.mark.parametrize("w,h,nimg", [(752, 480, 1), (333, 217, 1), (752, 480, 34), (640, 480, 33)])
def test_per_level_pyramid_path(oracle, monkeypatch, w, h, nimg):
    """The fallback the banded pyramid replaces (one launch per level, the small top levels fused into one
    workgroup per image for batches of 32 and more) stays bit-exact: forced by asking for more band workgroups
    than any launch has."""
    import gf_orb_slam2_amd as G
    monkeypatch.setenv("GFO_PYR_BAND_MIN_WG", "100000000")
    imgs = [synth_frame(w, h, 7 * w + i) for i in range(min(nimg, 3))]
    batch = [imgs[i % len(imgs)] for i in range(nimg)]
    ext = G.ORBextractor(900, 1.2, 8, 20, 7, max_batch=nimg)
    oe = oracle.OracleExtractor(900, 1.2, 8, 20, 7)
    kps, descs = ext.extract_batch(batch)
    ref = [oe(im) for im in imgs]
    for i in range(nimg):
        ok, od = ref[i % len(imgs)]
        assert kps[i].tobytes() == ok.tobytes() and (descs[i] == od).all()
    ext.ComputePyramid(imgs[0])
    oe.compute_pyramid(imgs[0])
    for l in range(8):
        np.testing.assert_array_equal(ext.pyramid_level(l), oe.level(l))
    ext.close()


def test_maximum_image_size(oracle):
    """The largest image the 12-bit coordinate packing admits (4000 px a side): 4000x3000, 5000 features, bit-exact;
    one pixel more is refused with GFO_ERR_INVALID instead of being mis-packed."""
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd._lib import GfoError
    base = synth_frame(1000, 750, 99)                  # tiled 4 x 4 (generating 12 MP of value noise takes 40 s)
    img = np.ascontiguousarray(np.tile(base, (4, 4)))
    assert img.shape == (3000, 4000)
    ext = G.ORBextractor(5000, 1.2, 8, 20, 7)
    oe = oracle.OracleExtractor(5000, 1.2, 8, 20, 7)
    n = _same(ext, oe, img)
    assert n >= 4000
    with pytest.raises(GfoError):
        ext(np.zeros((100, 4001), np.uint8))
    _same(ext, oe, synth_frame(752, 480, 5))      # the context is still usable after the refusal
    ext.close()


@pytest.mark.parametrize("nf,nl,w,h", [(4000, 1, 752, 480), (6000, 2, 1241, 376), (6000, 3, 640, 480)])
def test_large_level_quota_runs_from_global_memory(oracle, nf, nl, w, h):
    """More than 2040 features on one level: the level's quadtree tables no longer fit the 160 KB of LDS and the
    global-memory variant of the kernel takes over -- slower, same result."""
    import gf_orb_slam2_amd as G
    img = synth_frame(w, h, nf + nl)
    ext = G.ORBextractor(nf, 1.2, nl, 20, 7)
    assert ext.mnFeaturesPerLevel.max() > 2040
    n = _same(ext, oracle.OracleExtractor(nf, 1.2, nl, 20, 7), img)
    assert n > 500
    _same(ext, oracle.OracleExtractor(nf, 1.2, nl, 20, 7), synth_frame(w, h, 77))
    ext.close()


@pytest.mark.parametrize("qcap", [None, 512, 264])
def test_fast_queue_overflow_paths(oracle, monkeypatch, qcap):
    """k_fast keeps fewer queue entries than a cell has pixels (768, or GFO_FAST_QCAP read when the arena is planned): a cell
    of noise-like imagery must score what it has queued before it queues more (flush), and drop the corner list for the
    dense suppression pass when even its corners do not fit.  White noise at thresholds 1 / 1 passes most pixels through the
    compass test; saturated blocks add plateaus of equal scores.  A queue of 512 entries forces the flush in most cells, one of
    264 (a single pass of 256 pixels) the dense pass in nearly all of them."""
    import gf_orb_slam2_amd as G
    if qcap:
        monkeypatch.setenv("GFO_FAST_QCAP", str(qcap))
    rng = np.random.default_rng(123)
    noise = rng.integers(0, 256, (300, 420), dtype=np.uint8)
    blocks = noise.copy()
    blocks[40:120, 60:200] = 255
    blocks[150:260, 220:400] = rng.integers(120, 136, (110, 180), dtype=np.uint8)
    for ini, mn in ((1, 1), (7, 2), (40, 3)):
        ext = G.ORBextractor(3000, 1.2, 6, ini, mn)
        oe = oracle.OracleExtractor(3000, 1.2, 6, ini, mn)
        assert _same(ext, oe, noise) > 0
        _same(ext, oe, blocks)
        for l in range(6):
            oc = sorted(map(tuple, oe.level_candidates(l).tolist()))
            gc = sorted(map(tuple, ext.debug_level_candidates(l).tolist()))
            assert gc == oc, f"FAST candidates differ at level {l} (thresholds {ini}/{mn}, queue {qcap})"
        ext.close()


@pytest.mark.parametrize("groups", ["auto", "2,5", "1,99", "0"])
def test_quadtree_level_groups_and_key_overflow_to_l2(oracle, monkeypatch, groups):
    """k_quadtree launched as level ranges with their own node / key capacity (GFO_QT_GROUPS, opt-in: measured, not the default,
    profiles/quadtree_occupancy_r05.txt) and the batch default of 7 x quota LDS keys: a batch of 20 images (the grouped forms apply
    above GFO_FEW_MAX = 16 images) among them noise frames whose level-0 candidate count exceeds the LDS key capacity (those workgroups run on
    their keys in L2) -- every image equals the oracle."""
    import gf_orb_slam2_amd as G
    monkeypatch.setenv("GFO_QT_GROUPS", groups)
    rng = np.random.default_rng(3)
    imgs = [synth_frame(752, 480, 60 + i) for i in range(18)]
    noisy = synth_frame(752, 480, 7).astype(np.int32) + rng.integers(-60, 60, (480, 752))      # many more FAST candidates per level
    imgs += [np.clip(noisy, 0, 255).astype(np.uint8), rng.integers(0, 256, (480, 752), dtype=np.uint8)]
    ext = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=len(imgs))
    oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    ks, ds = ext.extract_batch(imgs)
    most = 0
    for i, img in enumerate(imgs):
        ok, od = oe(img)
        most = max(most, max(len(oe.level_candidates(l)) for l in range(8)))
        assert ks[i].tobytes() == ok.tobytes(), (groups, i)
        assert ds[i].tobytes() == od.tobytes(), (groups, i)
    assert most > 3072          # the L2-key path was taken by at least one (image, level)
    ext.close()


def _crowded_frame(oracle, rng, n, w, h, nproto=3000):
    """n keypoints over a w x h image, descriptors around prototypes (small distances and ties are common)"""
    kp = np.zeros(n, oracle.KEYPOINT_DTYPE)
    kp["x"] = rng.uniform(0, w, n).astype(np.float32)
    kp["y"] = rng.uniform(0, h, n).astype(np.float32)
    kp["octave"] = rng.choice(8, n, p=[0.3, 0.2, 0.15, 0.1, 0.1, 0.06, 0.05, 0.04])
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kp["size"] = 31.0; kp["response"] = 40; kp["class_id"] = -1
    protos = rng.integers(0, 256, (nproto, 32), dtype=np.uint8)
    desc = protos[rng.integers(0, nproto, n)].copy()
    for _ in range(6):
        sel = rng.random(n) < 0.5
        bits = rng.integers(0, 256, n)
        desc[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    return kp, desc


def test_matchers_at_the_keypoint_limit(oracle):
    """65 535 keypoints in a frame -- the most the 16-bit keypoint index of the candidate keys allows (include/gfo.h) -- through the
    matchers that carry that index: SearchByProjection against 20 000 map points, SearchForInitialization (a 30-px window: ~20 candidates
    a keypoint, a table of several hundred thousand entries), SearchForTriangulation and SearchByBoW(KF, KF) over 4096 nodes."""
    import gf_orb_slam2_amd as G
    import gf_cases
    rng = np.random.default_rng(65535)
    n, w, h = 65535, 1920.0, 1080.0
    kp, desc = _crowded_frame(oracle, rng, n, w, h)
    bounds = (0.0, 0.0, w, h)
    ext = G.ORBextractor(2000, 1.2, 8, 20, 7)
    sf = np.asarray(ext.GetScaleFactors(), np.float32)
    try:
        # SearchByProjection(F, MapPoints)
        m = 20000
        src = rng.integers(0, n, m)
        mps = np.zeros(m, oracle.MAP_POINT_DTYPE)
        mps["proj_x"] = kp["x"][src] + rng.normal(0, 1.5, m); mps["proj_y"] = kp["y"][src] + rng.normal(0, 1.5, m)
        mps["proj_xr"] = mps["proj_x"] - 10
        mps["level"] = kp["octave"][src]; mps["view_cos"] = 1.0; mps["flags"] = 5
        mpd = desc[src].copy()
        bits = rng.integers(0, 256, m)
        mpd[np.arange(m), bits >> 3] ^= (1 << (bits & 7)).astype(np.uint8)
        ref = oracle.search_by_projection(kp, desc, None, sf, bounds, mps, mpd, 1.0, 0.8, None)
        got = G.ORBmatcher(0.8, True, extractor=ext).SearchByProjection(kp, desc, None, sf, bounds, mps, mpd, 1.0, None)
        assert got[0] == ref[0] and ref[0] > 8000, (got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
        np.testing.assert_array_equal(got[2], ref[2])
        assert np.flatnonzero(ref[1] >= 0).max() > 65000    # keypoints at the top of the index range are matched
        # SearchForInitialization
        kp2, d2, prev = gf_cases.initialization_case(oracle, kp, desc, rng, flips=6, sigma=6.0)
        p_ref, p_got = prev.copy(), prev.copy()
        ref = oracle.search_for_initialization(kp, desc, p_ref, kp2, d2, bounds, 30, 0.9, True)
        got = G.ORBmatcher(0.9, True, extractor=ext).SearchForInitialization(kp, desc, p_got, kp2, d2, bounds, 30)
        assert got[0] == ref[0] and ref[0] > 5000, (got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
        assert p_got.tobytes() == p_ref.tobytes()
        # SearchForTriangulation / SearchByBoW(KF, KF): 4096 nodes of ~16 keypoints
        c = gf_cases.triangulation_case(oracle, kp, desc, rng, flips=6, noise=0.5, node_shift=0, fx=1000.0, fy=1000.0, cx=960.0, cy=540.0)
        node1 = (desc[:, 0].astype(np.int64) << 4) | (desc[:, 1] >> 4)
        node2 = (c["desc2"][:, 0].astype(np.int64) << 4) | (c["desc2"][:, 1] >> 4)
        fv1, fv2 = oracle.make_feature_vector(node1), oracle.make_feature_vector(node2)
        sg = (sf * sf).astype(np.float32)
        a = (kp, desc, c["has1"], c["ur1"], fv1, c["kp2"], c["desc2"], c["has2"], c["ur2"], fv2, sf, sg, c["f12"], c["ex"], c["ey"])
        ref = oracle.search_for_triangulation(*a, False, True)
        got = G.ORBmatcher(0.6, True, extractor=ext).SearchForTriangulation(*a, False)
        assert got[0] == ref[0] and ref[0] > 5000, (got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
        v1, v2 = (1 - c["has1"]).astype(np.uint8), (1 - c["has2"]).astype(np.uint8)
        ref = oracle.search_by_bow_keyframes(desc, kp["angle"], v1, fv1, c["desc2"], c["kp2"]["angle"], v2, fv2, 0.75, True)
        got = G.ORBmatcher(0.75, True, extractor=ext).SearchByBoWKeyFrames(desc, kp["angle"], v1, fv1, c["desc2"], c["kp2"]["angle"], v2, fv2)
        assert got[0] == ref[0] and ref[0] > 5000, (got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
    finally:
        ext.close()
