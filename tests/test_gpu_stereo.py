"""GPU parity tests of the stereo association (Frame::ComputeStereoMatches_Undistorted) and of
the device-chained extract -> match path, through the C ABI, vs the CPU oracle.  Index and
integer outputs bit-exact; float outputs compared by bit pattern."""
import numpy as np
import pytest

from conftest import synth_frame

pytestmark = pytest.mark.gpu

FX, BF = 435.2046959714599, 47.90639384423901   # EuRoC.yaml of ORB-SLAM2 (Camera.fx, Camera.bf)


@pytest.fixture(autouse=True, params=["default", "rows", "keypoints"])
def association_form(request, monkeypatch):
    """Every test of this module runs three times: the library's own choice (row form from four pairs on), the row
    form forced (GFO_STEREO_ROWS=5: k_stereo_match_rows also for a single pair, with a band height that does not divide
    the image) and the per-keypoint form forced (GFO_STEREO_ROWS=0: k_stereo_match)."""
    if request.param == "rows":
        monkeypatch.setenv("GFO_STEREO_ROWS", "5")
    elif request.param == "keypoints":
        monkeypatch.setenv("GFO_STEREO_ROWS", "0")
    return request.param


@pytest.fixture(scope="module")
def ext():
    import gf_orb_slam2_amd as G
    e = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=4)
    yield e
    e.close()


def _params(rows=480):
    import gf_orb_slam2_amd as G
    return G.StereoParams(rows, BF, BF / FX, 0.0)


def _cmp_stereo(got, ref):
    assert got[0] == ref[0], f"nmatched {got[0]} vs {ref[0]}"
    for name, a, b in zip(("u_right", "depth", "best_dist", "best_idx"), got[1:], ref[1:]):
        assert a.tobytes() == b.tobytes(), name


def test_stereo_host_arrays_euroc(ext, oracle, euroc_l, euroc_r):
    import gf_orb_slam2_amd as G
    oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    kl, dl = oe(euroc_l)
    kr, dr = oe(euroc_r)
    sf = oe.scale_factors
    p = _params()
    m = G.ORBmatcher(0.8, True, extractor=ext)
    got = m.ComputeStereoMatches(kl, dl, kr, dr, sf, p)
    ref = oracle.stereo_match(kl, dl, kr, dr, sf, p.n_rows, p.mbf, p.mb, p.min_x)
    _cmp_stereo(got, ref)
    assert (ref[1] > 0).sum() > 100       # the (unrectified) EuRoC pair still matches


def test_stereo_disparity_windows(ext, oracle, euroc_l, euroc_r):
    """per-keypoint [minD, maxD] windows (the map-point branch, Frame.cc:1220-1231)"""
    import gf_orb_slam2_amd as G
    oe = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    kl, dl = oe(euroc_l)
    kr, dr = oe(euroc_r)
    rng = np.random.default_rng(3)
    min_d = rng.uniform(0, 30, len(kl)).astype(np.float32)
    max_d = (min_d + rng.uniform(5, 80, len(kl))).astype(np.float32)
    p = _params()
    m = G.ORBmatcher(0.8, True, extractor=ext)
    got = m.ComputeStereoMatches(kl, dl, kr, dr, oe.scale_factors, p, min_d, max_d)
    ref = oracle.stereo_match(kl, dl, kr, dr, oe.scale_factors, p.n_rows, p.mbf, p.mb, p.min_x, min_d, max_d)
    _cmp_stereo(got, ref)


def test_stereo_edge_cases(ext, oracle):
    import gf_orb_slam2_amd as G
    m = G.ORBmatcher(0.8, True, extractor=ext)
    sf = np.array([1.0, 1.2, 1.44], np.float32)
    p = _params()
    kd = oracle.KEYPOINT_DTYPE
    # empty left / empty right
    z = np.zeros(0, kd); zd = np.zeros((0, 32), np.uint8)
    one = np.zeros(1, kd); one["x"] = 100; one["y"] = 50
    od = np.zeros((1, 32), np.uint8)
    assert m.ComputeStereoMatches(z, zd, one, od, sf, p)[0] == 0
    got = m.ComputeStereoMatches(one, od, z, zd, sf, p)
    ref = oracle.stereo_match(one, od, z, zd, sf, p.n_rows, p.mbf, p.mb, p.min_x)
    _cmp_stereo(got, ref)
    # identical descriptors, ties on distance: the lowest right index must win; rows out of range skipped
    rng = np.random.default_rng(11)
    n = 300
    kl = np.zeros(n, kd); kr = np.zeros(n, kd)
    kl["x"] = rng.uniform(100, 700, n).astype(np.float32); kl["y"] = rng.uniform(-5, 490, n).astype(np.float32)
    kl["octave"] = rng.integers(0, 3, n)
    kr["x"] = kl["x"] - rng.uniform(-3, 60, n).astype(np.float32); kr["y"] = kl["y"] + rng.uniform(-2, 2, n).astype(np.float32)
    kr["octave"] = rng.integers(0, 3, n)
    base = rng.integers(0, 256, (8, 32), dtype=np.uint8)
    dl = base[rng.integers(0, 8, n)]
    dr = base[rng.integers(0, 8, n)]
    got = m.ComputeStereoMatches(kl, dl, kr, dr, sf, p)
    ref = oracle.stereo_match(kl, dl, kr, dr, sf, p.n_rows, p.mbf, p.mb, p.min_x)
    _cmp_stereo(got, ref)


def test_right_keypoints_whose_row_band_misses_the_image(ext, oracle):
    """ADVICE r2 (k_stereo.hip row bands): host arrays may carry right keypoints far above or below the image.  The
    reference's row loop does not run for them (Frame.h:248-256: maxr < 0, or minr > nRows - 1), so they are nobody's
    candidate -- packed as min | max << 16 a negative maxr used to read as 65535 = "every row"."""
    import gf_orb_slam2_amd as G
    m = G.ORBmatcher(0.8, True, extractor=ext)
    sf = np.array([1.0, 1.2, 1.44], np.float32)
    p = _params()
    kd = oracle.KEYPOINT_DTYPE
    rng = np.random.default_rng(23)
    n = 64
    kl = np.zeros(n, kd); kr = np.zeros(3 * n, kd)
    kl["x"] = rng.uniform(200, 700, n).astype(np.float32); kl["y"] = rng.uniform(5, 470, n).astype(np.float32)
    dl = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    # three right copies of every left keypoint with the SAME descriptor and a valid disparity: one far above the image,
    # one far below it, one in place -- only the last may ever be matched
    for j, dy in enumerate((-700.0, +900.0, 0.5)):
        kr["x"][j * n:(j + 1) * n] = kl["x"] - 20.0
        kr["y"][j * n:(j + 1) * n] = kl["y"] + dy
    dr = np.tile(dl, (3, 1))
    got = m.ComputeStereoMatches(kl, dl, kr, dr, sf, p)
    ref = oracle.stereo_match(kl, dl, kr, dr, sf, p.n_rows, p.mbf, p.mb, p.min_x)
    _cmp_stereo(got, ref)
    assert (ref[4][ref[4] >= 0] >= 2 * n).all() and (ref[4] >= 2 * n).sum() > n // 2     # every accepted match is an in-place copy


def test_chained_extract_and_match_on_device(ext, oracle, euroc_l, euroc_r):
    import torch
    import gf_orb_slam2_amd as G
    sl = synth_frame(752, 480, 7)
    sr = np.roll(sl, -9, axis=1)
    imgs = np.stack([euroc_l, euroc_r, sl, sr])
    t = torch.from_numpy(imgs).cuda()
    ext.set_stream(torch.cuda.current_stream().cuda_stream)
    ext.extract_batch_device(t.data_ptr(), 4, 752, 480)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    p = _params()
    m.stereo_match_batch(p)
    torch.cuda.synchronize()
    for pair in range(2):
        oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
        kl, dl = oe(imgs[2 * pair]); kr, dr = oe(imgs[2 * pair + 1])
        gkl, gdl = ext.batch_fetch(2 * pair)
        assert gkl.tobytes() == kl.tobytes() and (gdl == dl).all()
        ref = oracle.stereo_match(kl, dl, kr, dr, oe.scale_factors, p.n_rows, p.mbf, p.mb, p.min_x)
        nm, u, dp, bd, bi = m.stereo_fetch(pair, len(kl))
        _cmp_stereo((nm, u, dp, bd, bi), ref)
    ext.set_stream(0)


def test_sad_subpixel_variant(ext, oracle, euroc_l, euroc_r):
    """Frame::ComputeStereoMatches, the variant behind ALTER_STEREO_MATCHING (Frame.cc:889-1078)"""
    import torch
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.synth import synth_stereo_pair
    sl, sr = synth_stereo_pair(752, 480, 3)
    imgs = np.stack([sl, sr, euroc_l, euroc_r])
    t = torch.from_numpy(imgs).cuda()
    ext.set_stream(torch.cuda.current_stream().cuda_stream)
    ext.extract_batch_device(t.data_ptr(), 4, 752, 480)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    m.stereo_match_sad_batch(BF, BF / FX)
    torch.cuda.synchronize()
    for pair in range(2):
        el = oracle.OracleExtractor(2000, 1.2, 8, 20, 7); er = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
        kl, dl = el(imgs[2 * pair]); kr, dr = er(imgs[2 * pair + 1])
        n, u, dp, bd = oracle.stereo_match_sad(el, er, kl, dl, kr, dr, BF, BF / FX)
        gn, gu, gdp, gbd, gbi = m.stereo_fetch(pair, len(kl))
        assert gn == n
        assert gu.tobytes() == u.tobytes() and gdp.tobytes() == dp.tobytes()
        np.testing.assert_array_equal(gbd, bd)
        assert n > (500 if pair == 0 else 50)
    ext.set_stream(0)


def test_host_keypoints_with_bad_octaves_are_refused(oracle):
    """the kernels index mvScaleFactors[octave] as Frame.h:244 does: a host keypoint outside the pyramid is an error,
    not an out-of-bounds read; overlapping device images are refused too"""
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(500, 1.2, 8, 20, 7)
    kd = oracle.KEYPOINT_DTYPE
    kl = np.zeros(4, kd); kl["x"] = [10, 20, 30, 40]; kl["y"] = 12
    kr = kl.copy()
    d = np.zeros((4, 32), np.uint8)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    sf = ext.GetScaleFactors()
    prm = G.StereoParams(480, 47.9, 0.11, 0.0)
    m.ComputeStereoMatches(kl, d, kr, d, sf, prm)
    for bad in (-1, 8, 1 << 20):
        k2 = kl.copy(); k2["octave"][2] = bad
        for args in ((k2, d, kr, d), (kl, d, k2, d)):
            with pytest.raises(G.GfoError) as e:
                m.ComputeStereoMatches(*args, sf, prm)
            assert e.value.code == -1
    import torch
    t = torch.zeros(2 * 240 * 320, dtype=torch.uint8, device="cuda")
    with pytest.raises(G.GfoError) as e:
        ext.extract_batch_device(t.data_ptr(), 2, 320, 240, pitch=320, img_stride=320 * 100)
    assert e.value.code == -1
    ext.close()


def test_stereo_frame_in_one_submission(oracle, euroc_l, euroc_r):
    """gfo_extract_stereo = ExtractORB x2 + ComputeStereoMatches_Undistorted (Frame.cc:84-100) as one H2D, one replay
    of the captured launch sequence, one D2H, one synchronisation: same bits as the oracle, on the first call (which
    captures the graph), on replays, after a change of image size (re-plan, new graph) and with other stereo
    parameters (a different graph key)."""
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=2)
    oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    sf = oe.scale_factors

    def check(l, r, prm):
        okl, odl = oe(l)
        okr, odr = oe(r)
        ref = oracle.stereo_match(okl, odl, okr, odr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
        kl, dl, kr, dr, nm, u, dp, bd, bi = ext.extract_stereo(l, r, prm)
        assert kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes()
        assert (dl == odl).all() and (dr == odr).all()
        assert nm == ref[0]
        for a, b in zip((u, dp, bd, bi), ref[1:]):
            assert a.tobytes() == b.tobytes()
        return nm

    p1 = G.StereoParams(480, BF, BF / FX, 0.0)
    for _ in range(3):                                   # capture, then two replays
        assert check(euroc_l, euroc_r, p1) > 500
    crop_l, crop_r = np.ascontiguousarray(euroc_l[40:400, 100:700]), np.ascontiguousarray(euroc_r[40:400, 100:700])
    check(crop_l, crop_r, G.StereoParams(360, BF, BF / FX, 0.0))         # other size: arena re-planned, graph rebuilt
    check(crop_l, crop_r, G.StereoParams(360, 30.0, 0.2, 0.0))           # other calibration: graph key differs
    check(euroc_l, euroc_r, p1)
    # the plain batch path shares the pinned staging / single-sync code
    (k0, k1), (d0, d1) = ext.extract_batch([euroc_l, euroc_r])
    ok, od = oe(euroc_l)
    assert k0.tobytes() == ok.tobytes() and (d0 == od).all()
    ext.close()


def test_chained_contexts_alternate_stereo_frames(oracle, euroc_l, euroc_r):
    """Two contexts alternating stereo frames as a pipelined application does, chained behind each other's pyramid
    (gfo_ctx_chain): every frame equals the unchained result, whichever context served it."""
    import gf_orb_slam2_amd as G
    prm = G.StereoParams(480, 47.90639384423901, 47.90639384423901 / 435.2046959714599, 0.0)
    exts = [G.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=2) for _ in range(3)]
    ms = [G.ORBmatcher(0.8, True, extractor=e) for e in exts]
    (rkl, rkr), (rdl, rdr) = exts[2].extract_batch([euroc_l, euroc_r])     # unchained reference run
    ms[2].stereo_match_batch(prm)
    ref = ms[2].stereo_fetch(0, max(len(rkl), 1))
    exts[0].chain_after(exts[1], exts[0].STAGE_PYRAMID)
    exts[1].chain_after(exts[0], exts[1].STAGE_PYRAMID)
    for it in range(6):
        k = it & 1
        (kl, kr), (dl, dr) = exts[k].extract_batch([euroc_l, euroc_r])
        ms[k].stereo_match_batch(prm)
        got = ms[k].stereo_fetch(0, max(len(kl), 1))
        assert kl.tobytes() == rkl.tobytes() and kr.tobytes() == rkr.tobytes() and (dl == rdl).all() and (dr == rdr).all()
        assert got[0] == ref[0]
        for a, b in zip(got[1:], ref[1:]):
            assert a.tobytes() == b.tobytes()
    for e in exts:
        e.close()


@pytest.mark.parametrize("nfeat,w,h", [(4000, 1241, 376), (3400, 1241, 376), (8000, 1920, 1080)])
def test_stereo_frame_with_more_keypoints_than_the_lds_table_holds(oracle, nfeat, w, h):
    """The per-frame association without a bucketing launch (k_stereo_match_direct) keeps every right keypoint of the pair in LDS
    (16 bytes each): frames whose keypoint stride exceeds what 64 KB hold (3584) must take the bucketed kernels instead -- 4000
    features on a KITTI-sized frame and 8000 at 1080p; 3400 stays on the direct form.  One submission, bit for bit as the oracle."""
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.synth import synth_stereo_pair
    l, r = synth_stereo_pair(w, h, 5)
    ext = G.ORBextractor(nfeat, 1.2, 8, 20, 7, max_batch=2)
    oe = oracle.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    prm = G.StereoParams(h, BF, BF / FX, 0.0)
    okl, odl = oe(l)
    okr, odr = oe(r)
    ref = oracle.stereo_match(okl, odl, okr, odr, oe.scale_factors, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
    for _ in range(2):
        kl, dl, kr, dr, nm, u, dp, bd, bi = ext.extract_stereo(l, r, prm)
        assert kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes() and (dl == odl).all() and (dr == odr).all()
        assert nm == ref[0] and len(kl) > 0.8 * nfeat
        for a, b in zip((u, dp, bd, bi), ref[1:]):
            assert a.tobytes() == b.tobytes()
    ext.close()
