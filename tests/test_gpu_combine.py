"""GPU parity tests of the frame combiner (gfo_ctx_set_combining, csrc/gfo_combine.hip): per-frame calls of K host
threads through K contexts are executed as shared device batches -- the reference's own call pattern, one frame per
call (Frame.cc:84-100, ORBextractor.cc:1112-1174), from many threads.  Every result must equal the oracle's, bit for
bit, whichever batch a frame lands in and whoever it shares it with."""
import threading

import numpy as np
import pytest

from conftest import synth_frame

pytestmark = pytest.mark.gpu

FX, BF = 435.2046959714599, 47.90639384423901


def _run_threads(fns):
    errors = []

    def wrap(i, fn):
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e!r}")

    ts = [threading.Thread(target=wrap, args=(i, f)) for i, f in enumerate(fns)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:4]


def test_eight_stereo_streams_share_device_batches(oracle, euroc_l, euroc_r):
    """K = 8 camera streams, one thread and one combining context each, 12 stereo frames per stream; the streams carry
    DIFFERENT images (EuRoC pair, its swap, synthetic pairs), so a frame delivered to the wrong caller or assembled from
    the wrong slot of a batch cannot pass.  Also: the engine really batched (fewer device batches than requests) and no
    context or arena was created after the first frames."""
    import gf_orb_slam2_amd as G
    K, REPS = 8, 12
    pairs = [(euroc_l, euroc_r), (euroc_r, euroc_l)] + [(synth_frame(752, 480, 10 + i), synth_frame(752, 480, 30 + i)) for i in range(2)]
    oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    prm = G.StereoParams(480, BF, BF / FX, 0.0)
    refs = []
    for l, r in pairs:
        okl, odl = oe(l)
        okr, odr = oe(r)
        refs.append((okl, odl, okr, odr) + tuple(oracle.stereo_match(okl, odl, okr, odr, oe.scale_factors, prm.n_rows, prm.mbf, prm.mb, prm.min_x)))
    exts = [G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=2, combining=True) for _ in range(K)]
    L = exts[0]._L
    bad = []

    def frame(k):
        l, r = pairs[k % len(pairs)]
        okl, odl, okr, odr, nm, u, dp, bd, bi = refs[k % len(pairs)]
        kl, dl, kr, dr, gnm, gu, gdp, gbd, gbi = exts[k].extract_stereo(l, r, prm)
        if not (kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes() and (dl == odl).all() and (dr == odr).all() and
                gnm == nm and gu.tobytes() == u.tobytes() and gdp.tobytes() == dp.tobytes() and gbd.tobytes() == bd.tobytes() and
                gbi.tobytes() == bi.tobytes()):
            bad.append(k)

    _run_threads([lambda k=k: frame(k) for k in range(K)])          # first frames: engine, slots, arenas
    created, planned = L.gfo_contexts_created(), L.gfo_arenas_planned()
    b0, r0 = exts[0].combiner_stats()
    _run_threads([lambda k=k: [frame(k) for _ in range(REPS)] for k in range(K)])
    b1, r1 = exts[0].combiner_stats()
    assert not bad, f"streams with a wrong result: {sorted(set(bad))}"
    assert r1 - r0 == K * REPS
    assert b1 - b0 < K * REPS, "eight concurrent streams never shared a batch"
    # steady state creates nothing (a slot the first frames did not reach may still be prepared once: <= 8 slots in all)
    assert L.gfo_contexts_created() - created <= 8 and L.gfo_arenas_planned() - planned <= 8
    created, planned = L.gfo_contexts_created(), L.gfo_arenas_planned()
    _run_threads([lambda k=k: [frame(k) for _ in range(4)] for k in range(K)])
    assert not bad
    assert (L.gfo_contexts_created(), L.gfo_arenas_planned()) == (created, planned)
    for e in exts:
        e.close()


def test_left_right_extractors_combine_single_images(oracle, euroc_l, euroc_r):
    """The adapter's pattern: the left and right ORBextractor of a stereo rig on two threads (Frame.cc:84-87), here four
    rigs at once, gfo_extract of ONE image per call -- combined requests of one image each -- then the host-array
    association on the left extractor's context."""
    import gf_orb_slam2_amd as G
    RIGS, REPS = 4, 6
    oe = oracle.OracleExtractor(1500, 1.2, 8, 20, 7)
    imgs = [euroc_l, euroc_r, synth_frame(752, 480, 3), synth_frame(752, 480, 4)]
    refs = [oe(im) for im in imgs]
    exts = [G.ORBextractor(1500, 1.2, 8, 20, 7, combining=True) for _ in range(2 * RIGS)]
    bad = []

    def cam(i):
        for _ in range(REPS):
            k, d = exts[i](imgs[i % 4])
            if k.tobytes() != refs[i % 4][0].tobytes() or not (d == refs[i % 4][1]).all():
                bad.append(i)

    _run_threads([lambda i=i: cam(i) for i in range(2 * RIGS)])
    assert not bad, sorted(set(bad))
    b, r = exts[0].combiner_stats()
    assert r == 2 * RIGS * REPS and b < r
    # the association on host arrays runs on the extractor's own context, combining or not
    m = G.ORBmatcher(0.8, True, extractor=exts[0])
    prm = G.StereoParams(480, BF, BF / FX, 0.0)
    got = m.ComputeStereoMatches(refs[0][0], refs[0][1], refs[1][0], refs[1][1], oe.scale_factors, prm)
    ref = oracle.stereo_match(refs[0][0], refs[0][1], refs[1][0], refs[1][1], oe.scale_factors, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
    assert got[0] == ref[0] and all(a.tobytes() == b_.tobytes() for a, b_ in zip(got[1:], ref[1:]))
    for e in exts:
        e.close()


def test_combining_context_alone_and_mixed_shapes(oracle, euroc_l, euroc_r):
    """One caller alone = a batch of one (same bits as the direct path); contexts with different parameters or image
    sizes get engines of their own; the device-side hooks of a combined call answer GFO_ERR_STATE instead of handing out
    another frame's pyramid; switching combining off restores the direct path on the same context."""
    import gf_orb_slam2_amd as G
    prm = G.StereoParams(480, BF, BF / FX, 0.0)
    a = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=2, combining=True)
    b = G.ORBextractor(700, 1.2, 6, 25, 9, combining=True)          # other parameters: its own engine
    oa, ob = oracle.OracleExtractor(2000, 1.2, 8, 20, 7), oracle.OracleExtractor(700, 1.2, 6, 25, 9)
    ka, da = a(euroc_l)
    assert ka.tobytes() == oa(euroc_l)[0].tobytes() and (da == oa(euroc_l)[1]).all()
    crop = np.ascontiguousarray(euroc_r[30:410, 60:700])
    for im in (euroc_r, crop, euroc_r):                              # b changes image size: engine per size
        kb, db = b(im)
        okb, odb = ob(im)
        assert kb.tobytes() == okb.tobytes() and (db == odb).all()
    with pytest.raises(G.GfoError) as e:
        a.pyramid_level(0)
    assert e.value.code == -5
    r = a.extract_stereo(euroc_l, euroc_r, prm)
    okl, odl = oa(euroc_l)
    okr, odr = oa(euroc_r)
    ref = oracle.stereo_match(okl, odl, okr, odr, oa.scale_factors, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
    assert r[0].tobytes() == okl.tobytes() and r[2].tobytes() == okr.tobytes() and r[4] == ref[0]
    with pytest.raises(G.GfoError):                                  # n_rows beyond what the engine planned
        a.extract_stereo(euroc_l, euroc_r, G.StereoParams(4000, BF, BF / FX, 0.0))
    a.set_combining(False)
    ka2, _ = a(euroc_l)
    assert ka2.tobytes() == ka.tobytes()
    oa(euroc_l)
    np.testing.assert_array_equal(a.pyramid_level(1), oa.level(1))   # direct path: the context owns its pyramid again
    a.close()
    b.close()


def test_context_ids_and_vocabulary_residency(oracle):
    """residency must be a property of the context, not of its address.  A context created right after another
    one was destroyed (very likely at the same address) has a new id and holds no vocabulary."""
    import gf_orb_slam2_amd as G
    e1 = G.ORBextractor(500, 1.2, 8, 20, 7)
    L = e1._L
    assert L.gfo_vocabulary_nodes(e1.handle) == 0
    voc = oracle.make_vocabulary(6, 3, seed=5)
    G.ORBVocabulary(voc, e1)
    assert L.gfo_vocabulary_nodes(e1.handle) == len(voc["first_child"])
    id1, addr1 = e1.ctx_id, e1.handle.value
    e1.close()
    e2 = G.ORBextractor(500, 1.2, 8, 20, 7)
    assert e2.ctx_id != id1 and e2.ctx_id > 0
    assert L.gfo_vocabulary_nodes(e2.handle) == 0, f"fresh context (same address: {e2.handle.value == addr1}) claims a vocabulary"
    e2.close()


def test_frames_from_buffers_the_caller_pinned(oracle, euroc_l, euroc_r):
    """gfo_host_register: images inside a pinned range, tight rows, skip the staging copy (left | right contiguous: one DMA
    copy; apart: one each; a pageable image beside a pinned one: staged) -- same bits on the direct and the combining path."""
    import ctypes as C
    import gf_orb_slam2_amd as G
    L = G.load_library()
    oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    okl, odl = oe(euroc_l)
    okr, odr = oe(euroc_r)
    prm = G.StereoParams(480, BF, BF / FX, 0.0)
    ref = oracle.stereo_match(okl, odl, okr, odr, oe.scale_factors, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
    buf = np.zeros((3, 480, 752), np.uint8)          # left | right contiguous, a third image apart from the first
    buf[0], buf[1], buf[2] = euroc_l, euroc_r, euroc_r
    assert L.gfo_host_register(C.c_void_p(buf.ctypes.data), buf.nbytes) == 0
    try:
        for combining in (False, True):
            e = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=2, combining=combining)
            for l, r in ((buf[0], buf[1]), (buf[0], buf[2]), (buf[0], euroc_r.copy())):    # one copy / two copies / staged
                kl, dl, kr, dr, nm, u, dp, bd, bi = e.extract_stereo(l, r, prm)
                assert kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes() and (dl == odl).all() and (dr == odr).all()
                assert nm == ref[0] and u.tobytes() == ref[1].tobytes() and bi.tobytes() == ref[4].tobytes()
            k1, d1 = e(buf[1])
            assert k1.tobytes() == okr.tobytes() and (d1 == odr).all()
            e.close()
    finally:
        assert L.gfo_host_unregister(C.c_void_p(buf.ctypes.data)) == 0
    assert L.gfo_host_unregister(C.c_void_p(buf.ctypes.data)) != 0      # not registered any more


def test_adapter_pattern_end_to_end_from_many_threads(oracle, euroc_l, euroc_r):
    """The drop-in adapter's whole per-frame pattern from six rigs at once: left and right extractor on two threads
    (gfo_extract, one image each), then Frame::ComputeStereoMatches_Undistorted on the LEFT extractor's context
    (gfo_stereo_match on host arrays) -- which a combining context now also runs as part of a shared device batch (one
    launch of the association kernels for all pairs in flight).  Rigs differ in images, calibration (two parameter sets:
    separate batches) and in whether they pass per-keypoint disparity windows (Frame.cc:1220-1231)."""
    import gf_orb_slam2_amd as G
    RIGS, REPS = 6, 5
    oe = oracle.OracleExtractor(1500, 1.2, 8, 20, 7)
    sf = oe.scale_factors
    pairs = [(euroc_l, euroc_r), (synth_frame(752, 480, 40), synth_frame(752, 480, 41)), (euroc_r, euroc_l)]
    ext_refs = [(oe(l), oe(r)) for l, r in pairs]
    calib = [G.StereoParams(480, BF, BF / FX, 0.0), G.StereoParams(480, 30.0, 0.12, 5.0)]
    rng = np.random.default_rng(3)
    cases = []
    for rig in range(RIGS):
        (kl, dl), (kr, dr) = ext_refs[rig % 3]
        prm = calib[rig % 2]
        win = None
        if rig % 3 != 1:     # two rigs in three track map points: windows around a guessed disparity
            d0 = rng.uniform(0, 60, len(kl)).astype(np.float32)
            win = (np.maximum(d0 - 8, 0).astype(np.float32), np.minimum(d0 + 8, np.float32(prm.mbf / prm.mb)).astype(np.float32))
        ref = oracle.stereo_match(kl, dl, kr, dr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x, *(win or (None, None)))
        cases.append((rig % 3, prm, win, ref))
    exts = [(G.ORBextractor(1500, 1.2, 8, 20, 7, combining=True), G.ORBextractor(1500, 1.2, 8, 20, 7, combining=True)) for _ in range(RIGS)]
    matchers = [G.ORBmatcher(0.8, True, extractor=el) for el, _ in exts]
    bad = []

    def rig_thread(rig):
        pi, prm, win, ref = cases[rig]
        l, r = pairs[pi]
        (okl, odl), (okr, odr) = ext_refs[pi]
        for _ in range(REPS):
            out = {}
            t = threading.Thread(target=lambda: out.__setitem__("r", exts[rig][1](r)))
            t.start()
            kl, dl = exts[rig][0](l)
            t.join()
            kr, dr = out["r"]
            if not (kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes() and (dl == odl).all() and (dr == odr).all()):
                bad.append((rig, "extract"))
                return
            got = matchers[rig].ComputeStereoMatches(kl, dl, kr, dr, sf, prm, *(win or (None, None)))
            if not (got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:]))):
                bad.append((rig, "stereo"))

    _run_threads([lambda rig=rig: rig_thread(rig) for rig in range(RIGS)])
    assert not bad, bad
    b, r = exts[0][0].combiner_stats()
    assert r == RIGS * REPS * 3 and b < r      # three requests per frame: two images and one association; batches were shared
    for el, er in exts:
        el.close()
        er.close()


@pytest.fixture
def patient_rigs():
    """the tests below assert COUNTERS of the paired path (every frame met its partner); Python threads hand the GIL over every 5 ms,
    the product's rig waits 2 ms: for these tests only, the first side waits 200 ms (include/gfo.h gfo_tuning_set; results never
    depend on the wait)"""
    from gf_orb_slam2_amd._lib import load_library
    L = load_library()
    before = L.gfo_tuning_get(b"pair_wait_us")
    assert L.gfo_tuning_set(b"pair_wait_us", 200000) == 0
    yield
    L.gfo_tuning_set(b"pair_wait_us", before)


def _rig_frame(el, er, m, l, r, sf, prm, win=(None, None)):
    """the adapter's per-frame pattern: right image on a thread of its own, left on this one, then the association"""
    out = {}
    t = threading.Thread(target=lambda: out.__setitem__("r", er(r)))
    t.start()
    kl, dl = el(l)
    t.join()
    kr, dr = out["r"]
    return (kl, dl, kr, dr), m.ComputeStereoMatches(kl, dl, kr, dr, sf, prm, *win)


def test_declared_stereo_rig_extracts_as_one_submission_and_answers_the_association(oracle, euroc_l, euroc_r, patient_rigs):
    """gfo_ctx_pair: the adapter's three calls per frame on a declared rig -- the two gfo_extract calls meet
    and run as ONE stereo submission, gfo_stereo_match on the arrays they returned is answered from it -- must return exactly what
    the undeclared pattern returns (= the oracle), frame after frame with CHANGING images; and every way of leaving the fast path
    must fall back to the computed answer: other calibration, disparity windows, modified arrays, a frame of one image only."""
    import gf_orb_slam2_amd as G
    oe = oracle.OracleExtractor(1990, 1.2, 8, 20, 7)
    sf = oe.scale_factors
    frames = [(euroc_l, euroc_r), (synth_frame(752, 480, 40), synth_frame(752, 480, 41)), (euroc_r, euroc_l), (euroc_l, euroc_r)]
    refs = [(oe(l), oe(r)) for l, r in frames]
    prm = G.StereoParams(480, BF, BF / FX, 0.0)
    el, er = G.ORBextractor(1990, 1.2, 8, 20, 7, combining=True), G.ORBextractor(1990, 1.2, 8, 20, 7, combining=True)
    m = G.ORBmatcher(0.8, True, extractor=el)
    el.pair_with(er, prm)

    def check_frame(i, prm_i=prm, win=(None, None)):
        (okl, odl), (okr, odr) = refs[i]
        (kl, dl, kr, dr), got = _rig_frame(el, er, m, *frames[i], sf, prm_i, win)
        assert kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes() and (dl == odl).all() and (dr == odr).all(), f"frame {i}"
        ref = oracle.stereo_match(okl, odl, okr, odr, sf, prm_i.n_rows, prm_i.mbf, prm_i.mb, prm_i.min_x, *win)
        assert got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:])), f"frame {i}"
        return kl, dl, kr, dr

    for rep in range(3):
        for i in range(len(frames)):
            check_frame(i)
    c = el.combiner_counters()
    assert c["rig_frames"] == 12 and c["rig_served"] == 12 and c["rig_alone"] == 0, c
    assert c["requests"] == 12        # ONE engine request per stereo frame: no separate right image, no association request
    # another calibration than the declared one: computed, not served
    other = G.StereoParams(480, 30.0, 0.12, 5.0)
    check_frame(1, other)
    assert el.combiner_counters()["rig_served"] == 12
    # disparity windows: computed
    (okl, _), _ = refs[2]
    d0 = np.random.default_rng(1).uniform(0, 60, len(okl)).astype(np.float32)
    win = (np.maximum(d0 - 8, 0).astype(np.float32), np.minimum(d0 + 8, np.float32(prm.mbf / prm.mb)).astype(np.float32))
    check_frame(2, prm, win)
    assert el.combiner_counters()["rig_served"] == 12
    # arrays that are NOT what the two calls returned (one descriptor bit flipped; keypoints as an undistortion would move them)
    kl, dl, kr, dr = check_frame(0)
    served = el.combiner_counters()["rig_served"]
    assert served == 13
    dl2 = dl.copy(); dl2[7, 3] ^= 1
    got = m.ComputeStereoMatches(kl, dl2, kr, dr, sf, prm)
    ref = oracle.stereo_match(kl, dl2, kr, dr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
    assert got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:]))
    kr2 = kr.copy(); kr2["x"][5] += np.float32(0.25)
    got = m.ComputeStereoMatches(kl, dl, kr2, dr, sf, prm)
    ref = oracle.stereo_match(kl, dl, kr2, dr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
    assert got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:]))
    assert el.combiner_counters()["rig_served"] == served
    # the same untouched arrays again: served again (the stored frame is still the last one)
    got = m.ComputeStereoMatches(kl, dl, kr, dr, sf, prm)
    assert el.combiner_counters()["rig_served"] == served + 1
    # a frame of ONE image (the left extractor used alone): extracted alone after the wait, correct, nothing served from it
    k0, d0_ = el(frames[1][0])
    assert k0.tobytes() == refs[1][0][0].tobytes() and (d0_ == refs[1][0][1]).all()
    assert el.combiner_counters()["rig_alone"] == 1
    got = m.ComputeStereoMatches(kl, dl, kr, dr, sf, prm)      # the stored frame is gone: computed
    assert el.combiner_counters()["rig_served"] == served + 1 and got[0] == oracle.stereo_match(kl, dl, kr, dr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x)[0]
    # the left extractor used on its own for a while: after three frames the rig stops waiting for a partner (no 2-ms stalls), and
    # the next declaration -- the adapter's, at the next stereo frame's association -- wakes it
    import time
    for _ in range(3):          # (the one-image frame above was the first of the streak: two more waits, then the rig sleeps)
        el(frames[1][0])
    assert el.combiner_counters()["rig_alone"] == 3
    t0 = time.perf_counter()
    for _ in range(5):
        el(frames[1][0])
    assert el.combiner_counters()["rig_alone"] == 3 and (time.perf_counter() - t0) / 5 < 0.0015      # asleep: nobody waited
    el.pair_with(er, prm)
    check_frame(2)
    assert el.combiner_counters()["rig_frames"] == 16
    # left and right extracted on ONE thread, the rig declared at every frame (what a single-threaded caller of the adapter would do):
    # the sides never meet; the waits die out -- three per dormancy, and every dormancy takes twice as many declarations to end
    waits0 = el.combiner_counters()["rig_alone"]
    t0 = time.perf_counter()
    for _ in range(40):
        el.pair_with(er, prm)
        kl, dl = el(frames[0][0])
        kr, dr = er(frames[0][1])
        assert kl.tobytes() == refs[0][0][0].tobytes() and kr.tobytes() == refs[0][1][0].tobytes()
    waits = el.combiner_counters()["rig_alone"] - waits0
    assert waits <= 3 * 6, waits              # dormancies end after 1, 2, 4, 8, 16 declarations: at most 6 of them in 40 frames
    el.pair_with(er, prm)
    # dissolved: the plain pattern
    el.pair_with(None, None)
    check_frame(3)
    assert el.combiner_counters()["rig_frames"] == 0      # (the rig's counters went with it)
    el.close()
    er.close()


def test_rigs_from_many_threads_and_a_partner_destroyed_mid_stream(oracle, euroc_l, euroc_r, patient_rigs):
    """six declared rigs at once (their stereo requests share device batches), then a rig whose right extractor is destroyed and
    re-created (Tracking::updateORBExtractor, src/Tracking.cc:298-320): the left side extracts alone until the rig is declared again"""
    import gf_orb_slam2_amd as G
    RIGS, REPS = 6, 6
    oe = oracle.OracleExtractor(1500, 1.2, 8, 20, 7)
    sf = oe.scale_factors
    pairs = [(euroc_l, euroc_r), (synth_frame(752, 480, 40), synth_frame(752, 480, 41)), (euroc_r, euroc_l)]
    refs = [(oe(l), oe(r)) for l, r in pairs]
    calib = [G.StereoParams(480, BF, BF / FX, 0.0), G.StereoParams(480, 30.0, 0.12, 5.0)]
    exts = [(G.ORBextractor(1500, 1.2, 8, 20, 7, combining=True), G.ORBextractor(1500, 1.2, 8, 20, 7, combining=True)) for _ in range(RIGS)]
    ms = [G.ORBmatcher(0.8, True, extractor=el) for el, _ in exts]
    bad = []

    def rig_thread(rig):
        el, er = exts[rig]
        prm = calib[rig % 2]
        for rep in range(REPS):
            pi = (rig + rep) % 3
            (okl, odl), (okr, odr) = refs[pi]
            el.pair_with(er, prm)                  # what the adapter does every frame
            (kl, dl, kr, dr), got = _rig_frame(el, er, ms[rig], *pairs[pi], sf, prm)
            ref = oracle.stereo_match(okl, odl, okr, odr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
            if not (kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes() and (dl == odl).all() and (dr == odr).all()
                    and got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:]))):
                bad.append((rig, rep))

    _run_threads([lambda rig=rig: rig_thread(rig) for rig in range(RIGS)])
    assert not bad, bad
    tot = [e[0].combiner_counters() for e in exts]
    assert sum(t["rig_frames"] for t in tot) >= RIGS * (REPS - 1) and sum(t["rig_served"] for t in tot) >= RIGS * (REPS - 1), tot
    # the right extractor of rig 0 goes away: the left one keeps working, alone
    el, er = exts[0]
    er.close()
    k, d = el(pairs[0][0])
    assert k.tobytes() == refs[0][0][0].tobytes()
    er2 = G.ORBextractor(1500, 1.2, 8, 20, 7, combining=True)
    el.pair_with(er2, calib[0])
    (kl, dl, kr, dr), got = _rig_frame(el, er2, ms[0], *pairs[1], sf, calib[0])
    ref = oracle.stereo_match(*refs[1][0], *refs[1][1], sf, 480, calib[0].mbf, calib[0].mb, calib[0].min_x)
    assert kl.tobytes() == refs[1][0][0].tobytes() and got[0] == ref[0] and got[1].tobytes() == ref[1].tobytes()
    exts[0] = (el, er2)
    for a, b in exts:
        a.close()
        b.close()


def _combine_subprocess(env_extra, body):
    """a fresh process (the combiner's test hooks are read when an engine is created)"""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = "import sys, threading, numpy as np\nsys.path.insert(0, %r)\nimport gf_orb_slam2_amd as G\nfrom oracle import orb_oracle as O\n" % ROOT + body
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env_extra), cwd=ROOT, timeout=170)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


_COMBINE_BODY = """
gold = 'tests/golden/'
l = np.fromfile(gold + 'EuRoC_l_752x480.u8', np.uint8).reshape(480, 752); r = np.fromfile(gold + 'EuRoC_r_752x480.u8', np.uint8).reshape(480, 752)
oe = O.OracleExtractor(1000, 1.2, 8, 20, 7)
refs = [oe(l), oe(r)]
exts = [G.ORBextractor(1000, 1.2, 8, 20, 7, combining=True) for _ in range(6)]
bad = []
def work(i):
    for rep in range(6):
        k, d = exts[i]([l, r][(i + rep) & 1])
        ok, od = refs[(i + rep) & 1]
        if k.tobytes() != ok.tobytes() or not (d == od).all(): bad.append((i, rep))
ts = [threading.Thread(target=work, args=(i,)) for i in range(6)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not bad, bad
print('COUNTERS', exts[0].combiner_counters())
"""


def test_combiner_prepares_slots_as_concurrency_shows_and_survives_allocation_failures():
    """(1) one caller alone prepares two batch slots, not all six; (2) a slot that cannot be prepared later leaves the
    engine with the slots it has; (3) an engine that can prepare none sends every caller down the direct path -- same results;
    (4) a batch that fails as a whole is re-run member by member on the direct path, so nobody inherits another frame's failure."""
    import ast
    import gf_orb_slam2_amd as G
    e = G.ORBextractor(1000, 1.2, 8, 20, 7, combining=True)
    img = synth_frame(752, 480, 3)
    for _ in range(4):
        e(img)
    c = e.combiner_counters()
    assert c["slots_prepared"] == 2 and c["engine_broken"] == 0 and c["batches"] == 4, c
    e.close()

    def counters(out):
        return ast.literal_eval(out.split("COUNTERS", 1)[1].strip())
    c = counters(_combine_subprocess({}, _COMBINE_BODY))
    assert 2 <= c["slots_prepared"] <= 6 and c["requests"] == 36 and c["batches"] < 36 and c["batches_redone"] == 0, c
    c = counters(_combine_subprocess({"GFO_COMBINE_FAIL_PREPARE": "2"}, _COMBINE_BODY))
    assert c["slots_prepared"] == 2 and c["engine_broken"] == 0 and c["requests"] == 36, c
    c = counters(_combine_subprocess({"GFO_COMBINE_FAIL_PREPARE": "0"}, _COMBINE_BODY))
    assert c["slots_prepared"] == 0 and c["engine_broken"] == 1 and c["batches"] == 0, c
    c = counters(_combine_subprocess({"GFO_COMBINE_FAIL_BATCH": "2"}, _COMBINE_BODY))
    assert c["batches_redone"] >= 1 and c["requests"] == 36, c


def test_rig_with_a_late_partner_at_the_product_wait(oracle, euroc_l, euroc_r):
    """the product's rig waits 2 ms (gfo_tuning_get("pair_wait_us")) for the partner's image.  A right camera thread that
    shows up 15 ms late -- a loaded host -- must cost nothing but time: the left side extracts alone after its wait, the late right
    image is extracted alone as well, the association is computed on request, and every array equals the oracle's; the next frame,
    on time, is paired again.  Results are asserted, not counters (which frame met its partner depends on the scheduler)."""
    import time
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd._lib import load_library
    L = load_library()
    assert L.gfo_tuning_get(b"pair_wait_us") == 2000 and L.gfo_tuning_get(b"pair_spin_us") == 400      # nothing left the product values behind
    assert L.gfo_tuning_set(b"no_such_key", 1) == -1 and L.gfo_tuning_get(b"no_such_key") == -1 and L.gfo_tuning_set(b"pair_wait_us", -5) == -1
    oe = oracle.OracleExtractor(1990, 1.2, 8, 20, 7)
    sf = oe.scale_factors
    frames = [(euroc_l, euroc_r), (synth_frame(752, 480, 40), synth_frame(752, 480, 41)), (euroc_r, euroc_l)]
    refs = [(oe(l), oe(r)) for l, r in frames]
    prm = G.StereoParams(480, BF, BF / FX, 0.0)
    el, er = G.ORBextractor(1990, 1.2, 8, 20, 7, combining=True), G.ORBextractor(1990, 1.2, 8, 20, 7, combining=True)
    m = G.ORBmatcher(0.8, True, extractor=el)
    el.pair_with(er, prm)
    alone0 = el.combiner_counters()["rig_alone"]
    for rep in range(9):
        i = rep % 3
        late = rep % 3 == 1                       # every third frame the right camera is late
        out = {}

        def right():
            if late:
                time.sleep(0.015)
            out["r"] = er(frames[i][1])
        t = threading.Thread(target=right)
        t.start()
        kl, dl = el(frames[i][0])
        t.join()
        kr, dr = out["r"]
        (okl, odl), (okr, odr) = refs[i]
        assert kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes() and (dl == odl).all() and (dr == odr).all(), (rep, late)
        got = m.ComputeStereoMatches(kl, dl, kr, dr, sf, prm)
        ref = oracle.stereo_match(okl, odl, okr, odr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
        assert got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:])), (rep, late)
        el.pair_with(er, prm)                     # what the adapter does with every frame's association
    assert el.combiner_counters()["rig_alone"] >= alone0 + 2      # the late frames did take the lone path (15 ms against a 2-ms wait)
    el.close()
    er.close()
