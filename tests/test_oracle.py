"""CPU tests of the oracle: the known answers SURVEY.md 8 fixed for this path, the committed
golden vectors, and cross-checks of each building block against an independent numpy statement."""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, synth_frame

FX, BF = 435.2046959714599, 47.90639384423901


# ---- known answers (level geometry, quotas, umax, pattern extent) -----------------
def test_tables_config_a(oracle):
    e = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    assert e.features_per_level.tolist() == [434, 362, 302, 251, 209, 175, 145, 122]
    assert e.umax.tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    np.testing.assert_allclose(e.scale_factors, [1.0, 1.2, 1.44, 1.728, 2.0736, 2.48832, 2.985985, 3.583182], rtol=1e-6)
    e.compute_pyramid(np.zeros((480, 752), np.uint8))
    assert [e.level_size(l) for l in range(8)] == [(752, 480), (627, 400), (522, 333), (435, 278), (363, 231),
                                                   (302, 193), (252, 161), (210, 134)]
    assert sum(w * h for w, h in (e.level_size(l) for l in range(8))) == 1117367


def test_tables_config_b(oracle):
    e = oracle.OracleExtractor(4000, 1.2, 8, 20, 7)
    assert e.features_per_level.tolist() == [869, 724, 603, 503, 419, 349, 291, 242]
    e.compute_pyramid(np.zeros((1080, 1920), np.uint8))
    assert [e.level_size(l) for l in range(8)] == [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521),
                                                   (772, 434), (643, 362), (536, 301)]


def test_disc_has_749_pixels(oracle):
    um = oracle.OracleExtractor().umax
    assert 31 + 2 * sum(2 * int(um[v]) + 1 for v in range(1, 16)) == 749


def test_pattern_extent():
    txt = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "gfo_pattern.inc")).read()
    import re
    vals = [int(v) for v in re.findall(r"-?\d+", txt.split("*/")[1])]
    assert len(vals) == 1024 and min(vals) == -13 and max(vals) == 12 + 1 - 1 or max(vals) <= 13
    r2 = max(vals[i] ** 2 + vals[i + 1] ** 2 for i in range(0, 1024, 2))
    assert r2 == 338        # radius 18.38 -> 37x37 window


# ---- golden vectors ---------------------------------------------------------------------------
@pytest.mark.parametrize("side", ["l", "r"])
def test_oracle_reproduces_golden_extraction(oracle, side):
    img = np.fromfile(os.path.join(GOLDEN, f"EuRoC_{side}_752x480.u8"), np.uint8).reshape(480, 752)
    kp, desc = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)(img)
    gkp = np.fromfile(os.path.join(GOLDEN, f"EuRoC_{side}_kp.bin"), oracle.KEYPOINT_DTYPE)
    gdesc = np.fromfile(os.path.join(GOLDEN, f"EuRoC_{side}_desc.bin"), np.uint8).reshape(-1, 32)
    assert kp.tobytes() == gkp.tobytes()
    np.testing.assert_array_equal(desc, gdesc)


def test_oracle_reproduces_golden_matches(oracle):
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)
    kr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_kp.bin"), kd)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    dr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_desc.bin"), np.uint8).reshape(-1, 32)
    sf = oracle.OracleExtractor().scale_factors
    g = np.load(os.path.join(GOLDEN, "EuRoC_stereo.npz"))
    nm, u, dp, bd, bi = oracle.stereo_match(kl, dl, kr, dr, sf, 480, BF, BF / FX, 0.0)
    assert nm == int(g["nmatched"])
    for a, b in ((u, g["u_right"]), (dp, g["depth"]), (bd, g["best_dist"]), (bi, g["best_idx"])):
        assert a.tobytes() == b.tobytes()
    p = np.load(os.path.join(GOLDEN, "EuRoC_projection.npz"))
    nmm, out_mp, out_sc = oracle.search_by_projection(kl, dl, u, sf, (0.0, 0.0, 752.0, 480.0), p["mps"], p["mp_desc"], 3.0, 0.8, p["taken"])
    assert nmm == int(p["nmatches"])
    np.testing.assert_array_equal(out_mp, p["out_mp"])
    np.testing.assert_array_equal(out_sc, p["out_score"])


# ---- building blocks vs independent numpy statements ------------------------------------------
RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1),
        (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_numpy(img, t):
    """FAST-9/16 + NMS straight from the definition (score = largest threshold still a corner)."""
    h, w = img.shape
    im = img.astype(np.int32)
    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = [im[y, x] - im[y + dy, x + dx] for dx, dy in RING]
            best = -1000
            for s in range(16):
                arc = [d[(s + k) % 16] for k in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > t:
                score[y, x] = best - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s > 0 or (img[y, x] is not None and score[y, x] == 0 and False):
                nb = score[y - 1:y + 2, x - 1:x + 2].copy()
                nb[1, 1] = -1
                if (s > nb).all():
                    out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed,t", [(0, 20), (1, 7), (2, 12)])
def test_fast_matches_definition(oracle, seed, t):
    rng = np.random.default_rng(seed)
    img = synth_frame(96, 64, 40 + seed)[:40, :44].copy()
    img[rng.integers(0, 40, 30), rng.integers(0, 44, 30)] = rng.integers(0, 256, 30)
    got = [tuple(r) for r in oracle.fast9_nms(img, t).tolist()]
    assert got == fast_numpy(img, t)


def test_resize_matches_float_bilinear_within_one(oracle):
    img = synth_frame(200, 120, 3)
    dw, dh = 167, 100
    got = oracle.resize_linear(img, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (200 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (120 / dh) - 0.5
    x0 = np.clip(np.floor(sx).astype(int), 0, 198); fx = np.clip(sx - x0, 0, 1)
    y0 = np.clip(np.floor(sy).astype(int), 0, 118); fy = np.clip(sy - y0, 0, 1)
    f = img.astype(np.float64)
    ref = ((f[y0][:, x0] * (1 - fx) + f[y0][:, x0 + 1] * fx) * (1 - fy)[:, None] +
           (f[y0 + 1][:, x0] * (1 - fx) + f[y0 + 1][:, x0 + 1] * fx) * fy[:, None])
    assert np.abs(got - ref).max() <= 1.0


def test_blur_is_exact_integer_convolution(oracle):
    img = synth_frame(90, 70, 9)
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    pad = np.pad(img.astype(np.int64), 3, mode="reflect")          # numpy 'reflect' == BORDER_REFLECT_101
    hp = sum(k[i] * pad[:, i:i + 90] for i in range(7))
    vp = sum(k[i] * hp[i:i + 70, :] for i in range(7))
    ref = np.minimum(255, (vp + 32768) >> 16).astype(np.uint8)
    np.testing.assert_array_equal(oracle.gaussian_blur7(img), ref)


def test_fast_atan2_close_to_libm(oracle):
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-2 ** 22, 2 ** 22, 2)
        if x == 0 and y == 0:
            continue
        a = oracle.fast_atan2(float(y), float(x))
        ref = math.degrees(math.atan2(y, x)) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.02
    assert oracle.fast_atan2(0.0, 0.0) == 0.0


def _product_trig():
    """The PRODUCT's trig header (include/gfo_sincos.h, what the HIP kernels compile) built as a host library of
    array functions -- so the oracle's independent restatements can be compared with it without a GPU."""
    import ctypes
    import subprocess
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    src, so = os.path.join(bdir, "product_trig.c"), os.path.join(bdir, "libproduct_trig.so")
    code = ('#include "../../include/gfo_sincos.h"\n'
            'void pt_atan2_n(const float* y, const float* x, int n, float* o) { for (int i = 0; i < n; i++) o[i] = gfo_fast_atan2f(y[i], x[i]); }\n'
            'void pt_sincos_n(const float* t, int n, float* s, float* c) { for (int i = 0; i < n; i++) gfo_sincosf(t[i], &s[i], &c[i]); }\n')
    if not os.path.exists(src) or open(src).read() != code:
        open(src, "w").write(code)
    hdr = os.path.join(ROOT, "include", "gfo_sincos.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["gcc", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src, "-lm"])
    L = ctypes.CDLL(so)
    vp = ctypes.c_void_p
    L.pt_atan2_n.argtypes = [vp, vp, ctypes.c_int, vp]
    L.pt_sincos_n.argtypes = [vp, ctypes.c_int, vp, vp]
    return L


def test_oracle_trig_is_independent_of_the_product_header_and_agrees_with_it(oracle):
    """oracle/orb_oracle.c restates fastAtan2 and sin/cos on its own (separately typed constants; long double sin/cos
    rounded once) -- it must not include the product's include/gfo_sincos.h -- and the two must agree bit for bit on
    every angle the fixtures produce and on 1M-sample sweeps: a mistyped polynomial constant on EITHER side fails here
    (CPU) and in every GPU parity test."""
    src = open(os.path.join(ROOT, "oracle", "orb_oracle.c")).read()
    assert '#include "../include/gfo_sincos.h"' not in src
    P = _product_trig()
    rng = np.random.default_rng(11)
    # fastAtan2 on intensity-centroid moments (|m| < 2^23, ORBextractor.cc:76-103) and on arbitrary floats
    n = 1_000_000
    y = rng.integers(-2 ** 23, 2 ** 23, n).astype(np.float32)
    x = rng.integers(-2 ** 23, 2 ** 23, n).astype(np.float32)
    y[:1000] = 0; x[1000:2000] = 0; y[2000:3000] = x[2000:3000]; y[3000:4000] = -x[3000:4000]
    got = np.zeros(n, np.float32)
    P.pt_atan2_n(y.ctypes.data, x.ctypes.data, n, got.ctypes.data)
    np.testing.assert_array_equal(got.view(np.uint32), oracle.fast_atan2_n(y, x).view(np.uint32))
    yf = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    xf = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 6, n)).astype(np.float32)
    P.pt_atan2_n(yf.ctypes.data, xf.ctypes.data, n, got.ctypes.data)
    np.testing.assert_array_equal(got.view(np.uint32), oracle.fast_atan2_n(yf, xf).view(np.uint32))
    # sin/cos on the angles of the golden keypoints and on a sweep of [0, 360) degrees, radians formed as the
    # reference does (float angle * (float)(CV_PI / 180.f), ORBextractor.cc:75,110)
    factor = np.float32(np.pi / np.float32(180.0))
    degs = [np.fromfile(os.path.join(GOLDEN, f"EuRoC_{s_}_kp.bin"), oracle.KEYPOINT_DTYPE)["angle"] for s_ in "lr"]
    degs.append(rng.uniform(0, 360, n).astype(np.float32))
    degs.append(np.nextafter(np.float32(360.0), np.float32(0)) * rng.random(n).astype(np.float32) ** 4)   # dense near 0
    for deg in degs:
        rad = (deg.astype(np.float32) * factor).astype(np.float32)
        s1 = np.zeros(len(rad), np.float32); c1 = np.zeros(len(rad), np.float32)
        P.pt_sincos_n(rad.ctypes.data, len(rad), s1.ctypes.data, c1.ctypes.data)
        s2, c2 = oracle.sincos_n(rad)
        np.testing.assert_array_equal(s1.view(np.uint32), s2.view(np.uint32))
        np.testing.assert_array_equal(c1.view(np.uint32), c2.view(np.uint32))


def test_shared_sincos_is_correctly_rounded_almost_everywhere(oracle):
    """the oracle's sin/cos vs float64 libm rounded to float, over the angles the descriptor uses."""
    rng = np.random.default_rng(1)
    deg = rng.uniform(0, 360, 20000).astype(np.float32)
    rad = (deg * np.float32(np.pi / 180.0)).astype(np.float32)
    bad = 0
    for t in rad:
        s, c = oracle.sincos(float(t))
        bad += (np.float32(s) != np.float32(math.sin(float(t)))) + (np.float32(c) != np.float32(math.cos(float(t))))
    assert bad == 0


def test_trig_and_fma_variants_change_no_descriptor_on_fixtures(oracle):
    """How far the deterministic arithmetic is from what the reference literally calls (libm cosf/sinf,
    possibly FMA-contracted rotation): counted, not hidden."""
    img = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_752x480.u8"), np.uint8).reshape(480, 752)
    base = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    _, d0 = base(img)
    alt = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    alt.set_variant(oracle.TRIG_LIBM, oracle.ROT_UNFUSED)
    _, d1 = alt(img)
    assert int((d0 != d1).any(axis=1).sum()) == 0
    alt.set_variant(oracle.TRIG_LIBM, oracle.ROT_FMA)
    _, d2 = alt(img)
    nbits = int(np.unpackbits(d0 ^ d2).sum())
    assert nbits <= 8, f"{nbits} descriptor bits depend on FMA contraction"


def test_hamming_is_popcount_of_xor(oracle):
    rng = np.random.default_rng(2)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert oracle.hamming256(a, b) == int(np.unpackbits(a ^ b).sum())
    z = np.zeros(32, np.uint8)
    assert oracle.hamming256(z, z) == 0 and oracle.hamming256(z, ~z) == 256


def test_extraction_invariants(oracle):
    e = oracle.OracleExtractor(1000, 1.2, 8, 20, 7)
    img = synth_frame(640, 480, 21)
    kp, desc = e(img)
    q = e.features_per_level
    assert len(kp) >= 900
    for l in range(8):
        n = e.level_keypoint_count(l)
        assert n <= max(q[l], 8) + 3                       # quadtree overshoot is at most 3 nodes
        w, h = e.level_size(l)
        sel = kp[kp["octave"] == l]
        s = e.scale_factors[l]
        xs = sel["x"] / s; ys = sel["y"] / s
        assert (xs > 18.9).all() and (xs < w - 18.9).all() and (ys > 18.9).all() and (ys < h - 18.9).all()
    assert (np.diff(kp["octave"]) >= 0).all()              # rows laid out level by level
    assert ((kp["angle"] >= 0) & (kp["angle"] < 360)).all()
    assert (kp["class_id"] == -1).all()
    # empty / degenerate inputs
    assert len(e(np.zeros((480, 640), np.uint8))[0]) == 0
    assert len(e(np.zeros((40, 40), np.uint8))[0]) == 0
    kp2, _ = e(img)
    assert kp2.tobytes() == kp.tobytes()                   # idempotent


def stereo_numpy(kl, dl, kr, dr, sf, n_rows, mbf, mb, min_x):
    """Predicate form of the stereo association (what the HIP kernel evaluates)."""
    nl = len(kl)
    u = np.full(nl, -1, np.float32); dp = np.full(nl, -1, np.float32)
    bd = np.full(nl, -1, np.int32); bi = np.full(nl, -1, np.int32)
    nm = 0
    r = (np.float32(2.0) * sf[kr["octave"]]).astype(np.float32)
    maxr = np.minimum(np.float32(n_rows - 1), np.ceil(kr["y"] + r)).astype(np.int32)
    minr = np.maximum(np.float32(0), np.floor(kr["y"] - r)).astype(np.int32)
    maxD = np.float32(mbf) / np.float32(mb)
    dist_all = np.unpackbits(dl[:, None, :] ^ dr[None, :, :], axis=2).sum(axis=2)
    for i in range(nl):
        vL, uL = kl["y"][i], kl["x"][i]
        if vL < 0 or vL > n_rows - 1:
            continue
        row = int(vL)
        band = (minr <= row) & (row <= maxr)
        if not band.any() or (uL - np.float32(0)) < min_x:
            continue
        nm += 1
        ok = band & (np.abs(kr["octave"] - kl["octave"][i]) <= 1) & (kr["x"] >= uL - maxD) & (kr["x"] <= uL)
        idx = np.nonzero(ok)[0]
        if len(idx) == 0:
            continue
        d = dist_all[i, idx]
        j = idx[np.argmin(d)]          # first minimum = lowest index
        if d.min() < 75 and d.min() < 100:
            disp = np.float32(uL - kr["x"][j])
            if disp >= 0 and disp < maxD:
                bu = kr["x"][j]
                if disp <= 0:
                    disp = np.float32(0.01); bu = np.float32(uL - np.float32(0.01))
                dp[i] = np.float32(mbf) / disp; u[i] = bu; bd[i] = d.min(); bi[i] = j
    acc = np.sort(bd[bd >= 0])
    if len(acc):
        th = np.float32(1.5) * np.float32(1.4) * np.float32(acc[len(acc) // 2])
        drop = (bd >= 0) & ~(bd.astype(np.float32) < th)
        u[drop] = -1; dp[drop] = -1; nm -= int(drop.sum())
    return nm, u, dp, bd, bi


def test_stereo_row_table_equals_predicate_form(oracle):
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)[::4]
    kr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_kp.bin"), kd)[::4]
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)[::4]
    dr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_desc.bin"), np.uint8).reshape(-1, 32)[::4]
    sf = oracle.OracleExtractor().scale_factors
    ref = oracle.stereo_match(kl, dl, kr, dr, sf, 480, BF, BF / FX, 0.0)
    got = stereo_numpy(kl, dl, kr, dr, sf, 480, BF, BF / FX, 0.0)
    assert got[0] == ref[0]
    for a, b in zip(got[1:], ref[1:]):
        np.testing.assert_array_equal(a, b)


def _stereo_fixture(oracle, step=1):
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)[::step]
    kr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_kp.bin"), kd)[::step]
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)[::step]
    dr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_desc.bin"), np.uint8).reshape(-1, 32)[::step]
    return kl, dl, kr, dr, oracle.OracleExtractor().scale_factors


def stereo_second_call_from_fresh_answers(oracle, kl, dl, kr, dr, sf, state, min_d, max_d, mbf, mb, online=False):
    """What the adapter (adapter/matchers_gfo.cc) derives for a call on a frame that already holds an association, from the
    answer a FRESH call with the same windows gives (gfo_stereo_match: arrays after its own cut, best_dist / best_idx before it):
    state = (mvuRight, mvDepth, mvDistIdx) is kept, accepted matches overwrite and are appended, the cut runs over everything."""
    ur, dp, di = state[0].copy(), state[1].copy(), [tuple(x) for x in state[2].tolist()]
    nm, ur2, dp2, bd, bi = oracle.stereo_match(kl, dl, kr, dr, sf, 480, mbf, mb, 0.0, min_d, max_d)
    acc = np.nonzero(bd >= 0)[0]
    visited = nm + int(((bd >= 0) & (dp2 < 0)).sum())              # the fresh call's count + what its own cut took away
    for i in acc:
        uL = kl["x"][i]
        bu = kr["x"][bi[i]]
        disp = np.float32(uL - bu)
        if disp <= 0:
            disp = np.float32(0.01)
            bu = np.float32(np.float64(uL) - 0.01)
        dp[i] = np.float32(mbf) / disp
        ur[i] = bu
        di.append((int(bd[i]), int(i)))
    nmatched = visited
    if not online and di:
        di.sort()
        th = np.float32(1.5) * np.float32(1.4) * np.float32(di[len(di) // 2][0])
        for d, i in reversed(di):
            if np.float32(d) < th:
                break
            ur[i] = -1
            dp[i] = -1
            nmatched -= 1
    return nmatched, ur, dp, np.array(di, np.int32).reshape(-1, 2)


def test_stereo_frame_state_first_call_is_the_fresh_call(oracle):
    """orc_stereo_frame (the Frame's stereo members across calls): its first call equals orc_stereo_match, offline and online"""
    kl, dl, kr, dr, sf = _stereo_fixture(oracle)
    for online in (False, True):
        ref = oracle.stereo_match(kl, dl, kr, dr, sf, 480, BF, BF / FX, 0.0, online=online)
        nm, ur, dp, di = oracle.StereoFrame(kl, dl, kr, dr, sf, 480, BF, BF / FX).match(online=online)
        assert nm == ref[0]
        np.testing.assert_array_equal(ur, ref[1])
        np.testing.assert_array_equal(dp, ref[2])
        want = [(int(ref[3][i]), i) for i in range(len(kl)) if ref[3][i] >= 0]
        assert sorted(map(tuple, di.tolist())) == sorted(want)
        if not online:
            assert [tuple(x) for x in di.tolist()] == sorted(want)         # an offline call leaves mvDistIdx sorted (:1296)


def test_stereo_frame_second_call_keeps_state(oracle):
    """Frame.cc:1173-1176 + Tracking.cc:941-954: a second call on the same frame does not reset; the literal form against the
    derivation the adapter makes from a fresh call's answer"""
    kl, dl, kr, dr, sf = _stereo_fixture(oracle)
    n = len(kl)
    rng = np.random.default_rng(5)
    mb = np.float32(BF / FX)
    F = oracle.StereoFrame(kl, dl, kr, dr, sf, 480, BF, mb)
    first = F.match()
    # windows around the first call's disparity for most matched keypoints, a wrong window for some (their match is rejected now
    # and the first call's value must survive), none for the rest
    disp = np.where(first[1] >= 0, kl["x"] - first[1], rng.uniform(0, 40, n)).astype(np.float32)
    has = rng.random(n) < 0.6
    wrong = has & (rng.random(n) < 0.3)
    centre = np.where(wrong, disp + 120, disp + rng.normal(0, 20, n)).astype(np.float32)
    full = np.float32(np.float32(BF) / mb)
    min_d = np.where(has, np.maximum(centre - 50, 0), 0).astype(np.float32)
    max_d = np.where(has, np.minimum(centre + 50, full), full).astype(np.float32)
    state1 = F.state()
    want = stereo_second_call_from_fresh_answers(oracle, kl, dl, kr, dr, sf, state1, min_d, max_d, BF, mb)
    got = F.match(min_d, max_d, has.astype(np.uint8))
    assert got[0] == want[0]
    np.testing.assert_array_equal(got[1], want[1])
    np.testing.assert_array_equal(got[2], want[2])
    np.testing.assert_array_equal(got[3], want[3])
    # the point of it: this is NOT what a fresh frame would answer with the same windows
    fresh = oracle.stereo_match(kl, dl, kr, dr, sf, 480, BF, mb, 0.0, min_d, max_d)
    kept = (fresh[1] < 0) & (got[1] >= 0)
    assert kept.sum() > 20, kept.sum()
    assert len(got[3]) > len(state1[2])                                    # mvDistIdx accumulated (:1282)
    # a third, online call on the same state: no cut, nothing reset
    state2 = F.state()
    want3 = stereo_second_call_from_fresh_answers(oracle, kl, dl, kr, dr, sf, state2, min_d, max_d, BF, mb, online=True)
    got3 = F.match(min_d, max_d, has.astype(np.uint8), online=True)
    assert got3[0] == want3[0]
    for a, b in zip(got3[1:], want3[1:]):
        np.testing.assert_array_equal(a, b)
    # PrepareStereoCandidates called by the caller (Tracking.cc:613) resets everything: the next call is a first call again
    F.prepare()
    again = F.match()
    assert again[0] == first[0]
    for a, b in zip(again[1:], first[1:]):
        np.testing.assert_array_equal(a, b)


def bow_python(kd_, ka, valid, kfv, fd, fa, ffv, ratio, ori, budget=0):
    """SearchByBoW written straight from ORBmatcher.cc:270-404 with Python containers."""
    kmap = {int(n): list(kfv[2][kfv[1][i]:kfv[1][i + 1]]) for i, n in enumerate(kfv[0])}
    fmap = {int(n): list(ffv[2][ffv[1][i]:ffv[1][i + 1]]) for i, n in enumerate(ffv[0])}
    out = [-1] * len(fd)
    hist = [[] for _ in range(30)]
    nm = 0
    for node in sorted(set(kmap) & set(fmap)):
        for rk in kmap[node]:
            if not valid[rk]:
                continue
            b1, bi, b2 = 256, -1, 256
            for rf in fmap[node]:
                if out[rf] >= 0:
                    continue
                d = int(np.unpackbits(kd_[rk] ^ fd[rf]).sum())
                if d < b1:
                    b2, b1, bi = b1, d, rf
                elif d < b2:
                    b2 = d
            if b1 <= 50 and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                out[bi] = int(rk)
                if ori:
                    rot = np.float32(ka[rk]) - np.float32(fa[bi])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360.0))
                    v = float(np.float32(rot * np.float32(1.0 / 30)))
                    b = int(np.floor(v + 0.5))
                    hist[0 if b == 30 else b].append(bi)
                nm += 1
                if budget and nm >= budget:          # BUDGETING_FEATURE_MATCHING, ORBmatcher.cc:360-365: out of THIS node's loop
                    break
    if ori:
        sizes = [len(h) for h in hist]
        order = sorted(range(30), key=lambda i: (-sizes[i], i))
        m1, m2, m3 = sizes[order[0]], sizes[order[1]], sizes[order[2]]
        keep = {order[0]} if m1 > 0 else set()
        if m1 > 0 and not (m2 < 0.1 * m1) and m2 > 0:
            keep.add(order[1])
            if not (m3 < 0.1 * m1) and m3 > 0:
                keep.add(order[2])
        for i in range(30):
            if i not in keep:
                for j in hist[i]:
                    out[j] = -1
                    nm -= 1
    return nm, np.array(out, np.int32)


def test_bow_oracle_against_python_statement(oracle):
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)[:600]
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)[:600]
    rng = np.random.default_rng(4)
    fd = dl.copy()
    for _ in range(8):
        sel = rng.random(len(fd)) < 0.5
        bits = rng.integers(0, 256, len(fd))
        fd[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    perm = rng.permutation(len(fd))
    fd = fd[perm]
    fa = ((kl["angle"][perm] + rng.normal(0, 20, len(fd))) % 360).astype(np.float32)
    valid = (rng.random(len(dl)) > 0.1).astype(np.uint8)
    kfv = oracle.make_feature_vector(dl[:, 0] >> 3)
    ffv = oracle.make_feature_vector(fd[:, 0] >> 3)
    for ratio, ori in ((0.7, True), (0.9, False)):
        ref = bow_python(dl, kl["angle"], valid, kfv, fd, fa, ffv, ratio, ori)
        got = oracle.search_by_bow(dl, kl["angle"], valid, kfv, fd, fa, ffv, ratio, ori)
        assert got[0] == ref[0]
        np.testing.assert_array_equal(got[1], ref[1])
    assert oracle.three_maxima([5, 1, 9, 0, 9, 3]) == (2, 4, 0)
    assert oracle.three_maxima([100, 5, 3] + [0] * 27) == (0, -1, -1)


def test_feature_budget_variant_of_the_oracle(oracle):
    """BUDGETING_FEATURE_MATCHING (ORBmatcher.h:36-37, ORBmatcher.cc:360-365, 1547-1552): SearchByBoW against the Python statement with
    the same break; SearchByProjection(Cur, Last) against its defining property -- without the rotation check a budgeted run IS the
    unbudgeted run over the prefix of queries that ends with the K-th accepted one"""
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    rng = np.random.default_rng(4)
    fd = dl.copy()
    bits = rng.integers(0, 256, (len(fd), 4))
    for j in range(4):
        fd[np.arange(len(fd)), bits[:, j] >> 3] ^= (1 << (bits[:, j] & 7)).astype(np.uint8)
    node_k = (dl[:, 0] >> 3).astype(np.int64)
    node_f = (fd[:, 0] >> 3).astype(np.int64)
    kfv, ffv = oracle.make_feature_vector(node_k), oracle.make_feature_vector(node_f)
    valid = np.ones(len(dl), np.uint8)
    full = oracle.search_by_bow(dl, kl["angle"], valid, kfv, fd, kl["angle"], ffv, 0.75, True)
    assert full[0] > 600
    for K in (1, 37, 150, 400, 100000):
        with oracle.feature_budget(K):
            got = oracle.search_by_bow(dl, kl["angle"], valid, kfv, fd, kl["angle"], ffv, 0.75, True)
        want = bow_python(dl, kl["angle"], valid, kfv, fd, kl["angle"], ffv, 0.75, True, budget=K)
        assert got[0] == want[0], K
        np.testing.assert_array_equal(got[1], want[1])
        if K == 100000:
            assert got[0] == full[0] and (got[1] == full[1]).all()
        elif K == 150:
            assert got[0] < full[0]                      # the budget bites; every node after the crossing still adds its first match
    assert oracle.search_by_bow(dl, kl["angle"], valid, kfv, fd, kl["angle"], ffv, 0.75, True)[0] == full[0]      # the switch is off again
    # projection queries (the (Cur, Last) form): m queries around the keypoints
    n, m = len(kl), 1200
    src = rng.integers(0, n, m)
    q = np.zeros(m, oracle.PROJ_QUERY_DTYPE)
    q["u"] = kl["x"][src] + rng.normal(0, 1.5, m); q["v"] = kl["y"][src] + rng.normal(0, 1.5, m); q["ur"] = q["u"] - 4
    q["radius"] = (np.float32(7.0) * oracle.OracleExtractor().scale_factors[kl["octave"][src]]).astype(np.float32)
    q["min_level"] = kl["octave"][src] - 1; q["max_level"] = kl["octave"][src] + 1
    q["angle"] = kl["angle"][src]; q["flags"] = np.where(rng.random(m) < 0.8, 1 | 4, 1)
    qd = dl[src].copy()
    bounds = (0.0, 0.0, 752.0, 480.0)
    plain = oracle.search_by_projection_queries(kl, dl, None, kl["angle"], bounds, q, qd, False, 0.9, 100, False)
    assert plain[0] > 800
    for K in (1, 150, 500):
        with oracle.feature_budget(K):
            got = oracle.search_by_projection_queries(kl, dl, None, kl["angle"], bounds, q, qd, False, 0.9, 100, False)
        assert got[0] == K
        # the prefix that ends with the K-th accepted query: the shortest prefix whose unbudgeted run accepts K
        lo, hi = 1, m
        while lo < hi:
            mid = (lo + hi) // 2
            if oracle.search_by_projection_queries(kl, dl, None, kl["angle"], bounds, q[:mid], qd[:mid], False, 0.9, 100, False)[0] >= K:
                hi = mid
            else:
                lo = mid + 1
        want = oracle.search_by_projection_queries(kl, dl, None, kl["angle"], bounds, q[:lo], qd[:lo], False, 0.9, 100, False)
        assert want[0] == K
        np.testing.assert_array_equal(got[1], want[1])
        np.testing.assert_array_equal(got[2], want[2])
    with oracle.feature_budget(150):                        # with the rotation check: the 150th match is not in the histogram, hence never cleared
        got = oracle.search_by_projection_queries(kl, dl, None, kl["angle"], bounds, q, qd, False, 0.9, 100, True)
    assert 0 < got[0] <= 150


def test_ocv_variant_switches_are_single_and_restorable(oracle):
    """oracle/ocv_variants.json is what the binding applies at load; every [OCV] switch changes its block and nothing
    else, and restoring the committed table restores the bytes (the golden vectors are made under it)."""
    committed = oracle.load_ocv_variants()
    assert oracle.get_ocv_variants() == committed == {"resize": 0, "atan_fma": 0, "blur_round": 0, "gauss_taps": [18, 34, 49, 55, 49, 34, 18]}
    img = synth_frame(320, 240, 2)
    base = (oracle.resize_linear(img, 267, 200), oracle.gaussian_blur7(img))
    y = np.arange(-500, 500, dtype=np.float32) * 37.0
    x = np.arange(1000, dtype=np.float32)[::-1] * 11.0 - 3000.0
    a0 = oracle.fast_atan2_n(y, x)
    try:
        oracle.set_ocv_variants(resize=1)
        r1 = oracle.resize_linear(img, 267, 200)
        assert (r1 != base[0]).any() and np.abs(r1.astype(int) - base[0]).max() == 1      # float vs 11-bit fixed point: one grey level
        assert (oracle.gaussian_blur7(img) == base[1]).all()                                # ... and nothing else moved
        oracle.set_ocv_variants(resize=0, blur_round=1)
        assert (oracle.gaussian_blur7(img) != base[1]).any() and (oracle.resize_linear(img, 267, 200) == base[0]).all()
        oracle.set_ocv_variants(blur_round=0, gauss_taps=[18, 34, 49, 54, 49, 34, 18])     # a kernel renormalised to sum 256
        assert (oracle.gaussian_blur7(img) != base[1]).mean() > 0.3
        oracle.set_ocv_variants(gauss_taps=committed["gauss_taps"], atan_fma=1)
        a1 = oracle.fast_atan2_n(y, x)
        assert (a1.view(np.uint32) != a0.view(np.uint32)).any() and np.abs(a1 - a0).max() < 1e-4
    finally:
        oracle.set_ocv_variants(**committed)
    assert (oracle.resize_linear(img, 267, 200) == base[0]).all() and (oracle.gaussian_blur7(img) == base[1]).all()
    assert (oracle.fast_atan2_n(y, x).view(np.uint32) == a0.view(np.uint32)).all()


def _run_cv2_check(extra_path=None):
    import subprocess
    import sys
    env = dict(os.environ)
    if extra_path:
        env["PYTHONPATH"] = extra_path + os.pathsep + env.get("PYTHONPATH", "")
    return subprocess.run([sys.executable, os.path.join(GOLDEN, "check_against_cv2.py")], capture_output=True, text=True, env=env, timeout=600)


def test_cv2_pin_script_skips_cleanly_without_opencv():
    """tests/golden/check_against_cv2.py is the one-command pin for whoever has cv2 3.4.x.  Here
    OpenCV is absent: it must say so and exit 0, comparing nothing."""
    try:
        import cv2  # noqa: F401
        pytest.skip("cv2 is importable here: run tests/golden/check_against_cv2.py itself, it is the pin")
    except ImportError:
        pass
    r = _run_cv2_check()
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cv2 absent" in r.stdout and "parity unpinned" in r.stdout


def test_cv2_pin_script_control_flow_against_a_mock(oracle, tmp_path):
    """The script cannot meet a real cv2 in this image, so its CONTROL FLOW is exercised with a mock module named cv2 that
    answers every call from the oracle itself -- once under the committed variant table (every stage must come out equal),
    once under a deliberately different one (Gaussian centre tap 54, fused atan2 Horner steps: the stages must come out
    DIFFERS, name the matching candidate and exit 1).  The mock pins NOTHING (its version string makes the script say so);
    it only guarantees that the day a real cv2 is present the script runs instead of dying on a typo."""
    mock = tmp_path / "cv2.py"
    mock.write_text('''
import os, sys, numpy as np
sys.path.insert(0, %r)
from oracle import orb_oracle as O
__version__ = "0.0-mock"
INTER_LINEAR = 1; BORDER_REFLECT_101 = 4; FAST_FEATURE_DETECTOR_TYPE_9_16 = 2
_ALT = os.environ.get("MOCK_CV2_ALT") == "1"
def _with(fn, **kw):
    cur = O.get_ocv_variants()
    O.set_ocv_variants(**kw)
    try: return fn()
    finally: O.set_ocv_variants(**cur)
def resize(src, size, interpolation=None): return _with(lambda: O.resize_linear(src, size[0], size[1]), resize=0)
def copyMakeBorder(a, t, b, l, r, kind): return np.pad(a, ((t, b), (l, r)), mode="reflect")
def getGaussianKernel(n, s):
    x = np.arange(n) - (n - 1) / 2.0; k = np.exp(-x * x / (2 * s * s)); return (k / k.sum()).reshape(-1, 1)
def GaussianBlur(a, ks, sx, sy, borderType=None):
    return _with(lambda: O.gaussian_blur7(a), blur_round=0, gauss_taps=[18, 34, 49, 54 if _ALT else 55, 49, 34, 18])
class _KP:
    def __init__(s, x, y, r): s.pt = (float(x), float(y)); s.response = float(r)
class _Det:
    def __init__(s, t): s.t = t
    def detect(s, img, mask): return [_KP(*r) for r in O.fast9_nms(img, s.t).tolist()]
def FastFeatureDetector_create(threshold=10, nonmaxSuppression=True, type=2): return _Det(threshold)
def fastAtan2(y, x): return float(_with(lambda: O.fast_atan2_n(np.float32([y]), np.float32([x]))[0], atan_fma=1 if _ALT else 0))
def phase(x, y, angleInDegrees=False): return _with(lambda: O.fast_atan2_n(y, x), atan_fma=1 if _ALT else 0)
''' % ROOT)
    r = _run_cv2_check(str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "EQUAL (but cv2 is not 3.4.x)" in r.stdout and "DIFFERS" not in r.stdout
    os.environ["MOCK_CV2_ALT"] = "1"
    try:
        r = _run_cv2_check(str(tmp_path))
    finally:
        del os.environ["MOCK_CV2_ALT"]
    assert r.returncode == 1, r.stdout[-3000:] + r.stderr[-3000:]
    assert "[DIFFERS] blur/EuRoC_l" in r.stdout and "round=0 centre=54" in r.stdout
    assert "[DIFFERS] fastAtan2/scalar" in r.stdout and '"atan_fma": 1' in r.stdout
    assert "[ok] resize/EuRoC_l" in r.stdout and "NOT PINNED" in r.stdout
    rep = os.path.join(GOLDEN, "cv2_pin_report.json")
    if os.path.exists(rep):
        os.remove(rep)       # a mock's report must not be mistaken for a pin


# ---- the one reference-compiled pin of the path: DBoW2's BowVector / FeatureVector (oracle/_ref) ----
BOW_W = {"TF_IDF": 0, "TF": 1, "IDF": 2, "BINARY": 3}
BOW_N = {"none": 0, "L1": 1, "L2": 2}


def _bow_fixture():
    return np.load(os.path.join(GOLDEN, "bow_fold.npz"))


def test_bow_fold_equals_the_reference_compiled_fixtures(oracle):
    """tests/golden/bow_fold.npz holds outputs of the REFERENCE's own BowVector.cpp / FeatureVector.cpp (compiled unmodified into
    oracle/_ref/libdbow2_fold.so by tests/golden/make_bow_fold_golden.py): the oracle's fold must give the same maps -- word ids,
    node ids, index lists, and every WordValue bit for bit -- for the 4 weightings x 3 norms on every stream, including empty and
    one-feature streams, stop words, negative / NaN / denormal weights."""
    fx = _bow_fixture()
    streams = sorted({k.split(".")[0] for k in fx.files})
    assert len(streams) == 10
    cases = 0
    for name in streams:
        word, weight, node = fx[f"{name}.word"], fx[f"{name}.weight"], fx[f"{name}.node"]
        for wn, w in BOW_W.items():
            for nn, nm in BOW_N.items():
                bw, bv, fn, fs, fi = oracle.bow_fold(word, weight, node, w, nm)
                pre = f"{name}.{wn}.{nn}."
                np.testing.assert_array_equal(bw, fx[pre + "bow_words"])
                assert bv.tobytes() == fx[pre + "bow_values"].tobytes(), pre
                np.testing.assert_array_equal(fn, fx[f"{name}.fv_nodes"])
                np.testing.assert_array_equal(fs, fx[f"{name}.fv_start"])
                np.testing.assert_array_equal(fi, fx[f"{name}.fv_items"])
                cases += 1
    assert cases == 120


def test_bow_fixture_streams_are_what_the_descent_gives(oracle):
    """the "voc" streams of the fixture are the oracle's descent on the stored descriptors: regenerate and compare (drift guard;
    the descent itself has no reference pin, which is why the streams are stored)."""
    fx = _bow_fixture()
    for name in sorted({k.split(".")[0] for k in fx.files if k.startswith("voc")}):
        seed, k, depth, n, levelsup = (int(x) for x in fx[f"{name}.params"])
        voc = oracle.make_vocabulary(k, depth, seed=seed, p_stop=0.1)
        word, weight, node = oracle.bow_stream(voc, fx[f"{name}.desc"], levelsup)
        np.testing.assert_array_equal(word, fx[f"{name}.word"])
        assert weight.tobytes() == fx[f"{name}.weight"].tobytes()
        np.testing.assert_array_equal(node, fx[f"{name}.node"])
        # and compute_bow = stream + fold
        got = oracle.compute_bow(voc, fx[f"{name}.desc"], levelsup, 0, 1)
        ref = oracle.bow_fold(word, weight, node, 0, 1)
        for a, b in zip(got, ref):
            assert a.tobytes() == b.tobytes()


def test_bow_fold_against_the_reference_build_live(oracle):
    """where oracle/_ref exists (built by __graft_entry__.build() in the build container, shipped as a .so to the GPU box): fresh
    random streams through both, bit for bit."""
    if not oracle.ref_available():
        pytest.skip("oracle/_ref/libdbow2_fold.so not built (no reference tree on this machine)")
    rng = np.random.default_rng(99)
    for trial in range(40):
        n = int(rng.integers(0, 3000))
        nwords = int(rng.integers(1, 400))
        word = rng.integers(0, nwords, n).astype(np.uint32)
        weight = np.where(rng.random(n) < 0.1, 0.0, np.log(rng.uniform(1.01, 1e4, n)))
        node = rng.integers(0, int(rng.integers(1, 60)), n).astype(np.uint32)
        for w in range(4):
            for nm in range(3):
                a = oracle.bow_fold(word, weight, node, w, nm)
                b = oracle.ref_bow_fold(word, weight, node, w, nm)
                for x, y in zip(a, b):
                    assert x.tobytes() == y.tobytes(), (trial, w, nm)


# ---- the good-feature matchers: SearchByProjection_Budget / _OnePoint / GetCandidates / MatchCandidates -------------------------------
@pytest.mark.parametrize("seed,m,th,ratio", [(1, 1500, 1.0, 0.8), (2, 4000, 0.5, 0.8), (3, 3000, 3.0, 0.9)])
def test_budget_matcher_without_a_clock_is_the_plain_overload(oracle, seed, m, th, ratio):
    """SearchByProjection_Budget (ORBmatcher.cc:45-153) differs from SearchByProjection (:155-241) by IncreaseFound() and the clock
    only: with a clock that never trips the two restatements -- written separately -- must agree on every slot."""
    import gf_cases as gc
    kl, dl, u, _ = gc.frame(oracle)
    sf = oracle.OracleExtractor(2000, 1.2, 8, 20, 7).scale_factors
    mps, mpd, taken = gc.contended_map(oracle, kl, dl, seed, m)
    ref = oracle.search_by_projection(kl, dl, u, sf, gc.BOUNDS, mps, mpd, th, ratio, taken)
    nm, out_mp, out_sc, out_pt, found = oracle.search_by_projection_budget(kl, dl, u, sf, gc.BOUNDS, mps, mpd, th, ratio, taken, 0)
    assert nm == ref[0] and nm > 100
    np.testing.assert_array_equal(out_mp, ref[1])
    np.testing.assert_array_equal(out_sc, ref[2])
    np.testing.assert_array_equal(found, (out_pt >= 0).astype(np.int32))
    assert (out_pt == -1).any() and (out_pt == -2).any() and (out_pt == -3).any() and (out_pt > -4).all()
    # the same through the per-point entry the greedy selection calls (ORBmatcher.h:71-150), in vector order
    pf = oracle.ProjectionFrame(kl, dl, u, sf, gc.BOUNDS, taken)
    why_code = {0: None, 1: -2, 2: -3, 3: -1}
    for p in range(m):
        best, why = pf.one_point(mps[p], mpd[p], th, ratio, p)
        if best >= 0:
            assert out_pt[p] == (best | (pf.state()[1][best] << 16)) if p % 97 == 0 else (out_pt[p] & 0xFFFF) == best
        else:
            assert out_pt[p] == why_code[why]
    st = pf.state()
    np.testing.assert_array_equal(st[0], out_mp)
    np.testing.assert_array_equal(st[1], out_sc)


def test_budget_matcher_clock_cuts_a_prefix(oracle):
    """Whatever the clock does, the answer is the full answer's prefix up to the point whose clock reading tripped -- the property the
    device entry point rests on (one call for all points, gfo_projection_points_prefix for the cut)."""
    import gf_cases as gc
    kl, dl, u, _ = gc.frame(oracle)
    sf = oracle.OracleExtractor(2000, 1.2, 8, 20, 7).scale_factors
    mps, mpd, taken = gc.contended_map(oracle, kl, dl, 5, 2500)
    full = oracle.search_by_projection_budget(kl, dl, u, sf, gc.BOUNDS, mps, mpd, 1.0, 0.8, taken, 0)
    for k in (1, 2, 17, 300, 100000):
        nm, out_mp, out_sc, out_pt, found = oracle.search_by_projection_budget(kl, dl, u, sf, gc.BOUNDS, mps, mpd, 1.0, 0.8, taken, k)
        cut = gc.clock_cut(full[3], k)
        np.testing.assert_array_equal(out_pt[:cut], full[3][:cut])
        assert (out_pt[cut:] == -4).all()
        cnt, pm, ps = gc.prefix_state(full[3], cut, len(kl))
        assert cnt == nm
        np.testing.assert_array_equal(out_mp, pm)
        np.testing.assert_array_equal(out_sc, ps)
    assert gc.clock_cut(full[3], 1) < 20


def test_candidate_lists_then_match_equals_one_point(oracle):
    """GetCandidates + MatchCandidates (ORBmatcher.h:152-250, the INFORMATION_EFFICIENCY_SCORE form) = SearchByProjection_OnePoint, in any
    order of the points: two frames driven side by side in a shuffled order."""
    import gf_cases as gc
    kl, dl, u, _ = gc.frame(oracle)
    sf = oracle.OracleExtractor(2000, 1.2, 8, 20, 7).scale_factors
    mps, mpd, taken = gc.contended_map(oracle, kl, dl, 9, 1200)
    a = oracle.ProjectionFrame(kl, dl, u, sf, gc.BOUNDS, taken)
    b = oracle.ProjectionFrame(kl, dl, u, sf, gc.BOUNDS, taken)
    cands = [b.candidates(mps[p], 1.0) for p in range(len(mps))]
    assert max(len(c) for c in cands) > 5
    for p in np.random.default_rng(0).permutation(len(mps)):
        ra, _ = a.one_point(mps[p], mpd[p], 1.0, 0.8, int(p))
        rb = b.match_candidates(mps[p], mpd[p], cands[p], 1.0, 0.8, int(p))
        assert ra == rb
    np.testing.assert_array_equal(a.state()[0], b.state()[0])
    np.testing.assert_array_equal(a.state()[1], b.state()[1])
    # a candidate list is GetFeaturesInArea's answer
    for p in (0, 5, 77):
        if mps["flags"][p] & 1 and not mps["flags"][p] & 2:
            r = (2.5 if mps["view_cos"][p] > 0.998 else 4.0) * sf[mps["level"][p]]
            ref = oracle.features_in_area(kl, gc.BOUNDS, mps["proj_x"][p], mps["proj_y"][p], np.float32(r), int(mps["level"][p]) - 1, int(mps["level"][p]))
            np.testing.assert_array_equal(cands[p], ref)


def bow_keyframes_python(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori):
    """ORBmatcher.cc:635-768 once more, in Python, written from the reference's text (not from the C restatement)."""
    n1 = len(d1)
    out = np.full(n1, -1, np.int64)
    matched2 = np.zeros(len(d2), bool)
    hist = [[] for _ in range(30)]
    nm = 0
    ids1, st1, it1 = fv1
    ids2, st2, it2 = fv2
    pos2 = {int(k): j for j, k in enumerate(ids2)}
    for a, node in enumerate(ids1):
        b = pos2.get(int(node))
        if b is None:
            continue
        for idx1 in it1[st1[a]:st1[a + 1]]:
            if not v1[idx1]:
                continue
            best1, best2, bidx = 256, 256, -1
            for idx2 in it2[st2[b]:st2[b + 1]]:
                if matched2[idx2] or not v2[idx2]:
                    continue
                dist = int(np.unpackbits(d1[idx1] ^ d2[idx2]).sum())
                if dist < best1:
                    best2, best1, bidx = best1, dist, int(idx2)
                elif dist < best2:
                    best2 = dist
            if best1 < 50 and np.float32(best1) < np.float32(ratio) * np.float32(best2):
                out[idx1] = bidx
                matched2[bidx] = True
                if ori:
                    rot = np.float32(a1[idx1]) - np.float32(a2[bidx])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360.0))
                    b_ = int(math.floor(float(np.float32(rot * np.float32(1.0 / 30))) + 0.5))    # C round(): halves away from zero
                    if b_ == 30:
                        b_ = 0
                    hist[b_].append(int(idx1))
                nm += 1
    if ori:
        keep = oracle_three_maxima([len(h) for h in hist])
        for i, h in enumerate(hist):
            if i in keep:
                continue
            for idx1 in h:
                out[idx1] = -1
                nm -= 1
    return nm, out


def oracle_three_maxima(sizes):
    from oracle import orb_oracle
    return set(orb_oracle.three_maxima(sizes))


@pytest.mark.parametrize("seed,shift,ratio,ori", [(0, 3, 0.7, True), (1, 5, 0.9, False), (2, 4, 0.95, True)])
def test_bow_keyframe_pair_oracle_against_python_statement(oracle, seed, shift, ratio, ori):
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)[:700]
    d1 = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)[:700]
    rng = np.random.default_rng(seed)
    d2 = d1.copy()
    for _ in range(24):
        sel = rng.random(len(d2)) < 0.5
        bits = rng.integers(0, 256, len(d2))
        d2[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    perm = rng.permutation(len(d2))
    d2 = d2[perm]
    a1 = kl["angle"].copy()
    a2 = ((a1[perm] + rng.normal(0, 8, len(d2))) % 360).astype(np.float32)
    v1 = (rng.random(len(d1)) > 0.1).astype(np.uint8)
    v2 = (rng.random(len(d2)) > 0.15).astype(np.uint8)
    fv1 = oracle.make_feature_vector((d1[:, 0] >> shift).astype(np.int64))
    fv2 = oracle.make_feature_vector((d2[:, 0] >> shift).astype(np.int64))
    ref = bow_keyframes_python(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori)
    got = oracle.search_by_bow_keyframes(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio, ori)
    assert got[0] == ref[0] and ref[0] > 30
    np.testing.assert_array_equal(got[1], ref[1])


def triangulation_python(c, sf, sg, only_stereo, ori, mono):
    """ORBmatcher.cc:770-935 (+ :251-268) once more, in Python, written from the reference's text (not from the C restatement)."""
    f = np.float32
    kp1, kp2, d1, d2 = c["kp1"], c["kp2"], c["desc1"], c["desc2"]
    F = c["f12"].reshape(3, 3)
    n1 = len(kp1)
    out = np.full(n1, -1, np.int64)
    hist = [[] for _ in range(30)]
    nm = 0
    ids1, st1, it1 = c["fv1"]
    ids2, st2, it2 = c["fv2"]
    pos2 = {int(k): j for j, k in enumerate(ids2)}

    def epipolar_ok(k1, k2):
        a = f(f(f(k1["x"] * F[0, 0]) + f(k1["y"] * F[1, 0])) + F[2, 0])
        b = f(f(f(k1["x"] * F[0, 1]) + f(k1["y"] * F[1, 1])) + F[2, 1])
        cc = f(f(f(k1["x"] * F[0, 2]) + f(k1["y"] * F[1, 2])) + F[2, 2])
        num = f(f(f(a * k2["x"]) + f(b * k2["y"])) + cc)
        den = f(f(a * a) + f(b * b))
        if den == 0:
            return False
        dsqr = f(f(num * num) / den)
        return float(dsqr) < 3.84 * float(sg[k2["octave"]])
    for a, node in enumerate(ids1):
        b = pos2.get(int(node))
        if b is None:
            continue
        for idx1 in it1[st1[a]:st1[a + 1]]:
            if c["has1"][idx1]:
                continue
            s1 = (not mono) and c["ur1"][idx1] >= 0
            if only_stereo and not s1:
                continue
            best, bidx = 50, -1
            for idx2 in it2[st2[b]:st2[b + 1]]:
                if c["has2"][idx2]:
                    continue
                s2 = (not mono) and c["ur2"][idx2] >= 0
                if only_stereo and not s2:
                    continue
                dist = int(np.unpackbits(d1[idx1] ^ d2[idx2]).sum())
                if dist > 50 or dist > best:
                    continue
                if not s1 and not s2:
                    dx, dy = f(c["ex"] - kp2["x"][idx2]), f(c["ey"] - kp2["y"][idx2])
                    if f(f(dx * dx) + f(dy * dy)) < f(f(100) * sf[kp2["octave"][idx2]]):
                        continue
                if epipolar_ok(kp1[idx1], kp2[idx2]):
                    best, bidx = dist, int(idx2)
            if bidx >= 0:
                out[idx1] = bidx
                nm += 1
                if ori:
                    rot = f(kp1["angle"][idx1]) - f(kp2["angle"][bidx])
                    if rot < 0:
                        rot = f(rot + f(360.0))
                    b_ = int(math.floor(float(f(rot * f(1.0 / 30))) + 0.5))
                    if b_ == 30:
                        b_ = 0
                    hist[b_].append(int(idx1))
    if ori:
        keep = oracle_three_maxima([len(h) for h in hist])
        for i, h in enumerate(hist):
            if i in keep:
                continue
            for idx1 in h:
                out[idx1] = -1
                nm -= 1
    return nm, out


@pytest.mark.parametrize("seed,only_stereo,ori,mono,forward", [(0, False, True, False, False), (1, True, False, False, False), (2, False, True, True, True)])
def test_triangulation_oracle_against_python_statement(oracle, seed, only_stereo, ori, mono, forward):
    import gf_cases
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), oracle.KEYPOINT_DTYPE)[:500]
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)[:500]
    c = gf_cases.triangulation_case(oracle, kl, dl, np.random.default_rng(seed), node_shift=5, forward=forward, noise=1.0)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    sg = (sf * sf).astype(np.float32)
    ref = triangulation_python(c, sf, sg, only_stereo, ori, mono)
    got = oracle.search_for_triangulation(c["kp1"], c["desc1"], c["has1"], None if mono else c["ur1"], c["fv1"], c["kp2"], c["desc2"], c["has2"],
                                          None if mono else c["ur2"], c["fv2"], sf, sg, c["f12"], c["ex"], c["ey"], only_stereo, ori)
    assert got[0] == ref[0] and ref[0] > (20 if only_stereo else 100)
    np.testing.assert_array_equal(got[1], ref[1])


def initialization_python(oracle, kp1, d1, prev, kp2, d2, bounds, window, ratio, ori):
    """ORBmatcher.cc:520-633 once more, in Python, from the reference's text; the window through features_in_area (tested on its own)."""
    f = np.float32
    n1, n2 = len(kp1), len(kp2)
    m12 = np.full(n1, -1, np.int64)
    mdist = np.full(n2, 2 ** 31 - 1, np.int64)
    m21 = np.full(n2, -1, np.int64)
    hist = [[] for _ in range(30)]
    nm = 0
    thefts = 0
    for i1 in range(n1):
        if kp1["octave"][i1] > 0:
            continue
        idx = oracle.features_in_area(kp2, bounds, prev[i1, 0], prev[i1, 1], f(window), 0, 0)
        best, best2, bidx = 2 ** 31 - 1, 2 ** 31 - 1, -1
        for i2 in idx:
            dist = int(np.unpackbits(d1[i1] ^ d2[i2]).sum())
            if mdist[i2] <= dist:
                continue
            if dist < best:
                best2, best, bidx = best, dist, int(i2)
            elif dist < best2:
                best2 = dist
        if best <= 50 and f(best) < f(f(best2) * f(ratio)):
            if m21[bidx] >= 0:
                m12[m21[bidx]] = -1
                nm -= 1
                thefts += 1
            m12[i1], m21[bidx], mdist[bidx] = bidx, i1, best
            nm += 1
            if ori:
                rot = f(kp1["angle"][i1]) - f(kp2["angle"][bidx])
                if rot < 0:
                    rot = f(rot + f(360.0))
                b_ = int(math.floor(float(f(rot * f(1.0 / 30))) + 0.5))
                hist[0 if b_ == 30 else b_].append(i1)
    if ori:
        keep = oracle_three_maxima([len(h) for h in hist])
        for i, h in enumerate(hist):
            if i in keep:
                continue
            for i1 in h:
                if m12[i1] >= 0:
                    m12[i1] = -1
                    nm -= 1
    for i1 in range(n1):
        if m12[i1] >= 0:
            prev[i1] = (kp2["x"][m12[i1]], kp2["y"][m12[i1]])
    return nm, m12, thefts


@pytest.mark.parametrize("seed,window,ratio,ori", [(0, 100, 0.9, True), (1, 40, 0.8, False), (2, 300, 0.9, True)])
def test_initialization_oracle_against_python_statement(oracle, seed, window, ratio, ori):
    import gf_cases
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), oracle.KEYPOINT_DTYPE)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    keep = np.flatnonzero(kl["octave"] <= 1)[:450]                      # level 0 searches; level 1 must be ignored on both sides
    kl, dl = kl[keep], dl[keep]
    kp2, d2, prev = gf_cases.initialization_case(oracle, kl, dl, np.random.default_rng(seed), flips=10, sigma=12.0)
    bounds = (0.0, 0.0, 752.0, 480.0)
    p_ref, p_got = prev.copy(), prev.copy()
    ref = initialization_python(oracle, kl, dl, p_ref, kp2, d2, bounds, window, ratio, ori)
    got = oracle.search_for_initialization(kl, dl, p_got, kp2, d2, bounds, window, ratio, ori)
    assert got[0] == ref[0] and ref[0] > 40
    np.testing.assert_array_equal(got[1], ref[1])
    assert p_got.tobytes() == p_ref.tobytes() and p_got.tobytes() != prev.tobytes()
    assert ref[2] > 0                                                    # keypoints of F2 did change hands


def test_oracle_reproduces_golden_good_feature_matchers(oracle):
    """tests/golden/EuRoC_gf_matchers.npz (made by tests/golden/make_gf_golden.py): the oracle still says what it said when the vectors
    were committed -- SearchByProjection_Budget at th 0.5 / 1 / with a clock, GetCandidates for every point, SearchByBoW(KF, KF)."""
    kd = oracle.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), kd)
    kr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_kp.bin"), kd)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    dr = np.fromfile(os.path.join(GOLDEN, "EuRoC_r_desc.bin"), np.uint8).reshape(-1, 32)
    u = np.load(os.path.join(GOLDEN, "EuRoC_stereo.npz"))["u_right"]
    p = np.load(os.path.join(GOLDEN, "EuRoC_projection.npz"))
    g = np.load(os.path.join(GOLDEN, "EuRoC_gf_matchers.npz"))
    sf = oracle.OracleExtractor().scale_factors
    b = (0.0, 0.0, 752.0, 480.0)
    for tag, th in (("th05", 0.5), ("th1", 1.0)):
        nm, out_mp, out_sc, out_pt, found = oracle.search_by_projection_budget(kl, dl, u, sf, b, p["mps"], p["mp_desc"], th, 0.8, p["taken"], 0)
        assert nm == int(g[f"{tag}_nmatches"])
        for a, name in ((out_mp, "out_mp"), (out_sc, "out_score"), (out_pt, "out_point"), (found, "found")):
            np.testing.assert_array_equal(a, g[f"{tag}_{name}"], err_msg=f"{tag} {name}")
    nm5, mp5, sc5, _, _ = oracle.search_by_projection_budget(kl, dl, u, sf, b, p["mps"], p["mp_desc"], 1.0, 0.8, p["taken"], 5)
    assert nm5 == int(g["th1_trip5_nmatches"])
    np.testing.assert_array_equal(mp5, g["th1_trip5_out_mp"]); np.testing.assert_array_equal(sc5, g["th1_trip5_out_score"])
    pf = oracle.ProjectionFrame(kl, dl, u, sf, b, None)
    st = g["th1_cand_start"]
    for i in range(0, len(p["mps"]), 7):
        np.testing.assert_array_equal(pf.candidates(p["mps"][i], 1.0), g["th1_cand_idx"][st[i]:st[i + 1]])
    n1 = (dl[:, 0] >> 2).astype(np.int64); n2 = (dr[:, 0] >> 2).astype(np.int64)
    for ori in (0, 1):
        nmk, o12 = oracle.search_by_bow_keyframes(dl, kl["angle"], g["bowkf_valid1"], oracle.make_feature_vector(n1), dr, kr["angle"], g["bowkf_valid2"],
                                                  oracle.make_feature_vector(n2), 0.75, bool(ori))
        assert nmk == int(g[f"bowkf_ori{ori}_nmatches"])
        np.testing.assert_array_equal(o12, g[f"bowkf_ori{ori}_out12"])
