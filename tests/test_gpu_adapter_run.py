"""The C++ drop-in adapters EXECUTED on the GPU: tests/_build/adapter_run = adapter/ORBextractor_gfo.cc +
adapter/matchers_gfo.cc compiled against the reference's UNCHANGED headers (include/ORBextractor.h, Frame.h, MapPoint.h,
ORBmatcher.h), linked with libgfo.so, driven the way Frame.cc / Tracking.cc drive the reference's classes
(tests/host/adapter_run.cc lists the scenarios).  This file makes its inputs from the oracle's keypoints, runs the program once
and compares everything it wrote with the oracle, bit for bit.

The binary is built by __graft_entry__.build() where the reference headers are mounted (the build container) and travels to the
GPU box with the snapshot, like libgfo.so."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLDEN

pytestmark = pytest.mark.gpu

EXE = os.path.join(ROOT, "tests", "_build", "adapter_run")
FX, FY, CX, CY = 435.2046959714599, 435.2046959714599, 367.4517211914062, 252.2008514404297
MBF = np.float32(47.906)
MB = np.float32(np.float32(47.906) / np.float32(435.2))
NF = 20
BUDGET = 150            # MAX_NUM_FEATURE_MATCHING of include/ORBmatcher.h:37, what tests/host/Makefile gives adapter_run_budget
f32 = np.float32


def _frames(img, f):
    return np.ascontiguousarray(np.roll(img, -3 * f, axis=1))


def _mm(R, X):
    """(3x3) x (n x 3)^T as the plain float loops of the cv stand-in: s = 0; s += R[i,k] * x[k] for k = 0, 1, 2."""
    out = np.zeros((len(X), 3), f32)
    for i in range(3):
        s = np.zeros(len(X), f32)
        for k in range(3):
            s = (s + (f32(R[i, k]) * X[:, k]).astype(f32)).astype(f32)
        out[:, i] = s
    return out


def _pose(rx, ry, rz, t):
    cx_, sx = np.cos(rx), np.sin(rx)
    cy_, sy = np.cos(ry), np.sin(ry)
    cz, sz = np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx_, -sx], [0, sx, cx_]])
    Ry = np.array([[cy_, 0, sy], [0, 1, 0], [-sy, 0, cy_]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = t
    return T.astype(f32)


@pytest.fixture(scope="module")
def run(tmp_path_factory, oracle, euroc_l, euroc_r):
    if not os.path.exists(EXE):
        if os.path.exists("/root/reference/include/ORBextractor.h"):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "host")])
        else:
            # (a skip, not a failure: the suite runs with -x and everything behind this file would go unreported; the skip reason is
            #  in the report, and GPUTEST's pass count shows these eleven tests missing)
            pytest.skip("tests/_build/adapter_run is missing and the reference headers are not mounted here: "
                        "__graft_entry__.build() makes the binary in the build container and it travels with the snapshot")
    ind, outd = tmp_path_factory.mktemp("adapter_in"), tmp_path_factory.mktemp("adapter_out")
    oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    sf = oe.scale_factors
    ref = {"sf": sf, "frames": []}
    for f in range(NF):
        kl, dl = oe(_frames(euroc_l, f))
        kr, dr = oe(_frames(euroc_r, f))
        st = oracle.stereo_match(kl, dl, kr, dr, sf, 480, MBF, MB, 0.0)
        ref["frames"].append((kl, dl, kr, dr, st))
    rng = np.random.default_rng(5)

    # D: frame 2 carries map points -> per-keypoint disparity windows (Frame.cc:1220-1231)
    kl, dl, kr, dr, st = ref["frames"][2]
    n = len(kl)
    T = _pose(0.01, -0.02, 0.015, [0.05, -0.03, 0.1])
    has = rng.random(n) < 0.5
    bad = has & (rng.random(n) < 0.1)
    disp = rng.uniform(1.0, 130.0, n)
    z = (47.906 / disp)
    z[rng.random(n) < 0.05] *= -1                     # behind the camera: WorldToCameraPoint says no, the default window stays
    pw = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1, 1, n), z], 1).astype(f32)
    rec = np.zeros(n, np.dtype([("has", "<i4"), ("bad", "<i4"), ("p", "<f4", 3)]))
    rec["has"], rec["bad"], rec["p"] = has, bad, pw
    rec.tofile(ind / "D_windows.bin")
    T.tofile(ind / "D_pose.bin")
    pc = (_mm(T[:3, :3], pw) + T[:3, 3][None, :]).astype(f32)          # mRcw * Pw + mtcw
    min_d = np.zeros(n, f32)
    max_d = np.full(n, MBF / MB, f32)
    ok = has & ~bad & (pc[:, 2] > 0)
    d = (MBF / pc[:, 2]).astype(f32)
    min_d[ok] = np.maximum((d - f32(50.0)).astype(f32), f32(0))[ok]
    max_d[ok] = np.minimum((d + f32(50.0)).astype(f32), f32(MBF / MB))[ok]
    # D is the SECOND call on frame 2 (its first was part A's, without windows): the literal member state of the oracle
    SF = oracle.StereoFrame(kl, dl, kr, dr, sf, 480, MBF, MB)
    ref["D_first"] = SF.match()
    ref["D"] = SF.match(min_d, max_d, has.astype(np.uint8))
    ref["D3"] = SF.match(min_d, max_d, has.astype(np.uint8), online=True)
    ref["D2"] = oracle.stereo_match(kl, dl, kr, dr, sf, 480, MBF, MB, 0.0, min_d, max_d)        # the same windows on a fresh frame
    SF4 = oracle.StereoFrame(kl, dl, kr, dr, sf, 480, MBF, MB)
    SF4.match(min_d, max_d, has.astype(np.uint8))
    SF4.prepare()
    ref["D4"] = SF4.match(min_d, max_d, has.astype(np.uint8))
    ref["D_windows_used"] = int(ok.sum())
    # DD: the same frame as a DELAYED_STEREO_MATCHING build drives it (tests/_build/adapter_run_delayed): map points arrive in two
    # stages, an online call after each, then two offline calls (Frame.cc:1186-1199)
    stage = np.where(has, np.where(rng.random(n) < 0.6, 1, 2), 0).astype(np.int32)
    stage.tofile(ind / "DD_stage.bin")
    SD = oracle.StereoFrame(kl, dl, kr, dr, sf, 480, MBF, MB, delayed=True)
    SD.prepare()
    ref["DD"] = []
    for has_now, online in ((stage == 1, True), (stage >= 1, True), (stage >= 1, False), (stage >= 1, False)):
        ref["DD"].append(SD.match(min_d, max_d, has_now.astype(np.uint8), online=online) + (SD.matched(),))

    # G: the online call on frame 0 (no outlier cut)
    kl, dl = ref["frames"][0][0], ref["frames"][0][1]
    kr, dr = oe(np.ascontiguousarray(np.roll(euroc_l, -10, axis=1)))         # the left image shifted: disparity 10, small distances
    ref["G"] = [oracle.stereo_match(kl, dl, kr, dr, sf, 480, MBF, MB, 0.0, online=on) for on in (False, True)]

    # E: SearchByProjection(F, local map, 3) on frame 1
    kl, dl, kr, dr, st = ref["frames"][1]
    n, m = len(kl), 3000
    mps = np.zeros(m, oracle.MAP_POINT_DTYPE)
    src = rng.integers(0, n, m)
    mpd = dl[src].copy()
    nflip = rng.integers(0, 40, m)
    for j in range(m):
        for b in rng.integers(0, 256, nflip[j]):
            mpd[j, b >> 3] ^= np.uint8(1 << (b & 7))
    mpd[2500:] = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    mps["proj_x"] = kl["x"][src] + rng.normal(0, 2, m)
    mps["proj_y"] = kl["y"][src] + rng.normal(0, 2, m)
    mps["proj_xr"] = np.where(st[1][src] >= 0, st[1][src] + rng.normal(0, 1.5, m), mps["proj_x"] - 10)
    mps["level"] = np.clip(kl["octave"][src] + rng.integers(-1, 2, m), 0, 7)
    mps["view_cos"] = rng.choice([1.0, 0.9985, 0.99], m)
    fl = np.full(m, 1 | 4, np.int32)
    fl[rng.random(m) < 0.05] = 4
    fl[rng.random(m) < 0.04] |= 2
    fl[rng.random(m) < 0.2] &= ~4
    mps["flags"] = fl
    tk = np.zeros(n, np.uint8)
    r = rng.random(n)
    tk[r < 0.08] = 1
    tk[(r >= 0.08) & (r < 0.12)] = 2
    mps.tofile(ind / "E_map.bin")
    mpd.tofile(ind / "E_map_desc.bin")
    tk.tofile(ind / "E_taken.bin")
    ref["E"] = oracle.search_by_projection(kl, dl, st[1], sf, (0.0, 0.0, 752.0, 480.0), mps, mpd, 3.0, 0.8, (tk == 1).astype(np.uint8))
    # E2: SearchByProjection_Budget on the same map at Tracking.cc:2166's th = 0.5: a clock that never trips / trips at its first reading
    ref["E2"] = [oracle.search_by_projection_budget(kl, dl, st[1], sf, (0.0, 0.0, 752.0, 480.0), mps, mpd, 0.5, 0.8, (tk == 1).astype(np.uint8), trip)
                 for trip in (0, 1)]

    # E3: SearchByProjection_OnePoint pick by pick in a made-up order (2000 of the 3000 points, shuffled; entry 7 of the vector is NULL)
    order = rng.permutation(np.setdiff1d(np.arange(m), [7]))[:2000].astype(np.int32)
    order.tofile(ind / "E3_order.bin")
    pf = oracle.ProjectionFrame(kl, dl, st[1], sf, (0.0, 0.0, 752.0, 480.0), (tk == 1).astype(np.uint8))
    ncand = np.array([0 if j == 7 else len(pf.candidates(mps[j], 1.0)) for j in range(m)], np.int32)
    res = np.array([pf.one_point(mps[j], mpd[j], 1.0, 0.8, int(j))[0] for j in order], np.int32)
    ref["E3"] = (order, res, ncand) + pf.state()

    # F: SearchByProjection(Cur = frame 1, Last = frame 0): the adapter projects on the host (ORBmatcher.cc:1451-1502), so this side
    # states the same float expressions independently
    kl0, dl0 = ref["frames"][0][0], ref["frames"][0][1]
    kl1, dl1, _, _, st1 = ref["frames"][1]
    n0 = len(kl0)
    ref["F"] = []
    variants = [(7.0, 1.0, [0.0, 0.0, 0.0]), (7.0, 1.0, [0.0, 0.0, 0.5]), (15.0, 1.0, [0.0, 0.0, -0.5]), (7.0, 0.0, [0.02, 0.0, 0.01])]
    for v, (th, ori, dt) in enumerate(variants):
        Tc = _pose(0.004, -0.006, 0.003, [0.01, -0.02, 0.03])
        Tl = Tc.copy()
        Tl[:3, 3] = (Tl[:3, 3] + np.asarray(dt, f32)).astype(f32)
        has = rng.random(n0) < 0.7
        outl = rng.random(n0) < 0.06
        obs = np.where(rng.random(n0) < 0.85, 3, 0).astype(np.int32)
        zc = rng.uniform(1.5, 25.0, n0)
        zc[rng.random(n0) < 0.03] *= -1
        # world points that project near where the keypoint moved to in frame 1 (the image rotates left by 3 columns)
        uc = kl0["x"] - 3.0 + rng.normal(0, 1.5, n0)
        vc = kl0["y"] + rng.normal(0, 1.5, n0)
        uc[rng.random(n0) < 0.03] += 900.0            # outside the image bounds
        pcam = np.stack([(uc - CX) / FX * zc, (vc - CY) / FY * zc, zc], 1)
        Rc, tc = Tc[:3, :3].astype(np.float64), Tc[:3, 3].astype(np.float64)
        pw = ((pcam - tc[None, :]) @ Rc).astype(f32)                       # Rc^T (pc - tc)
        qd = dl0.copy()
        for j in range(n0):
            for b in rng.integers(0, 256, rng.integers(0, 50)):
                qd[j, b >> 3] ^= np.uint8(1 << (b & 7))
        rec = np.zeros(n0, np.dtype([("has", "<i4"), ("outlier", "<i4"), ("obs", "<i4"), ("p", "<f4", 3)]))
        rec["has"], rec["outlier"], rec["obs"], rec["p"] = has, outl, obs, pw
        rec.tofile(ind / f"F{v}_last.bin")
        qd.tofile(ind / f"F{v}_last_desc.bin")
        np.concatenate([Tc.ravel(), Tl.ravel(), np.array([FX, FY, CX, CY, th, ori], f32)]).astype(f32).tofile(ind / f"F{v}_calib.bin")
        # --- what the adapter computes on the host, restated (every operation rounded to float like the C++ expressions)
        Rcw, tcw = Tc[:3, :3], Tc[:3, 3]
        Rlw, tlw = Tl[:3, :3], Tl[:3, 3]
        twc = _mm((-(Rcw.T)).astype(f32), tcw[None, :])[0]                   # -Rcw.t() * tcw
        tlc = (_mm(Rlw, twc[None, :])[0] + tlw).astype(f32)
        fwd, bwd = bool(tlc[2] > MB), bool(-tlc[2] > MB)
        x3 = (_mm(Rcw, pw) + tcw[None, :]).astype(f32)
        with np.errstate(divide="ignore"):
            invz = (1.0 / x3[:, 2].astype(np.float64)).astype(f32)
        fx, fy, cx, cy = f32(FX), f32(FY), f32(CX), f32(CY)
        u = ((((fx * x3[:, 0]).astype(f32)) * invz).astype(f32) + cx).astype(f32)
        vv = ((((fy * x3[:, 1]).astype(f32)) * invz).astype(f32) + cy).astype(f32)
        keep = has & ~outl & ~(invz < 0) & ~(u < 0) & ~(u > 752) & ~(vv < 0) & ~(vv > 480)
        qi = np.nonzero(keep)[0]
        q = np.zeros(len(qi), oracle.PROJ_QUERY_DTYPE)
        octv = kl0["octave"][qi]
        q["u"], q["v"] = u[qi], vv[qi]
        q["ur"] = (u[qi] - (MBF * invz[qi]).astype(f32)).astype(f32)
        q["radius"] = (f32(th) * sf[octv]).astype(f32)
        if fwd:
            q["min_level"], q["max_level"] = octv, -1
        elif bwd:
            q["min_level"], q["max_level"] = 0, octv
        else:
            q["min_level"], q["max_level"] = octv - 1, octv + 1
        q["angle"] = kl0["angle"][qi]
        q["flags"] = 1 | np.where(obs[qi] > 0, 4, 0)
        nm, out_q, _ = oracle.search_by_projection_queries(kl1, dl1, st1[1], kl1["angle"], (0.0, 0.0, 752.0, 480.0), q, qd[qi], False, 0.0,
                                                           100, bool(ori), np.zeros(len(kl1), np.uint8))
        ref["F"].append({"nm": nm, "idx": np.where(out_q >= 0, qi[np.maximum(out_q, 0)], -1).astype(np.int32), "visible": len(qi),
                         "fwd": fwd, "bwd": bwd})
        with oracle.feature_budget(BUDGET):        # the same call in a BUDGETING_FEATURE_MATCHING build (tests/_build/adapter_run_budget)
            nmb, out_qb, _ = oracle.search_by_projection_queries(kl1, dl1, st1[1], kl1["angle"], (0.0, 0.0, 752.0, 480.0), q, qd[qi], False, 0.0,
                                                                 100, bool(ori), np.zeros(len(kl1), np.uint8))
        ref.setdefault("F_budget", []).append({"nm": nmb, "idx": np.where(out_qb >= 0, qi[np.maximum(out_qb, 0)], -1).astype(np.int32)})
    # H: SearchByProjection(Cur = frame 1, KeyFrame = frame 0, sAlreadyFound, th, ORBdist): the adapter's host side restated
    # (ORBmatcher.cc:1607-1650; PredictScale / the distance range hand back what the harness stored, adapter_link_support.cc)
    th_h, orbdist, ori_h = 10.0, 90, 1.0
    Tc = _pose(-0.005, 0.004, 0.002, [0.02, 0.01, -0.03])
    has = rng.random(n0) < 0.75
    badk = has & (rng.random(n0) < 0.05)
    foundk = has & (rng.random(n0) < 0.1)
    zc = rng.uniform(1.5, 25.0, n0)
    uc = kl0["x"] - 3.0 + rng.normal(0, 2.0, n0)
    vc = kl0["y"] + rng.normal(0, 2.0, n0)
    uc[rng.random(n0) < 0.03] -= 900.0
    pcam = np.stack([(uc - CX) / FX * zc, (vc - CY) / FY * zc, zc], 1)
    Rc, tc = Tc[:3, :3].astype(np.float64), Tc[:3, 3].astype(np.float64)
    pw = ((pcam - tc[None, :]) @ Rc).astype(f32)
    lev = np.clip(kl0["octave"] + rng.integers(-1, 2, n0), 0, 7).astype(np.int32)
    Rcw, tcw = Tc[:3, :3], Tc[:3, 3]
    Ow = _mm((-(Rcw.T)).astype(f32), tcw[None, :])[0]                     # -Rcw.t() * tcw
    PO = (pw - Ow[None, :]).astype(f32)
    dist3 = np.sqrt((PO.astype(np.float64) ** 2)[:, 0] + (PO.astype(np.float64) ** 2)[:, 1] + (PO.astype(np.float64) ** 2)[:, 2]).astype(f32)   # cv::norm -> float
    dmin = (dist3 * rng.choice([0.5, 0.9, 1.2], n0, p=[0.5, 0.4, 0.1])).astype(f32)      # some points out of their distance range
    dmax = (dist3 * rng.choice([2.0, 1.1, 0.8], n0, p=[0.5, 0.4, 0.1])).astype(f32)
    qd = dl0.copy()
    for j in range(n0):
        for b in rng.integers(0, 256, rng.integers(0, 45)):
            qd[j, b >> 3] ^= np.uint8(1 << (b & 7))
    rec = np.zeros(n0, np.dtype([("has", "<i4"), ("bad", "<i4"), ("found", "<i4"), ("level", "<i4"), ("p", "<f4", 3), ("dmin", "<f4"), ("dmax", "<f4")]))
    rec["has"], rec["bad"], rec["found"], rec["level"], rec["p"], rec["dmin"], rec["dmax"] = has, badk, foundk, lev, pw, dmin, dmax
    rec.tofile(ind / "H_kf.bin")
    qd.tofile(ind / "H_kf_desc.bin")
    np.concatenate([Tc.ravel(), np.array([FX, FY, CX, CY, th_h, orbdist, ori_h], f32)]).astype(f32).tofile(ind / "H_calib.bin")
    n1 = len(kl1)
    tkh = (rng.random(n1) < 0.1).astype(np.uint8)
    tkh.tofile(ind / "H_taken.bin")
    x3 = (_mm(Rcw, pw) + tcw[None, :]).astype(f32)
    with np.errstate(divide="ignore"):
        invz = (1.0 / x3[:, 2].astype(np.float64)).astype(f32)
    fx, fy, cx, cy = f32(FX), f32(FY), f32(CX), f32(CY)
    u = ((((fx * x3[:, 0]).astype(f32)) * invz).astype(f32) + cx).astype(f32)
    vv = ((((fy * x3[:, 1]).astype(f32)) * invz).astype(f32) + cy).astype(f32)
    keep = has & ~badk & ~foundk & ~(u < 0) & ~(u > 752) & ~(vv < 0) & ~(vv > 480) & ~(dist3 < dmin) & ~(dist3 > dmax)
    qi = np.nonzero(keep)[0]
    q = np.zeros(len(qi), oracle.PROJ_QUERY_DTYPE)
    q["u"], q["v"], q["ur"] = u[qi], vv[qi], 0.0
    q["radius"] = (f32(th_h) * sf[lev[qi]]).astype(f32)
    q["min_level"], q["max_level"] = lev[qi] - 1, lev[qi] + 1
    q["angle"] = kl0["angle"][qi]
    q["flags"] = 1 | 4
    nm, out_q, _ = oracle.search_by_projection_kf(kl1, dl1, kl1["angle"], (0.0, 0.0, 752.0, 480.0), q, qd[qi], orbdist, bool(ori_h), tkh)
    ref["H"] = {"nm": nm, "idx": np.where(out_q >= 0, qi[np.maximum(out_q, 0)], -1).astype(np.int32), "queries": len(qi),
                "cleared": int((out_q == -2).sum())}       # (a slot this call wrote and its rotation check cleared is NULL again: -1)

    # L: loop closing's projection matchers under a similarity, into KeyFrame = frame 1.  Candidate points = frame 0's keypoints moved to where
    # they land in frame 1, as world points under Scw = s [R | t]; the adapter's host side restated with the stand-in's float semantics
    # (ORBmatcher.cc:415-477 = :1098-1161: decomposition, projection, distance range, viewing angle, predicted level, radius)
    mL = n0
    s_scale = f32(1.7)
    Tl = _pose(0.006, -0.004, 0.005, [0.03, -0.01, 0.02])
    Scw = Tl.copy()
    Scw[:3, :] = (Scw[:3, :] * s_scale).astype(f32)
    sR = Scw[:3, :3]
    scw = f32(np.sqrt(np.sum(sR[0].astype(np.float64) ** 2)))          # sqrt(row(0).dot(row(0))): a double sum, a float result
    RcwL = (sR / scw).astype(f32)
    tcwL = (Scw[:3, 3] / scw).astype(f32)
    OwL = _mm((-(RcwL.T)).astype(f32), tcwL[None, :])[0]
    zc = rng.uniform(1.5, 25.0, mL)
    zc[rng.random(mL) < 0.03] *= -1
    uc = kl0["x"] - 3.0 + rng.normal(0, 1.5, mL)
    vc = kl0["y"] + rng.normal(0, 1.5, mL)
    uc[rng.random(mL) < 0.03] += 900.0
    pcam = np.stack([(uc - CX) / FX * zc, (vc - CY) / FY * zc, zc], 1)
    pwL = ((pcam - tcwL.astype(np.float64)[None, :]) @ RcwL.astype(np.float64)).astype(f32)      # Rcw^T (pc - tcw)
    POL = (pwL - OwL[None, :]).astype(f32)
    dist3 = np.sqrt(np.sum(POL.astype(np.float64) ** 2, axis=1)).astype(f32)
    nrm = (POL / np.maximum(dist3, 1e-6)[:, None]).astype(f32)            # looking straight at the camera ...
    side = rng.random(mL) < 0.08                                          # ... except a few seen from the side (beyond 60 degrees)
    nrm[side] = np.stack([nrm[side, 1], -nrm[side, 0], np.zeros(int(side.sum()), f32)], 1).astype(f32)
    dminL = (dist3 * rng.choice([0.5, 0.9, 1.2], mL, p=[0.6, 0.35, 0.05])).astype(f32)
    dmaxL = (dist3 * rng.choice([2.0, 1.1, 0.8], mL, p=[0.6, 0.35, 0.05])).astype(f32)
    levL = np.clip(kl0["octave"] + rng.integers(0, 2, mL), 0, 7).astype(np.int32)
    badL = rng.random(mL) < 0.04
    qdL = dl0.copy()
    for j in range(mL):
        for b_ in rng.integers(0, 256, rng.integers(0, 40)):
            qdL[j, b_ >> 3] ^= np.uint8(1 << (b_ & 7))
    rec = np.zeros(mL, np.dtype([("bad", "<i4"), ("level", "<i4"), ("p", "<f4", 3), ("n", "<f4", 3), ("dmin", "<f4"), ("dmax", "<f4")]))
    rec["bad"], rec["level"], rec["p"], rec["n"], rec["dmin"], rec["dmax"] = badL, levL, pwL, nrm, dminL, dmaxL
    rec.tofile(ind / "L_points.bin")
    qdL.tofile(ind / "L_points_desc.bin")
    th_proj, th_fuse = 10, 4.0
    np.concatenate([Scw.ravel(), np.array([FX, FY, CX, CY, th_proj, th_fuse], f32)]).astype(f32).tofile(ind / "L_scw.bin")
    matched0 = np.where(rng.random(n1) < 0.08, rng.integers(0, mL, n1), -1).astype(np.int32)        # vpMatched on entry: candidate points (so they are "already found")
    matched0.tofile(ind / "L_matched.bin")
    kfmp0 = np.full(n1, -1, np.int32)
    r_ = rng.random(n1)
    kfmp0[r_ < 0.06] = rng.integers(0, mL, int((r_ < 0.06).sum()))                                  # the keyframe already sees some candidate points
    kfmp0[(r_ >= 0.06) & (r_ < 0.30)] = -2                                                           # foreign good points
    kfmp0[(r_ >= 0.30) & (r_ < 0.33)] = -3                                                           # foreign bad points
    kfmp0.tofile(ind / "L_kfmp.bin")
    x3 = (_mm(RcwL, pwL) + tcwL[None, :]).astype(f32)
    fx, fy, cx, cy = f32(FX), f32(FY), f32(CX), f32(CY)

    def visible(invz, th):
        x = (x3[:, 0] * invz).astype(f32); y = (x3[:, 1] * invz).astype(f32)
        u = ((fx * x).astype(f32) + cx).astype(f32); v = ((fy * y).astype(f32) + cy).astype(f32)
        dot = np.sum(POL.astype(np.float64) * nrm.astype(np.float64), axis=1)
        ok = ~(x3[:, 2] < 0) & (u >= 0) & (u < 752) & (v >= 0) & (v < 480) & ~(dist3 < dminL) & ~(dist3 > dmaxL) & ~(dot < 0.5 * dist3.astype(np.float64))
        return u, v, ok, (f32(th) * sf[levL]).astype(f32)

    with np.errstate(divide="ignore"):
        inv_f = (f32(1) / x3[:, 2]).astype(f32)                             # SearchByProjection: `1 / z` in float
        inv_d = (1.0 / x3[:, 2].astype(np.float64)).astype(f32)             # Fuse: `1.0 / z`, a double quotient rounded to float
    # L1
    u, v, ok, rad = visible(inv_f, th_proj)
    found0 = set(int(j) for j in matched0 if j >= 0)
    keep = ok & ~badL & ~np.isin(np.arange(mL), list(found0))
    qi = np.nonzero(keep)[0]
    q = np.zeros(len(qi), oracle.PROJ_QUERY_DTYPE)
    q["u"], q["v"], q["radius"] = u[qi], v[qi], rad[qi]
    q["min_level"], q["max_level"], q["flags"] = levL[qi] - 1, levL[qi], 1 | 4
    nmL, out_q, _ = oracle.search_by_projection_queries(kl1, dl1, None, None, (0.0, 0.0, 752.0, 480.0), q, qdL[qi], False, 0.0, 50, False,
                                                        (matched0 >= 0).astype(np.uint8))
    ref["L1"] = {"nm": nmL, "matched": np.where(out_q >= 0, qi[np.maximum(out_q, 0)], matched0).astype(np.int32), "queries": len(qi)}
    # L2
    u, v, ok, rad = visible(inv_d, th_fuse)
    seen = set(int(j) for j in kfmp0 if j >= 0)                            # pKF->GetMapPoints(): the good points the keyframe holds
    seen = set(j for j in seen if not badL[j])
    keep = ok & ~badL & ~np.isin(np.arange(mL), list(seen))
    qi = np.nonzero(keep)[0]
    q = np.zeros(len(qi), oracle.PROJ_QUERY_DTYPE)
    q["u"], q["v"], q["radius"] = u[qi], v[qi], rad[qi]
    q["min_level"], q["max_level"], q["flags"] = levL[qi] - 1, levL[qi], 1
    _, _, _, out_p = oracle.search_by_projection_queries_points(kl1, dl1, None, None, (0.0, 0.0, 752.0, 480.0), q, qdL[qi], False, 0.0, 50, False, None)
    kf_after = kfmp0.copy()
    kf_after[kfmp0 == -2] = -100 - np.nonzero(kfmp0 == -2)[0]
    kf_after[kfmp0 == -3] = -100 - np.nonzero(kfmp0 == -3)[0]
    bad_foreign = set(int(i) for i in np.nonzero(kfmp0 == -3)[0])
    repl = np.full(mL, -1, np.int32); obs_at = np.full(mL, -1, np.int32)
    nf = 0
    for k_, j in enumerate(qi):                                            # :1194-1208, point after point
        if out_p[k_] < 0:
            continue
        best = int(out_p[k_]) & 0xFFFF
        holder = int(kf_after[best])
        if holder != -1:
            is_bad = (holder <= -100 and (-100 - holder) in bad_foreign) or (holder >= 0 and bool(badL[holder]))
            if not is_bad:
                repl[j] = holder
        else:
            obs_at[j] = best
            kf_after[best] = j
        nf += 1
    ref["L2"] = {"nf": nf, "replace": repl, "kf_after": kf_after.astype(np.int32), "observed_at": obs_at, "queries": len(qi)}

    # L3: Fuse(KeyFrame = frame 1 posed at Tl, the same candidate records, th = 3): host side restated (ORBmatcher.cc:939-1004), the search by
    # the oracle (reprojection-error gate), the side effects (:1067-1083 with the link support's Replace) in the vector's order
    th3 = f32(3.0)
    Rk, tk = Tl[:3, :3], Tl[:3, 3]
    Owk = _mm((-(Rk.T)).astype(f32), tk[None, :])[0]
    np.concatenate([Tl.ravel(), Owk, np.array([th3], f32)]).astype(f32).tofile(ind / "L3_pose.bin")
    # (the candidate records were made for the similarity's Rcw / tcw; under Tl they land elsewhere -- some inside the image, which is all this needs)
    pcam3 = np.stack([(uc - CX) / FX * zc, (vc - CY) / FY * zc, zc], 1)
    pw3 = ((pcam3 - tk.astype(np.float64)[None, :]) @ Rk.astype(np.float64)).astype(f32)
    rec3 = rec.copy()
    rec3["p"] = pw3
    PO3 = (pw3 - Owk[None, :]).astype(f32)
    d3 = np.sqrt(np.sum(PO3.astype(np.float64) ** 2, axis=1)).astype(f32)
    n3 = (PO3 / np.maximum(d3, 1e-6)[:, None]).astype(f32)
    n3[side] = np.stack([n3[side, 1], -n3[side, 0], np.zeros(int(side.sum()), f32)], 1).astype(f32)
    rec3["n"] = n3
    rec3["dmin"] = (d3 * rng.choice([0.5, 0.9, 1.2], mL, p=[0.6, 0.35, 0.05])).astype(f32)
    rec3["dmax"] = (d3 * rng.choice([2.0, 1.1, 0.8], mL, p=[0.6, 0.35, 0.05])).astype(f32)
    rec3.tofile(ind / "L3_points.bin")
    ur_kf = np.where(rng.random(n1) < 0.7, kl1["x"] - rng.uniform(0.5, 40.0, n1), -1.0).astype(f32)
    ur_kf[rng.random(n1) < 0.03] = 0.0
    ur_kf.tofile(ind / "L3_uright.bin")
    slot0 = np.full(n1, -1, np.int32)
    r_ = rng.random(n1)
    slot0[r_ < 0.05] = rng.choice(mL, int((r_ < 0.05).sum()), replace=False)
    slot0[(r_ >= 0.05) & (r_ < 0.20)] = -2
    slot0[(r_ >= 0.20) & (r_ < 0.35)] = -3
    slot0[(r_ >= 0.35) & (r_ < 0.38)] = -4
    slot0.tofile(ind / "L3_kfmp.bin")
    null3 = (rng.random(mL) < 0.03).astype(np.uint8)
    null3.tofile(ind / "L3_null.bin")
    x3k = (_mm(Rk, pw3) + tk[None, :]).astype(f32)
    with np.errstate(divide="ignore"):
        invz3 = (f32(1) / x3k[:, 2]).astype(f32)
    xk = (x3k[:, 0] * invz3).astype(f32); yk = (x3k[:, 1] * invz3).astype(f32)
    u3 = ((fx * xk).astype(f32) + cx).astype(f32); v3 = ((fy * yk).astype(f32) + cy).astype(f32)
    ur3 = (u3 - (f32(MBF) * invz3).astype(f32)).astype(f32)
    dot3 = np.sum(PO3.astype(np.float64) * n3.astype(np.float64), axis=1)
    ok3 = ~(x3k[:, 2] < 0) & (u3 >= 0) & (u3 < 752) & (v3 >= 0) & (v3 < 480) & ~(d3 < rec3["dmin"]) & ~(d3 > rec3["dmax"]) & ~(dot3 < 0.5 * d3.astype(np.float64))
    in_kf0 = np.zeros(mL, bool)
    in_kf0[slot0[slot0 >= 0]] = True
    searched = ok3 & ~badL & ~in_kf0 & (null3 == 0)
    qi3 = np.nonzero(searched)[0]
    q3 = np.zeros(len(qi3), oracle.PROJ_QUERY_DTYPE)
    q3["u"], q3["v"], q3["ur"] = u3[qi3], v3[qi3], ur3[qi3]
    q3["radius"] = (th3 * sf[levL[qi3]]).astype(f32)
    q3["min_level"], q3["max_level"], q3["flags"] = levL[qi3] - 1, levL[qi3], 1
    inv_sig = (f32(1.0) / (sf * sf).astype(f32)).astype(f32)
    ref["L3_inv_sigma2"] = inv_sig
    outp3 = oracle.search_for_fusion(kl1, dl1, ur_kf, (0.0, 0.0, 752.0, 480.0), inv_sig, q3, qdL[qi3], 50)
    found_of = np.full(mL, -1, np.int64)
    found_of[qi3] = outp3
    # the serial part (:1067-1083) with the link support's Replace: holder codes as in the harness (candidate index, -100 - keypoint for a foreign point)
    holder = np.full(n1, -1, np.int64)
    cand_bad = badL.copy(); cand_repl = np.full(mL, -1, np.int64); cand_at = np.full(mL, -1, np.int64); cand_obs = np.full(mL, 3, np.int64)
    f_bad = np.zeros(n1, bool); f_repl = np.full(n1, -1, np.int64); f_obs = np.zeros(n1, np.int64); f_at = np.full(n1, -1, np.int64)
    for i in range(n1):
        v_ = int(slot0[i])
        if v_ >= 0:
            holder[i] = v_; cand_at[v_] = i; cand_obs[v_] += 1
        elif v_ <= -2:
            holder[i] = -100 - i; f_obs[i] = (5 if v_ == -3 else 1) + 1; f_bad[i] = v_ == -4; f_at[i] = i
    nf3 = 0
    for j in range(mL):
        if null3[j] or not searched[j]:
            continue
        if cand_bad[j] or cand_at[j] >= 0:                  # isBad() || IsInKeyFrame(pKF) at the point's own turn
            continue
        v_ = int(found_of[j])
        if v_ < 0:
            continue
        best = v_ & 0xFFFF
        h = int(holder[best])
        if h != -1:
            h_bad = bool(f_bad[-100 - h]) if h <= -100 else bool(cand_bad[h])
            if not h_bad:
                h_obs = int(f_obs[-100 - h]) if h <= -100 else int(cand_obs[h])
                if h_obs > cand_obs[j]:                      # pMP->Replace(pMPinKF): the candidate goes, the keyframe's point stays (it is in the keyframe: its slot stands)
                    cand_bad[j] = True; cand_repl[j] = h     # (the candidate has no observation in this keyframe to move)
                else:                                        # pMPinKF->Replace(pMP): the holder's observation of this keyframe moves to the candidate
                    if h <= -100:
                        f_bad[-100 - h] = True; f_repl[-100 - h] = j; f_at[-100 - h] = -1
                    else:
                        cand_bad[h] = True; cand_repl[h] = j; cand_at[h] = -1
                    holder[best] = j; cand_at[j] = best; cand_obs[j] += 1
        else:
            cand_at[j] = best; cand_obs[j] += 1; holder[best] = j
        nf3 += 1
    ref["L3"] = {"nf": nf3, "kf_after": holder.astype(np.int32), "bad": cand_bad.astype(np.int32), "replaced_by": cand_repl.astype(np.int32),
                 "observed_at": cand_at.astype(np.int32), "foreign_bad": f_bad.astype(np.int32), "foreign_replaced_by": f_repl.astype(np.int32),
                 "searched": int(searched.sum())}

    # L4: SearchBySim3(KF1 = frame 0, KF2 = frame 1): both keyframes at pose Tl, map points of each placed where the OTHER frame's matching
    # keypoint sits (frame 1 = frame 0 moved by 3 columns), a similarity close to the identity; host side restated (ORBmatcher.cc:1224-1300, 1338-1380)
    s12 = f32(1.02)
    R12 = _pose(0.001, -0.0015, 0.0008, [0, 0, 0])[:3, :3]
    t12 = np.array([0.004, -0.003, 0.006], f32)
    th4 = f32(7.5)
    recdt = np.dtype([("bad", "<i4"), ("level", "<i4"), ("p", "<f4", 3), ("n", "<f4", 3), ("dmin", "<f4"), ("dmax", "<f4")])

    def kf_points(kfrom, shift, nn):
        z = rng.uniform(2.0, 20.0, nn)
        uu = kfrom["x"] + shift + rng.normal(0, 1.2, nn)
        vv_ = kfrom["y"] + rng.normal(0, 1.2, nn)
        pc = np.stack([(uu - CX) / FX * z, (vv_ - CY) / FY * z, z], 1)
        pw_ = ((pc - tk.astype(np.float64)[None, :]) @ Rk.astype(np.float64)).astype(f32)
        r4 = np.zeros(nn, recdt)
        has = rng.random(nn) < 0.8
        r4["level"] = np.where(has, np.clip(kfrom["octave"] + rng.integers(0, 2, nn), 0, 7), -1)
        r4["bad"] = rng.random(nn) < 0.04
        r4["p"] = pw_
        dd = np.sqrt(np.sum(pc ** 2, axis=1)).astype(f32)
        r4["dmin"] = (dd * rng.choice([0.5, 0.9, 1.3], nn, p=[0.6, 0.35, 0.05])).astype(f32)
        r4["dmax"] = (dd * rng.choice([2.0, 1.1, 0.7], nn, p=[0.6, 0.35, 0.05])).astype(f32)
        return r4

    def flipped(dsrc):
        dd = dsrc.copy()
        for j in range(len(dd)):
            for b_ in rng.integers(0, 256, rng.integers(0, 60)):
                dd[j, b_ >> 3] ^= np.uint8(1 << (b_ & 7))
        return dd

    r41, r42 = kf_points(kl0, -3.0, n0), kf_points(kl1, 3.0, n1)
    d41, d42 = flipped(dl0), flipped(dl1)
    r41.tofile(ind / "L4_points1.bin"); r42.tofile(ind / "L4_points2.bin")
    d41.tofile(ind / "L4_desc1.bin"); d42.tofile(ind / "L4_desc2.bin")
    m12_0 = np.where(rng.random(n0) < 0.05, rng.integers(0, n1, n0), -1).astype(np.int32)
    m12_0[(m12_0 >= 0) & (r42["level"][np.maximum(m12_0, 0)] < 0)] = -1          # (a pair can only name a keypoint of KF2 that has a map point)
    m12_0.tofile(ind / "L4_matches12.bin")
    np.concatenate([Tl.ravel(), Tl.ravel(), np.array([s12], f32), R12.ravel(), t12, np.array([th4], f32)]).astype(f32).tofile(ind / "L4_calib.bin")
    sR12 = (R12 * s12).astype(f32)
    sR21 = (R12.T * f32(1.0 / float(s12))).astype(f32)                           # (1.0 / s12) * R12.t(): the factor rounded to float once
    t21 = _mm((-sR21).astype(f32), t12[None, :])[0]
    already1 = m12_0 >= 0
    already2 = np.zeros(n1, bool)
    already2[m12_0[m12_0 >= 0]] = True                                            # GetIndexInKeyFrame(pKF2) of the matched point = its keypoint there

    def direction(rfrom, dfrom, already, first_R, first_t, second_R, second_t, kto, dto, nto):
        has = rfrom["level"] >= 0
        pc_a = (_mm(first_R, rfrom["p"]) + first_t[None, :]).astype(f32)
        pc_b = (_mm(second_R, pc_a) + second_t[None, :]).astype(f32)
        with np.errstate(divide="ignore"):
            invz = (1.0 / pc_b[:, 2].astype(np.float64)).astype(f32)
        x_ = (pc_b[:, 0] * invz).astype(f32); y_ = (pc_b[:, 1] * invz).astype(f32)
        u_ = ((fx * x_).astype(f32) + cx).astype(f32); v_ = ((fy * y_).astype(f32) + cy).astype(f32)
        dist = np.sqrt(np.sum(pc_b.astype(np.float64) ** 2, axis=1)).astype(f32)
        keep_ = has & ~already & (rfrom["bad"] == 0) & ~(pc_b[:, 2] < 0) & (u_ >= 0) & (u_ < 752) & (v_ >= 0) & (v_ < 480) & ~(dist < rfrom["dmin"]) & ~(dist > rfrom["dmax"])
        qi_ = np.nonzero(keep_)[0]
        qq = np.zeros(len(qi_), oracle.PROJ_QUERY_DTYPE)
        lv = rfrom["level"][qi_]
        qq["u"], qq["v"] = u_[qi_], v_[qi_]
        qq["radius"] = (th4 * sf[lv]).astype(f32)
        qq["min_level"], qq["max_level"], qq["flags"] = lv - 1, lv, 1
        op = oracle.search_by_projection_queries_points(kto, dto, None, None, (0.0, 0.0, 752.0, 480.0), qq, dfrom[qi_], False, 0.0, 100, False, None)[3]
        match = np.full(len(rfrom), -1, np.int64)
        match[qi_] = np.where(op >= 0, op & 0xFFFF, -1)
        return match, len(qi_)

    vn1, nq1 = direction(r41, d41, already1, Rk, tk, sR21, t21, kl1, dl1, n1)
    vn2, nq2 = direction(r42, d42, already2, Rk, tk, sR12, t12, kl0, dl0, n0)
    out12 = m12_0.copy()
    nfound = 0
    for i1 in range(n0):
        i2 = int(vn1[i1])
        if i2 >= 0 and int(vn2[i2]) == i1:
            out12[i1] = i2
            nfound += 1
    ref["L4"] = {"nfound": nfound, "matches12": out12.astype(np.int32), "queries": (nq1, nq2)}

    # I: SearchByBoW(KeyFrame = frame 0, F = frame 1): feature vectors = a node id per keypoint (similar descriptors share a node)
    knode = (dl0[:, 0].astype(np.int32) >> 2)                       # 64 "vocabulary nodes" from the descriptors' first bits
    fnode = (dl1[:, 0].astype(np.int32) >> 2)
    knode[rng.random(n0) < 0.05] = -1                              # stopped words: not in the vector
    fnode[rng.random(n1) < 0.05] = -1
    kvalid = rng.choice([0, 1, 2], n0, p=[0.25, 0.7, 0.05]).astype(np.uint8)    # no map point / a good one / a bad one
    knode.astype(np.int32).tofile(ind / "I_kf_nodes.bin")
    fnode.astype(np.int32).tofile(ind / "I_f_nodes.bin")
    kvalid.tofile(ind / "I_kf_valid.bin")
    ref["I"] = [oracle.search_by_bow(dl0, kl0["angle"], (kvalid == 1).astype(np.uint8), oracle.make_feature_vector(knode), dl1, kl1["angle"],
                                     oracle.make_feature_vector(fnode), 0.7, bool(o)) for o in (0, 1)]
    # I2: SearchByBoW(KeyFrame 1 = frame 0, KeyFrame 2 = frame 1): the second keyframe carries a map-point list as well
    kvalid2 = rng.choice([0, 1, 2], n1, p=[0.2, 0.75, 0.05]).astype(np.uint8)
    kvalid2.tofile(ind / "I2_kf2_valid.bin")
    ref["I2"] = [oracle.search_by_bow_keyframes(dl0, kl0["angle"], (kvalid == 1).astype(np.uint8), oracle.make_feature_vector(knode), dl1, kl1["angle"],
                                                (kvalid2 == 1).astype(np.uint8), oracle.make_feature_vector(fnode), 0.75, bool(o)) for o in (0, 1)]
    # I3: SearchForTriangulation(KF1 = frame 0, KF2 = frame 0 seen from a second pose): the epipole from the member's own cv::Mat lines
    # (float, the stand-in's plain loops), flags 0 / 1 / 2 = no map point / a good one / a bad one (a bad one still occupies its keypoint)
    import gf_cases
    c3 = gf_cases.triangulation_case(oracle, kl0, dl0, np.random.default_rng(31), fx=FX, fy=FY, cx=CX, cy=CY)
    h31 = np.where(c3["has1"] > 0, rng.choice([1, 2], n0, p=[0.9, 0.1]), 0).astype(np.uint8)
    h32 = np.where(c3["has2"] > 0, rng.choice([1, 2], n0, p=[0.9, 0.1]), 0).astype(np.uint8)
    ow1 = np.array([0.4, -0.2, 0.1], f32)                              # the first keyframe's centre in the world: Tcw1 = [I | -ow1]
    T1 = np.eye(4, dtype=f32); T1[:3, 3] = -ow1
    T2 = np.eye(4, dtype=f32); T2[:3, :3] = c3["R21"].astype(f32); T2[:3, 3] = (c3["t21"] - c3["R21"] @ ow1.astype(np.float64)).astype(f32)
    c2 = np.zeros(3, f32)
    for i_ in range(3):
        s_ = f32(0)
        for k_ in range(3):
            s_ = f32(s_ + f32(T2[i_, k_] * ow1[k_]))
        c2[i_] = f32(s_ + T2[i_, 3])
    invz3 = f32(1) / c2[2]
    ex3 = f32(f32(f32(f32(FX) * c2[0]) * invz3) + f32(CX)); ey3 = f32(f32(f32(f32(FY) * c2[1]) * invz3) + f32(CY))
    c3["kp2"].tofile(ind / "I3_kp2.bin"); c3["desc2"].tofile(ind / "I3_desc2.bin")
    c3["node1"].astype(np.int32).tofile(ind / "I3_nodes1.bin"); c3["node2"].astype(np.int32).tofile(ind / "I3_nodes2.bin")
    h31.tofile(ind / "I3_has1.bin"); h32.tofile(ind / "I3_has2.bin")
    c3["ur1"].tofile(ind / "I3_uright1.bin"); c3["ur2"].tofile(ind / "I3_uright2.bin")
    np.concatenate([T1.ravel(), T2.ravel(), ow1, c3["f12"].ravel(), np.array([FX, FY, CX, CY], f32)]).astype(f32).tofile(ind / "I3_geom.bin")
    sf3 = np.cumprod(np.concatenate([[f32(1)], np.full(7, f32(1.2))]).astype(f32)).astype(f32)
    ref["I3"] = [oracle.search_for_triangulation(kl0, dl0, h31, c3["ur1"], c3["fv1"], c3["kp2"], c3["desc2"], h32, c3["ur2"], c3["fv2"], sf3, (sf3 * sf3).astype(f32),
                                                 c3["f12"], ex3, ey3, bool(only), not only) for only in (0, 1)]
    # M: SearchForInitialization(F1 = frame 0, F2 = a displaced resampling of it), twice on the same vbPrevMatched
    kpm, dm, prevm = gf_cases.initialization_case(oracle, kl0, dl0, np.random.default_rng(41))
    kpm.tofile(ind / "M_kp2.bin"); dm.tofile(ind / "M_desc2.bin"); prevm.tofile(ind / "M_prev.bin")
    ref["M"] = []
    pm = prevm.copy()
    for _ in range(2):
        nm_, m12_ = oracle.search_for_initialization(kl0, dl0, pm, kpm, dm, (0.0, 0.0, 752.0, 480.0), 100, 0.9, True)
        ref["M"].append((nm_, m12_.copy(), pm.copy()))
    with oracle.feature_budget(BUDGET):
        ref["I_budget"] = [oracle.search_by_bow(dl0, kl0["angle"], (kvalid == 1).astype(np.uint8), oracle.make_feature_vector(knode), dl1, kl1["angle"],
                                                oracle.make_feature_vector(fnode), 0.7, bool(o)) for o in (0, 1)]

    # J: Frame::ComputeBoW() on frame 1 with vocabularies of several shapes / weightings / scorings (levelsup = 4, Frame.cc:666)
    ref["J"] = []
    for v, (k, depth, weighting, scoring, norm) in enumerate([(10, 3, 0, 0, 1), (6, 5, 1, 1, 2), (8, 4, 2, 5, 0), (9, 2, 3, 0, 1)]):
        voc = oracle.make_vocabulary(k, depth, seed=40 + v, p_stop=0.08)
        voc["weight"] = voc["weight64"].astype(np.float32)           # DBoW2 holds ONE weight per node, a double
        n = len(voc["first_child"])
        np.array([n, k, depth, weighting, scoring], np.int32).tofile(ind / f"J{v}_voc_hdr.bin")
        voc["first_child"].astype(np.int32).tofile(ind / f"J{v}_first.bin")
        voc["n_children"].astype(np.int32).tofile(ind / f"J{v}_nch.bin")
        voc["word_id"].astype(np.int32).tofile(ind / f"J{v}_word.bin")
        voc["weight64"].astype(np.float64).tofile(ind / f"J{v}_weight.bin")
        voc["descriptors"].astype(np.uint8).tofile(ind / f"J{v}_desc.bin")
        ref["J"].append(oracle.compute_bow(voc, dl1, 4, weighting, norm))
    # J4: a vocabulary of the size the reference loads (ORBvoc: k = 10, L = 6, 1 111 111 nodes), through the ORBVocabulary object:
    # VocabularyView::flatten() walks a million Node objects, the fingerprint samples 64 of them per call (main run only: GFO_ADAPTER_BIGVOC)
    vocb = oracle.make_vocabulary_full(10, 6, seed=21)
    np.array([len(vocb["first_child"]), 10, 6, 0, 0], np.int32).tofile(ind / "J4_voc_hdr.bin")
    vocb["first_child"].astype(np.int32).tofile(ind / "J4_first.bin")
    vocb["n_children"].astype(np.int32).tofile(ind / "J4_nch.bin")
    vocb["word_id"].astype(np.int32).tofile(ind / "J4_word.bin")
    vocb["weight64"].astype(np.float64).tofile(ind / "J4_weight.bin")
    vocb["descriptors"].astype(np.uint8).tofile(ind / "J4_desc.bin")
    ref["Jbig"] = oracle.compute_bow(vocb, dl1, 4, 0, 1)
    del vocb

    env = dict(os.environ)
    env.pop("GFO_COMBINE", None)
    env.pop("GFO_FULL_PYRAMID", None)
    p = subprocess.run([EXE, GOLDEN, str(ind), str(outd), str(NF)], capture_output=True, text=True, timeout=600, env=dict(env, GFO_ADAPTER_BIGVOC="1"))
    ref["rc"], ref["stderr"], ref["out"] = p.returncode, p.stderr, outd
    ref["report"] = dict(l.split() for l in open(outd / "report.txt").read().splitlines() if l.strip()) if (outd / "report.txt").exists() else {}
    ref["oracle"] = oracle
    # a second, short run with GFO_FULL_PYRAMID=1: operator() then leaves the levels' PIXELS in mvImagePyramid (what the SAD stereo
    # variant reads, Frame.cc:994,1016); no frame combiner in that mode (the levels must be in the extractor's own context)
    outd2 = tmp_path_factory.mktemp("adapter_out_full")
    p2 = subprocess.run([EXE, GOLDEN, str(ind), str(outd2), "2"], capture_output=True, text=True, timeout=600, env=dict(env, GFO_FULL_PYRAMID="1"))
    ref["full"] = {"rc": p2.returncode, "stderr": p2.stderr, "out": outd2}
    # a third run with GFO_DEVICES=0,0,0: three SLOTS (this box has one GPU; a node lists its eight): extractors are placed per slot,
    # the rigs are co-located on their first frame -- contexts are destroyed and re-created mid-stream -- and every result is the same
    outd4 = tmp_path_factory.mktemp("adapter_out_devices")
    p4 = subprocess.run([EXE, GOLDEN, str(ind), str(outd4), "6"], capture_output=True, text=True, timeout=600, env=dict(env, GFO_DEVICES="0,0,0"))
    ref["devices"] = {"rc": p4.returncode, "stderr": p4.stderr, "out": outd4,
                      "report": dict(l.split() for l in open(outd4 / "report.txt").read().splitlines() if l.strip()) if (outd4 / "report.txt").exists() else {}}
    outd5 = tmp_path_factory.mktemp("adapter_out_budget")
    if os.path.exists(EXE + "_budget"):
        p5 = subprocess.run([EXE + "_budget", GOLDEN, str(ind), str(outd5), "3"], capture_output=True, text=True, timeout=600, env=env)
        ref["budget"] = {"rc": p5.returncode, "stderr": p5.stderr, "out": outd5}
    outd3 = tmp_path_factory.mktemp("adapter_out_delayed")
    if os.path.exists(EXE + "_delayed"):
        p3 = subprocess.run([EXE + "_delayed", GOLDEN, str(ind), str(outd3), "3"], capture_output=True, text=True, timeout=600, env=env)
        ref["delayed"] = {"rc": p3.returncode, "stderr": p3.stderr, "out": outd3}
    keep = os.path.join(ROOT, "gpurun_out")          # on the GPU box: the program's own report comes back with the call
    if os.path.isdir(keep) and (outd / "report.txt").exists():
        with open(os.path.join(keep, "adapter_run_report.txt"), "w") as fh:
            fh.write(open(outd / "report.txt").read() + "--- stderr ---\n" + p.stderr[-4000:])
    return ref


def _rd(run, name, dtype):
    return np.fromfile(run["out"] / name, dtype)


def test_adapter_program_ran_clean(run):
    """the checks that need no oracle: sizes of mvImagePyramid, context counts, untouched outputs on an empty image, released
    descriptors on a cornerless one (tests/host/adapter_run.cc CHECK lines)"""
    assert run["rc"] == 0, run["stderr"][-4000:]
    assert run["report"].get("check_failures") == "0"
    assert run["report"]["contexts_created_in_steady_state"] == "0"
    assert run["report"]["contexts_created_by_reconstruction"] == "1"
    assert "[gfo]" not in run["stderr"], run["stderr"][-2000:]         # no library error was reported and swallowed


def test_operator_call_on_two_threads_equals_the_oracle(run):
    """ORBextractor::operator() (ORBextractor.h:89-91) called like Frame.cc:84-87 for 20 frames: keypoints (28-byte cv::KeyPoint)
    and descriptor rows, both cameras, bit for bit; the last frame is a non-continuous cv::Mat view (step != cols)"""
    kd = run["oracle"].KEYPOINT_DTYPE
    for f, (kl, dl, kr, dr, st) in enumerate(run["frames"]):
        for side, k, d in (("l", kl, dl), ("r", kr, dr)):
            gk = _rd(run, f"A_f{f:02d}_k{side}.bin", kd)
            gd = _rd(run, f"A_f{f:02d}_d{side}.bin", np.uint8).reshape(-1, 32)
            assert gk.tobytes() == k.tobytes(), f"frame {f} {side}: keypoints"
            assert gd.tobytes() == d.tobytes(), f"frame {f} {side}: descriptors"
    sz = _rd(run, "A_level_sizes.bin", np.int32).reshape(8, 2)
    oe = run["oracle"].OracleExtractor(2000, 1.2, 8, 20, 7)
    oe.compute_pyramid(np.zeros((480, 752), np.uint8))
    assert [tuple(s) for s in sz] == [tuple(oe.level_size(l)) for l in range(8)]
    tabs = _rd(run, "A_tables.bin", np.float32).reshape(4, 8)
    for got, want in zip(tabs, (oe.scale_factors, oe.inv_scale_factors, oe.level_sigma2, oe.inv_level_sigma2)):
        assert got.tobytes() == np.asarray(want, np.float32).tobytes()


def test_stereo_association_member_equals_the_oracle_and_uses_the_rig(run):
    """Frame::ComputeStereoMatches_Undistorted (Frame.cc:1167-1316) as swapped in by adapter/matchers_gfo.cc: mvuRight, mvDepth,
    mvDistIdx and the return value of every frame; from the second frame on the two operator() calls went to the device as one
    rig submission and the member was answered from it"""
    for f, (kl, dl, kr, dr, st) in enumerate(run["frames"]):
        nm, ur, dp, bd, bi = st
        assert _rd(run, f"A_f{f:02d}_uright.bin", np.float32).tobytes() == ur.tobytes(), f
        assert _rd(run, f"A_f{f:02d}_depth.bin", np.float32).tobytes() == dp.tobytes(), f
        want = sorted((int(bd[i]), i) for i in range(len(bd)) if bd[i] >= 0)
        got = _rd(run, f"A_f{f:02d}_distidx.bin", np.int32).reshape(-1, 2)
        assert [tuple(x) for x in got.tolist()] == want, f
        assert int(_rd(run, f"A_f{f:02d}_nstereo.bin", np.int32)[0]) == nm
    rep = run["report"]
    assert int(rep["combiner_rig_frames"]) >= NF - 3, rep            # the first frame declares the rig; a partner may miss its window once
    assert int(rep["combiner_rig_answers"]) >= NF - 3, rep
    assert int(rep["combiner_broken"]) == 0


def test_compute_pyramid_levels_with_their_frames(run):
    """ORBextractor::ComputePyramid (ORBextractor.h:132, .cc:1176-1201): every level a view into its (w+38) x (h+38) buffer whose
    19-px frame is the reflect-101 border"""
    oe = run["oracle"].OracleExtractor(2000, 1.2, 8, 20, 7)
    img = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_752x480.u8"), np.uint8).reshape(480, 752)
    oe.compute_pyramid(img)
    for l in range(8):
        w, h = oe.level_size(l)
        got = _rd(run, f"B_level{l}_framed_{w + 38}x{h + 38}.bin", np.uint8).reshape(h + 38, w + 38)
        assert got.tobytes() == oe.level(l, padded=True).tobytes(), l
    kd = run["oracle"].KEYPOINT_DTYPE
    kl, dl = run["frames"][0][0], run["frames"][0][1]
    assert _rd(run, "B_after_kl.bin", kd).tobytes() == kl.tobytes()
    assert _rd(run, "B_after_dl.bin", np.uint8).tobytes() == dl.tobytes()


def test_extractor_recreated_at_one_address(run, euroc_l, euroc_r):
    """Tracking::updateORBExtractor (Tracking.cc:298-320): other parameters at the same address -> the new object's results"""
    O = run["oracle"]
    kd = O.KEYPOINT_DTYPE
    k, d = O.OracleExtractor(1000, 1.2, 8, 20, 7)(euroc_l)
    assert _rd(run, "C_first_kl.bin", kd).tobytes() == k.tobytes() and _rd(run, "C_first_dl.bin", np.uint8).tobytes() == d.tobytes()
    assert _rd(run, "C_third_kl.bin", kd).size > 0
    e2 = O.OracleExtractor(1500, 1.2, 8, 12, 5)
    k, d = e2(euroc_r)
    assert _rd(run, "C_second_kl.bin", kd).tobytes() == k.tobytes() and _rd(run, "C_second_dl.bin", np.uint8).tobytes() == d.tobytes()
    k, d = e2(euroc_l)
    assert _rd(run, "C_third_kl.bin", kd).tobytes() == k.tobytes() and _rd(run, "C_third_dl.bin", np.uint8).tobytes() == d.tobytes()


def _stereo_state_equal(run, tag, want):
    nm, ur, dp, di = want
    assert _rd(run, f"{tag}_f02_uright.bin", np.float32).tobytes() == ur.tobytes(), tag
    assert _rd(run, f"{tag}_f02_depth.bin", np.float32).tobytes() == dp.tobytes(), tag
    assert _rd(run, f"{tag}_f02_distidx.bin", np.int32).reshape(-1, 2).tolist() == di.tolist(), tag
    assert int(_rd(run, f"{tag}_f02_nstereo.bin", np.int32)[0]) == nm, tag


def test_stereo_member_second_call_on_a_frame_keeps_its_state(run):
    """Tracking.cc:941-954 calls ComputeStereoMatches_Undistorted a second time on mCurrentFrame once map points narrowed the windows
    (Frame.cc:1220-1231).  Nothing is reset (mvRowIndices.size() == nRows, :1173-1176): a keypoint whose narrowed window rejects
    its match keeps the first call's uRight / depth, accepted matches are appended to mvDistIdx (:1282) and the cut runs over the
    accumulated list (:1290-1313).  Compared with the oracle's literal member state (orc_stereo_frame); a third, online call on the
    same frame; and the same frame after the caller's own PrepareStereoCandidates (Tracking.cc:613)."""
    assert run["D_windows_used"] > 300
    first, second, fresh = run["D_first"], run["D"], run["D2"]
    assert first[1].tobytes() == run["frames"][2][4][1].tobytes()     # the state part A left is the fresh call's
    _stereo_state_equal(run, "D", second)
    _stereo_state_equal(run, "D3", run["D3"])
    _stereo_state_equal(run, "D4", run["D4"])
    # what makes it a second call: values survive that a fresh frame with the same windows does not have, the list has grown
    survived = (fresh[1] < 0) & (second[1] >= 0)
    assert survived.sum() > 20, survived.sum()
    assert len(second[3]) > len(first[3])
    assert len(run["D3"][3]) > len(second[3])


def test_extractors_spread_over_device_slots_and_rigs_are_colocated(run):
    """GFO_DEVICES (adapter/gfo_context_table.h): the adapter places each ORBextractor on the least-loaded listed device and the
    stereo member moves a rig onto one device before it pairs the two extractors.  Run with three slots (one GPU listed three
    times): placement and moves are real -- contexts die and are re-created between frames -- and nothing in the results may change:
    the program's own checks (every camera of part K equals part A) pass and part A still equals the oracle."""
    d = run["devices"]
    assert d["rc"] == 0, d["stderr"][-3000:]
    rep = d["report"]
    assert rep.get("check_failures") == "0"
    assert "[gfo]" not in d["stderr"], d["stderr"][-2000:]
    assert int(rep["contexts_moved"]) >= 1                                   # at least one rig was found on two slots and moved
    slots = [int(rep[f"K_camera{k}_slot"]) for k in range(3)]
    assert len(set(slots)) >= 2, slots                                       # three cameras do not pile onto one slot
    assert all(int(rep[f"K_camera{k}_device"]) == 0 for k in range(3))
    assert int(run["report"]["contexts_moved"]) == 0                         # the default (one device): nothing ever moves
    kd = run["oracle"].KEYPOINT_DTYPE
    for f in range(3):
        kl, dl, kr, dr, st = run["frames"][f]
        assert np.fromfile(d["out"] / f"A_f{f:02d}_kl.bin", kd).tobytes() == kl.tobytes(), f
        assert np.fromfile(d["out"] / f"A_f{f:02d}_dr.bin", np.uint8).tobytes() == dr.tobytes(), f
        assert np.fromfile(d["out"] / f"A_f{f:02d}_uright.bin", np.float32).tobytes() == st[1].tobytes(), f


def test_stereo_member_of_a_delayed_stereo_matching_build(run):
    """adapter/matchers_gfo.cc + the harness compiled with -DDELAYED_STEREO_MATCHING (include/Frame.h:40): the online call visits only
    unvisited keypoints that carry a map point, the offline call the other unvisited ones (Frame.cc:1186-1199), state and
    mvDistIdx run through all four calls of the frame; against the oracle's literal member (orc_stereo_frame, delayed = 1)"""
    if "delayed" not in run:
        pytest.skip("tests/_build/adapter_run_delayed is missing (built by __graft_entry__.build() where the reference headers are)")
    d = run["delayed"]
    assert d["rc"] == 0, d["stderr"][-3000:]
    assert "[gfo]" not in d["stderr"], d["stderr"][-2000:]
    kd = run["oracle"].KEYPOINT_DTYPE
    assert np.fromfile(d["out"] / "DD_kl.bin", kd).tobytes() == run["frames"][2][0].tobytes()
    for call, (nm, ur, dp, di, matched) in enumerate(run["DD"]):
        t = f"DD{call + 1}"
        assert np.fromfile(d["out"] / f"{t}_f02_uright.bin", np.float32).tobytes() == ur.tobytes(), t
        assert np.fromfile(d["out"] / f"{t}_f02_depth.bin", np.float32).tobytes() == dp.tobytes(), t
        assert np.fromfile(d["out"] / f"{t}_f02_distidx.bin", np.int32).reshape(-1, 2).tolist() == di.tolist(), t
        assert int(np.fromfile(d["out"] / f"{t}_f02_nstereo.bin", np.int32)[0]) == nm, t
        assert np.fromfile(d["out"] / f"{t}_f02_matched.bin", np.uint8).tobytes() == matched.tobytes(), t
    n1, n2, n3, n4 = (r[0] for r in run["DD"])
    assert n1 > 200 and n2 > 100 and n3 > 300                     # every stage visited its share
    assert n4 <= 0 and run["DD"][3][4].all()                      # the last call visits nothing; its cut only counts down (:1311)
    assert (run["DD"][0][4].sum() < run["DD"][1][4].sum() < run["DD"][2][4].sum())


def test_stereo_member_with_map_point_windows_on_a_fresh_frame(run):
    """Frame.cc:1220-1231: the adapter flattens MapPoint::isBad / GetWorldPos / Frame::WorldToCameraPoint into minD / maxD; the
    first call of a frame that already carries map points"""
    nm, ur, dp, bd, bi = run["D2"]
    assert _rd(run, "D2_f02_uright.bin", np.float32).tobytes() == ur.tobytes()
    assert _rd(run, "D2_f02_depth.bin", np.float32).tobytes() == dp.tobytes()
    want = sorted((int(bd[i]), i) for i in range(len(bd)) if bd[i] >= 0)
    assert [tuple(x) for x in _rd(run, "D2_f02_distidx.bin", np.int32).reshape(-1, 2).tolist()] == want
    assert int(_rd(run, "D2_f02_nstereo.bin", np.int32)[0]) == nm
    assert ur.tobytes() != run["frames"][2][4][1].tobytes()          # the windows changed the association


def test_stereo_member_online_call_keeps_every_accepted_match(run):
    """ComputeStereoMatches_Undistorted(true): the reference's cut is under `if (!isOnline)` (Frame.cc:1290); the adapter restores
    what the library's cut cleared with the reference's own expressions (:1271-1281).  Same arrays, offline then online."""
    for tag, (nm, ur, dp, bd, bi) in zip(("Goff", "Gon"), run["G"]):
        assert _rd(run, f"{tag}_f00_uright.bin", np.float32).tobytes() == ur.tobytes(), tag
        assert _rd(run, f"{tag}_f00_depth.bin", np.float32).tobytes() == dp.tobytes(), tag
        assert int(_rd(run, f"{tag}_f00_nstereo.bin", np.int32)[0]) == nm, tag
        want = sorted((int(bd[i]), i) for i in range(len(bd)) if bd[i] >= 0)
        assert sorted(tuple(x) for x in _rd(run, f"{tag}_f00_distidx.bin", np.int32).reshape(-1, 2).tolist()) == want, tag
    off, on = run["G"]
    assert on[0] > off[0] + 20 and (on[1] >= 0).sum() > (off[1] >= 0).sum() + 20     # the offline call cut matches the online call keeps


def test_search_by_projection_member(run):
    """ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (ORBmatcher.h:64, .cc:155-241) through MapPoint objects:
    which map point ends in which mvpMapPoints slot, mvpMatchScore, the count"""
    nm, out_mp, out_sc = run["E"]
    assert nm > 300
    got = _rd(run, "E_out_mp.bin", np.int32)
    np.testing.assert_array_equal(got, out_mp)
    sc = _rd(run, "E_out_score.bin", np.int32)
    np.testing.assert_array_equal(sc[out_mp >= 0], out_sc[out_mp >= 0])
    assert (sc[out_mp < 0] == 0).all()
    assert int(_rd(run, "E_nmatches.bin", np.int32)[0]) == nm


def test_search_by_projection_budget_member(run):
    """ORBmatcher::SearchByProjection_Budget(F, MapPoints, th, time_constr) (ORBmatcher.h:67, .cc:45-153), the matcher of the reference's
    default (good-feature) build behind Tracking::SearchAdditionalMatchesInFrame: slots, scores, the count, IncreaseFound() per match --
    with a budget no call can spend (the oracle's clock never trips) and with none left on entry (its first reading trips)."""
    for v, (nm, out_mp, out_sc, out_pt, found) in enumerate(run["E2"]):
        np.testing.assert_array_equal(_rd(run, f"E2_f{v:02d}_out_mp.bin", np.int32), out_mp)
        sc = _rd(run, f"E2_f{v:02d}_out_score.bin", np.int32)
        np.testing.assert_array_equal(sc[out_mp >= 0], out_sc[out_mp >= 0])
        assert (sc[out_mp < 0] == 0).all()
        np.testing.assert_array_equal(_rd(run, f"E2_f{v:02d}_found.bin", np.int32), found)
        assert int(_rd(run, f"E2_f{v:02d}_nmatches.bin", np.int32)[0]) == nm
    assert run["E2"][0][0] > 200 and run["E2"][1][0] <= 1 and (run["E2"][1][3] == -4).sum() > 2900


def test_one_point_matcher_of_the_good_feature_selection_loop(run):
    """ORBmatcher::SearchByProjection_OnePoint (include/ORBmatcher.h:71-150), an inline member no link can swap, as
    adapter/good_feature_matching_gfo.h offers it to Observability::runActiveMapMatching: the candidate table from ONE device call, then
    2000 picks in an order made up on the Python side -- every pick's return value, the slots and scores the picks leave, and
    mvMatchCandidates.size() of every point (GetCandidates, :152-172) against the oracle's literal statement."""
    order, res, ncand, out_mp, out_sc = run["E3"]
    np.testing.assert_array_equal(_rd(run, "E3_ncand.bin", np.int32), ncand)
    np.testing.assert_array_equal(_rd(run, "E3_results.bin", np.int32), res)
    np.testing.assert_array_equal(_rd(run, "E3_out_mp.bin", np.int32), out_mp)
    sc = _rd(run, "E3_out_score.bin", np.int32)
    np.testing.assert_array_equal(sc[out_mp >= 0], out_sc[out_mp >= 0])
    assert (res >= 0).sum() > 150 and ncand.max() > 3


def test_search_by_projection_last_frame_member(run):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono, numVisible) (ORBmatcher.cc:1440-1593): the adapter's
    host-side projection (:1451-1502) + the device search, neutral / forward / backward motion and the orientation check off"""
    assert [(v["fwd"], v["bwd"]) for v in run["F"]] == [(False, False), (True, False), (False, True), (False, False)]
    for v, want in enumerate(run["F"]):
        got = _rd(run, f"F{v}_out_last_idx.bin", np.int32)
        nm, vis = _rd(run, f"F{v}_nmatches_visible.bin", np.int32)
        assert vis == want["visible"], v
        np.testing.assert_array_equal(got, want["idx"], err_msg=f"variant {v}")
        assert nm == want["nm"], v
        assert want["nm"] > 200, (v, want["nm"])


def test_search_by_projection_keyframe_member(run):
    """ORBmatcher::SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1595-1721) through a KeyFrame
    and MapPoint objects: already-found / bad points skipped, the distance range, the predicted level window, slots that were set
    before the call untouched"""
    want = run["H"]
    assert want["queries"] > 800 and want["nm"] > 150 and want["cleared"] > 10
    got = _rd(run, "H_out_kf_idx.bin", np.int32)
    np.testing.assert_array_equal(got, want["idx"])
    assert int(_rd(run, "H_nmatches.bin", np.int32)[0]) == want["nm"]


def test_search_by_bow_member(run):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (ORBmatcher.cc:270-404): the adapter flattens two real
    DBoW2::FeatureVector objects (the reference's own class) and the keyframe's map-point list; with and without the rotation check"""
    for tag, (nm, out) in zip(("", "_ori"), run["I"]):
        got = _rd(run, f"I_out_kf_idx{tag}.bin", np.int32)
        np.testing.assert_array_equal(got, out, err_msg=tag)
        assert int(_rd(run, f"I_nmatches{tag}.bin", np.int32)[0]) == nm
        assert nm > 100, (tag, nm)


def test_loop_closing_projection_members_under_a_similarity(run):
    """ORBmatcher::SearchByProjection(KeyFrame*, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:406-518) and Fuse(KeyFrame*, Scw, vpPoints, th,
    vpReplacePoint) (:1089-1212), called from a thread of their own with the calling thread's device context: which candidate point ends in
    which vpMatched slot; for Fuse every point's replacement, the keyframe's map points afterwards and where each point was added as an
    observation -- the reference's serial side effects applied to ONE device call's independent searches."""
    l1 = run["L1"]
    np.testing.assert_array_equal(_rd(run, "L1_matched.bin", np.int32), l1["matched"])
    assert int(_rd(run, "L1_nmatches.bin", np.int32)[0]) == l1["nm"] and l1["nm"] > 200 and l1["queries"] > 800
    l2 = run["L2"]
    assert int(_rd(run, "L2_nfused.bin", np.int32)[0]) == l2["nf"] and l2["nf"] > 200
    np.testing.assert_array_equal(_rd(run, "L2_replace.bin", np.int32), l2["replace"])
    np.testing.assert_array_equal(_rd(run, "L2_kf_after.bin", np.int32), l2["kf_after"])
    np.testing.assert_array_equal(_rd(run, "L2_observed_at.bin", np.int32), l2["observed_at"])
    assert (l2["replace"] >= 0).any() and (l2["replace"] <= -100).any() and (l2["observed_at"] >= 0).sum() > 100


def test_local_mapping_fuse_member(run):
    """ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (ORBmatcher.cc:937-1087), LocalMapping::SearchInNeighbors' matcher: every
    point's gated search from ONE device call, then the reference's serial side effects -- which candidate replaced which point or was
    replaced, who is bad afterwards, the keyframe's slots, the observations added -- against the oracle's search + a Python walk of
    :1067-1083.  Replace() runs in both directions (foreign points with 2 and 6 observations against candidates with 3)."""
    l3 = run["L3"]
    assert int(_rd(run, "L3_nfused.bin", np.int32)[0]) == l3["nf"] and l3["nf"] > 150 and l3["searched"] > 500
    for name in ("kf_after", "bad", "replaced_by", "observed_at", "foreign_bad", "foreign_replaced_by"):
        np.testing.assert_array_equal(_rd(run, f"L3_{name}.bin", np.int32), l3[name], err_msg=name)
    assert (l3["replaced_by"] <= -100).any() and (l3["foreign_replaced_by"] >= 0).any() and (l3["observed_at"] >= 0).sum() > 100


def test_search_by_sim3_member(run):
    """ORBmatcher::SearchBySim3(KF1, KF2, vpMatches12, s12, R12, t12, th) (ORBmatcher.cc:1214-1438), loop closing's guided matcher: both
    directions' independent searches from two device calls, the pairs both directions agree on, the pairs matched on entry left alone."""
    l4 = run["L4"]
    assert int(_rd(run, "L4_nfound.bin", np.int32)[0]) == l4["nfound"] and l4["nfound"] > 150 and min(l4["queries"]) > 600
    np.testing.assert_array_equal(_rd(run, "L4_matches12_out.bin", np.int32), l4["matches12"])


def test_search_by_bow_between_keyframes_member(run):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (ORBmatcher.cc:635-768), loop closing's matcher, called from a
    thread of its own: two KeyFrame objects with their map-point lists and real DBoW2::FeatureVector objects; the device context is
    the calling thread's (no Frame, no extractor in the call)"""
    for tag, (nm, out) in zip(("", "_ori"), run["I2"]):
        got = _rd(run, f"I2_out_idx2{tag}.bin", np.int32)
        np.testing.assert_array_equal(got, out, err_msg=tag)
        assert int(_rd(run, f"I2_nmatches{tag}.bin", np.int32)[0]) == nm
        assert nm > 80, (tag, nm)


def test_search_for_triangulation_member(run):
    """ORBmatcher::SearchForTriangulation(KF1, KF2, F12, vMatchedPairs, bOnlyStereo) (ORBmatcher.cc:770-935), local mapping's matcher, from a
    thread of its own: two KeyFrame objects (map-point lists, mvuRight, real DBoW2::FeatureVector objects, poses), the epipole from the
    member's own cv::Mat lines, vMatchedPairs cleared and refilled in ascending first index"""
    for name, (nm, out) in zip(("I3_pairs.bin", "I3_pairs_stereo.bin"), run["I3"]):
        flat = _rd(run, name, np.int32)
        assert int(flat[-2]) == nm and int(flat[-1]) == nm, (name, flat[-2:], nm)
        pairs = flat[:-2].reshape(-1, 2)
        want = np.flatnonzero(out >= 0)
        np.testing.assert_array_equal(pairs[:, 0], want, err_msg=name)
        np.testing.assert_array_equal(pairs[:, 1], out[want], err_msg=name)
        assert nm > 100, (name, nm)


def test_keyframe_matchers_from_two_threads_at_once(run):
    """LocalMapping and LoopClosing run beside each other: SearchForTriangulation on one thread and SearchByBoW(KF, KF) on another, forty
    calls each at the same time on the same two KeyFrame objects, each thread on its own device context -- every answer is the one the
    single call gave"""
    bad, done, nt0, nb0 = _rd(run, "T_concurrent.bin", np.int32)
    assert bad == 0 and done == 80, (bad, done)
    assert nt0 == run["I3"][0][0] and nb0 > 50, (nt0, nb0)


def test_search_for_initialization_member(run):
    """ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (ORBmatcher.cc:520-633), the monocular bootstrap's
    matcher: vnMatches12 replaced, vbPrevMatched updated in place, and the second call on the updated vector"""
    for (a, b), (nm, m12, prev) in zip((("M_matches12.bin", "M_prev_out.bin"), ("M_matches12_again.bin", "M_prev_again.bin")), run["M"]):
        got = _rd(run, a, np.int32)
        assert int(got[-1]) == nm and nm > 100, (a, got[-1], nm)
        np.testing.assert_array_equal(got[:-1], m12, err_msg=a)
        assert _rd(run, b, np.float32).tobytes() == prev.tobytes(), b
    assert run["M"][1][2].tobytes() != run["M"][0][2].tobytes() or run["M"][1][0] == run["M"][0][0]


def test_matcher_members_of_a_budgeting_feature_matching_build(run):
    """adapter/matchers_gfo.cc + the harness compiled with -DBUDGETING_FEATURE_MATCHING -DMAX_NUM_FEATURE_MATCHING=150
    (include/ORBmatcher.h:36-37): SearchByProjection(Cur, Last) ends with the 150th match, which stays out of the rotation histogram
    (ORBmatcher.cc:1547-1552); SearchByBoW leaves a node's loop once 150 is reached, every later node still adds its first match
    (:360-365).  Against the oracle compiled the same way (orc_set_feature_budget); everything else of the program is unchanged."""
    if "budget" not in run:
        pytest.skip("tests/_build/adapter_run_budget is missing (built by __graft_entry__.build() where the reference headers are)")
    b = run["budget"]
    assert b["rc"] == 0, b["stderr"][-3000:]
    assert "[gfo]" not in b["stderr"], b["stderr"][-2000:]
    for v, want in enumerate(run["F_budget"]):
        got = np.fromfile(b["out"] / f"F{v}_out_last_idx.bin", np.int32)
        nm = int(np.fromfile(b["out"] / f"F{v}_nmatches_visible.bin", np.int32)[0])
        np.testing.assert_array_equal(got, want["idx"], err_msg=f"variant {v}")
        assert nm == want["nm"] and nm <= BUDGET, v
        assert run["F"][v]["nm"] > BUDGET                                  # the budget bites in every variant
    for tag, (nm, out), (nm_full, _) in zip(("", "_ori"), run["I_budget"], run["I"]):
        np.testing.assert_array_equal(np.fromfile(b["out"] / f"I_out_kf_idx{tag}.bin", np.int32), out, err_msg=tag)
        assert int(np.fromfile(b["out"] / f"I_nmatches{tag}.bin", np.int32)[0]) == nm
        assert nm < nm_full, (tag, nm, nm_full)
    # an extraction and an association of that build: untouched by the macro
    kd = run["oracle"].KEYPOINT_DTYPE
    assert np.fromfile(b["out"] / "A_f01_kl.bin", kd).tobytes() == run["frames"][1][0].tobytes()
    assert np.fromfile(b["out"] / "A_f01_uright.bin", np.float32).tobytes() == run["frames"][1][4][1].tobytes()


def test_compute_bow_member(run):
    """Frame::ComputeBoW() (Frame.cc:661-668) through an ORBVocabulary object: the adapter reads the tree through a derived type,
    uploads it once per context, and fills mBowVec / mFeatVec from the device's flattened maps -- word ids, WordValues (bit for bit,
    doubles), node ids and feature index lists equal the oracle's literal statement of TemplatedVocabulary::transform, for four
    vocabulary shapes x weightings x scorings"""
    assert len(run["J"]) == 4
    assert max(len(j[2]) for j in run["J"]) > 3                    # the depth-5 tree spreads the features over several nodes
    for v, (bw, bv, fn, fs, fi) in enumerate(run["J"]):
        assert len(bw) > 20 and len(fn) >= 1, v          # (levelsup 4 on a tree of depth <= 4 puts every feature under the root)
        np.testing.assert_array_equal(_rd(run, f"J{v}_bow_words.bin", np.uint32), bw, err_msg=str(v))
        assert _rd(run, f"J{v}_bow_values.bin", np.float64).tobytes() == bv.tobytes(), v
        np.testing.assert_array_equal(_rd(run, f"J{v}_fv_nodes.bin", np.uint32), fn, err_msg=str(v))
        np.testing.assert_array_equal(_rd(run, f"J{v}_fv_start.bin", np.int32), fs, err_msg=str(v))
        np.testing.assert_array_equal(_rd(run, f"J{v}_fv_items.bin", np.uint32), fi, err_msg=str(v))


def test_compute_bow_member_with_a_vocabulary_of_reference_size(run):
    """the same member on an ORBVocabulary object of ORBvoc's shape (k = 10, L = 6: 1 111 111 Node objects, a million words): the
    adapter's VocabularyView::flatten() + gfo_vocabulary_upload happen once (the first call), every later call pays the 64-node
    fingerprint and the device call; results equal the oracle's"""
    bw, bv, fn, fs, fi = run["Jbig"]
    rep = run["report"]
    assert int(rep["Jbig_nodes"]) == 1111111
    np.testing.assert_array_equal(_rd(run, "J4_bow_words.bin", np.uint32), bw)
    assert _rd(run, "J4_bow_values.bin", np.float64).tobytes() == bv.tobytes()
    np.testing.assert_array_equal(_rd(run, "J4_fv_nodes.bin", np.uint32), fn)
    np.testing.assert_array_equal(_rd(run, "J4_fv_start.bin", np.int32), fs)
    np.testing.assert_array_equal(_rd(run, "J4_fv_items.bin", np.uint32), fi)
    assert int(bw.max()) > 900000 and len(fn) > 50
    # a steady-state call does not depend on the size of the tree (the small vocabulary of J0 on the same frame)
    assert int(rep["Jbig_ComputeBoW_us"]) < 3 * int(rep["J_ComputeBoW_us"]) + 100, rep


def test_three_cameras_through_the_adapters(run):
    """part K of the program: six ORBextractor objects, three camera threads (each creating a thread per right image), one rig per
    camera -- every frame of every camera equals the single-camera results of part A; the rigs answered associations"""
    assert run["rc"] == 0, run["stderr"][-3000:]
    rep = run["report"]
    answers = [int(rep[f"K_camera{k}_rig_answers"]) for k in range(3)]
    assert all(a >= 6 for a in answers), answers           # 12 frames per camera; a rig needs a frame or two to form under load


def test_full_pyramid_mode_leaves_the_levels_in_mvImagePyramid(run, euroc_l):
    """GFO_FULL_PYRAMID=1: after operator() every mvImagePyramid[l] is a view into its 19-px framed buffer holding the level's pixels
    (ORBextractor.cc:1182-1197) -- compared with the oracle's pyramid of the same image; keypoints unchanged by the mode"""
    full = run["full"]
    assert full["rc"] == 0, full["stderr"][-3000:]
    O = run["oracle"]
    oe = O.OracleExtractor(2000, 1.2, 8, 20, 7)
    last = _frames(euroc_l, 1)                      # two frames in this run: the last one is frame 1 (padded view input)
    oe.compute_pyramid(last)
    for l in range(8):
        w, h = oe.level_size(l)
        got = np.fromfile(full["out"] / f"A_full_level{l}_{w + 38}x{h + 38}.bin", np.uint8).reshape(h + 38, w + 38)
        assert got.tobytes() == oe.level(l, padded=True).tobytes(), l
    kd = O.KEYPOINT_DTYPE
    assert np.fromfile(full["out"] / "A_f01_kl.bin", kd).tobytes() == run["frames"][1][0].tobytes()
