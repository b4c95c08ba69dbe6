"""Concurrency check of the host-array matcher calls: K host threads, each with its OWN extractor / matcher / vocabulary (contexts are
thread-compatible, not thread-safe), loop over extract -> SearchByProjection -> SearchByProjection(Cur, Last) -> ComputeBoW -> SearchByBoW
on their own inputs; every result is compared with the answer the oracle gave for those inputs before the threads started.  Anything
shared between contexts that is not meant to be (function attributes, lazily initialised statics, pinned staging) shows up as a mismatch.
usage (through gpurun): python tools/stress_matchers_threads.py [threads] [iterations] [seed]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_frame
from oracle import orb_oracle as O

O.build()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ITER = int(sys.argv[2]) if len(sys.argv) > 2 else 150
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bad = []
cases = []
for t in range(K):
    rng = np.random.default_rng(seed * 100 + t)
    w, h = [(752, 480), (640, 480), (1241, 376), (376, 240)][t % 4]
    nf = [2000, 1000, 1500, 500][t % 4]
    img = synth_frame(w, h, 300 + t)
    oe = O.OracleExtractor(nf, 1.2, 8, 20, 7)
    kp, desc = oe(img)
    n = len(kp)
    sf = oe.scale_factors
    bounds = (0.0, 0.0, float(w), float(h))
    M = int(rng.choice([800, 3000, 9000]))
    src = rng.integers(0, n, M)
    mps = np.zeros(M, O.MAP_POINT_DTYPE)
    mps["proj_x"] = kp["x"][src] + rng.normal(0, 2, M)
    mps["proj_y"] = kp["y"][src] + rng.normal(0, 2, M)
    mps["proj_xr"] = mps["proj_x"] - 5
    mps["level"] = kp["octave"][src]
    mps["view_cos"] = 1.0
    mps["flags"] = 1 | 4
    mpd = desc[src].copy()
    mpd[np.arange(M), rng.integers(0, 32, M)] ^= np.uint8(1 << int(rng.integers(0, 8)))
    th = float(rng.choice([1.0, 3.0]))
    u = np.full(n, -1, np.float32)
    ref_p = O.search_by_projection(kp, desc, u, sf, bounds, mps, mpd, th, 0.8)
    nq = min(n, 1200)
    qs = rng.choice(n, nq, replace=False)
    q = np.zeros(nq, O.PROJ_QUERY_DTYPE)
    q["u"] = kp["x"][qs] + rng.normal(0, 2, nq); q["v"] = kp["y"][qs] + rng.normal(0, 2, nq); q["ur"] = q["u"] - 5
    q["radius"] = 7.0 * sf[kp["octave"][qs]]
    q["min_level"] = kp["octave"][qs] - 1; q["max_level"] = kp["octave"][qs] + 1
    q["angle"] = kp["angle"][qs]; q["flags"] = 1 | 4
    qd = desc[qs].copy()
    ka = kp["angle"].copy()
    ref_q = O.search_by_projection_queries(kp, desc, u, ka, bounds, q, qd, False, 0.9, 100, True, None)
    voc = O.make_vocabulary(int(rng.integers(4, 11)), int(rng.integers(2, 5)), seed=t)
    lv = int(rng.integers(1, 3))
    ref_c = O.compute_bow(voc, desc, lv, 0, 1)
    fd = desc.copy(); fd[:, 5] ^= 3
    ref_cf = O.compute_bow(voc, fd, lv, 0, 1)
    kfv, ffv = (ref_c[2], ref_c[3], ref_c[4]), (ref_cf[2], ref_cf[3], ref_cf[4])
    valid = np.ones(n, np.uint8)
    ref_b = O.search_by_bow(desc, ka, valid, kfv, fd, ka, ffv, 0.7, True)
    cases.append(dict(img=img, nf=nf, kp=kp, desc=desc, sf=sf, bounds=bounds, mps=mps, mpd=mpd, th=th, u=u, ref_p=ref_p, q=q, qd=qd, ka=ka, ref_q=ref_q,
                      voc=voc, lv=lv, ref_c=ref_c, fd=fd, kfv=kfv, ffv=ffv, valid=valid, ref_b=ref_b))


def worker(t):
    c = cases[t]
    ext = G.ORBextractor(c["nf"], 1.2, 8, 20, 7)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    m9 = G.ORBmatcher(0.9, True, extractor=ext)
    m7 = G.ORBmatcher(0.7, True, extractor=ext)
    V = G.ORBVocabulary(c["voc"], ext)
    for it in range(ITER):
        kp, desc = ext(c["img"])
        if kp.tobytes() != c["kp"].tobytes() or not (desc == c["desc"]).all():
            bad.append((t, it, "extract"))
        r = m.SearchByProjection(kp, desc, c["u"], c["sf"], c["bounds"], c["mps"], c["mpd"], c["th"], None)
        if r[0] != c["ref_p"][0] or not (r[1] == c["ref_p"][1]).all() or not (r[2] == c["ref_p"][2]).all():
            bad.append((t, it, "projection"))
        r = m9.SearchByProjectionQueries(kp, desc, c["u"], c["ka"], c["bounds"], c["q"], c["qd"], False, 100, None)
        if r[0] != c["ref_q"][0] or not (r[1] == c["ref_q"][1]).all():
            bad.append((t, it, "queries"))
        (bw, bv), fv = V.compute_bow(desc, c["lv"], "TF_IDF", "L1")
        rc = c["ref_c"]
        if not (np.array_equal(bw, rc[0]) and bv.tobytes() == rc[1].tobytes() and np.array_equal(fv[0], rc[2]) and np.array_equal(fv[1], rc[3]) and np.array_equal(fv[2], rc[4])):
            bad.append((t, it, "compute_bow"))
        r = m7.SearchByBoW(desc, c["ka"], c["valid"], c["kfv"], c["fd"], c["ka"], c["ffv"])
        if r[0] != c["ref_b"][0] or not (r[1] == c["ref_b"][1]).all():
            bad.append((t, it, "bow"))
    ext.close()


t0 = time.time()
ths = [threading.Thread(target=worker, args=(t,)) for t in range(K)]
for t in ths: t.start()
for t in ths: t.join()
print(f"stress_matchers_threads: {K} threads x {ITER} iterations x 5 calls = {K * ITER * 5} calls, {len(bad)} mismatches {bad[:8]}, {time.time() - t0:.1f} s", flush=True)
sys.exit(1 if bad else 0)
