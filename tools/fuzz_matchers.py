"""Randomised parity sweep of the HOST-ARRAY matcher calls against the oracle, bit for bit (not part of the test suite: minutes of GPU +
oracle time): SearchByProjection (map-point and query form, both round-0 forms, candidate-list pools of random size), SearchByBoW (nodes
from a handful to hundreds of keypoints, both k_bow_match paths), SearchForTriangulation (random relative poses), SearchForInitialization (windows from none to the whole frame), ComputeBoW (vocabularies of random shape, 1..20 000 descriptors).
Synthetic frames: keypoints clustered like corners are, descriptors drawn around a few hundred prototypes so that small distances and
exact ties are common.   usage: python tools/fuzz_matchers.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import gf_cases
import gf_orb_slam2_amd as G
from oracle import orb_oracle as O

O.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ext = G.ORBextractor(1000, 1.2, 8, 20, 7)
SF = ext.GetScaleFactors()
bad = 0
count = {"map": 0, "query": 0, "bow": 0, "cbow": 0, "tri": 0, "init": 0}
matches = {"map": 0, "query": 0, "bow": 0, "cbow_words": 0}
t0 = time.time()


def flips(d, kmax):
    d = d.copy()
    k = int(rng.integers(0, kmax + 1))
    if k and len(d):
        fl = rng.integers(0, 256, (len(d), k))
        on = rng.random((len(d), k)) < 0.6
        for j in range(k):
            sel = on[:, j]
            d[sel, fl[sel, j] >> 3] ^= (1 << (fl[sel, j] & 7)).astype(np.uint8)
    return d


def frame(n, w, h):
    """n keypoints in clusters, descriptors around prototypes"""
    kp = np.zeros(n, O.KEYPOINT_DTYPE)
    nc = max(1, int(rng.integers(1, 60)))
    cx, cy = rng.uniform(0, w, nc), rng.uniform(0, h, nc)
    c = rng.integers(0, nc, n)
    spread = float(rng.choice([3.0, 15.0, 60.0, 400.0]))
    kp["x"] = np.clip(cx[c] + rng.normal(0, spread, n), -5, w + 5).astype(np.float32)     # a few slightly outside the bounds
    kp["y"] = np.clip(cy[c] + rng.normal(0, spread, n), -5, h + 5).astype(np.float32)
    if rng.random() < 0.3:
        kp["x"] = np.round(kp["x"])                                                       # integer positions: window-edge ties
        kp["y"] = np.round(kp["y"])
    kp["octave"] = rng.integers(0, 8, n)
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kp["size"] = 31.0
    kp["response"] = 50
    kp["class_id"] = -1
    protos = rng.integers(0, 256, (max(1, int(rng.integers(1, 300))), 32), dtype=np.uint8)
    desc = flips(protos[rng.integers(0, len(protos), n)], int(rng.integers(0, 12)))
    return kp, desc


def mismatch(what, **kw):
    global bad
    bad += 1
    print(f"MISMATCH {what}: {kw}", flush=True)


for it in range(cases):
    kind = rng.choice(["map", "query", "bow", "cbow", "tri", "init"], p=[0.3, 0.2, 0.16, 0.12, 0.11, 0.11])
    count[kind] += 1
    w, h = float(rng.choice([320, 752, 1241, 1920])), float(rng.choice([240, 480, 376, 1080]))
    bounds = (float(rng.choice([0.0, -12.5])), float(rng.choice([0.0, -7.25])), w, h)
    if kind in ("map", "query"):
        n = int(rng.choice([0, 1, 50, 700, 2000, 4000, 9000]))
        m = int(rng.choice([0, 1, 17, 500, 1500, 4000, 12000, 20000]))
        kp, desc = frame(n, w, h)
        u_right = None if rng.random() < 0.3 else np.where(rng.random(n) < 0.5, kp["x"] - rng.uniform(0, 40, n), -1).astype(np.float32)
        taken = None if rng.random() < 0.4 else (rng.random(n) < rng.uniform(0, 0.5)).astype(np.uint8)
        os.environ["GFO_PROJ_WAVE"] = str(int(rng.random() < 0.7))
        os.environ["GFO_PROJ_SPILL_CAP"] = str(int(rng.choice([0, 500, 20000, 1 << 30])))
        src = rng.integers(0, max(n, 1), m)
        jitter = float(rng.choice([0.5, 2.0, 6.0, 30.0]))
        px = (kp["x"][src] if n else np.zeros(m)) + rng.normal(0, jitter, m)
        py = (kp["y"][src] if n else np.zeros(m)) + rng.normal(0, jitter, m)
        qd = flips(desc[src] if n else np.zeros((m, 32), np.uint8), int(rng.integers(0, 40)))
        lvl = np.clip((kp["octave"][src] if n else np.zeros(m, np.int32)) + rng.integers(-1, 2, m), -1, 8)
        if kind == "map":
            mps = np.zeros(m, O.MAP_POINT_DTYPE)
            mps["proj_x"], mps["proj_y"] = px, py
            mps["proj_xr"] = px - rng.uniform(0, 40, m)
            mps["level"] = lvl
            mps["view_cos"] = rng.choice([1.0, 0.9985, 0.99, 0.5], m)
            fl = np.full(m, 1 | 4, np.int32)
            fl[rng.random(m) < 0.05] &= ~1
            fl[rng.random(m) < 0.05] |= 2
            fl[rng.random(m) < rng.uniform(0, 0.6)] &= ~4
            mps["flags"] = fl
            th = float(rng.choice([1.0, 1.0, 3.0, 5.0, 7.0, 15.0]))
            ratio = float(rng.choice([0.6, 0.8, 0.9, 1.0]))
            # (levels -1 and 8 are outside the frame's scale table: library and oracle both skip such a point)
            ref = O.search_by_projection(kp, desc, u_right, SF, bounds, mps, qd, th, ratio, taken)
            mt = G.ORBmatcher(ratio, True, extractor=ext)
            got = mt.SearchByProjection(kp, desc, u_right, SF, bounds, mps, qd, th, taken)
            if rng.random() < 0.5:
                # the good-feature build's forms of the same loop: what every point did at its turn (SearchByProjection_Budget,
                # ORBmatcher.cc:45-153), and the candidate table + per-point match driven in vector order (ORBmatcher.h:71-250)
                count["gf"] = count.get("gf", 0) + 1
                rb = O.search_by_projection_budget(kp, desc, u_right, SF, bounds, mps, qd, th, ratio, taken, 0)
                gp = mt.SearchByProjectionPoints(kp, desc, u_right, SF, bounds, mps, qd, th, taken)
                if gp[0] != rb[0] or not (gp[1] == rb[1]).all() or not (gp[2] == rb[2]).all() or not (gp[3] == rb[3]).all():
                    mismatch("points", it=it, n=n, m=m, wave=os.environ["GFO_PROJ_WAVE"], cap=os.environ["GFO_PROJ_SPILL_CAP"], got=gp[0], ref=rb[0])
                start, cand = mt.GetCandidates(kp, desc, u_right, SF, bounds, mps, qd, th, cap=int(rng.choice([0, 64, 32 * max(m, 1)])))
                slot = np.zeros(max(n, 1), np.uint8) if taken is None else taken.copy()
                okc = True
                for p in range(m):
                    r, d = mt.MatchCandidates(cand[start[p]:start[p + 1]], slot)
                    want = int(rb[3][p])
                    if r >= 0:
                        okc = okc and want == (r | (d << 16))
                        slot[r] = 1 if mps["flags"][p] & 4 else 0
                    else:
                        okc = okc and want == r
                    if not okc:
                        mismatch("table", it=it, n=n, m=m, point=p, got=r, want=want)
                        break
        else:
            q = np.zeros(m, O.PROJ_QUERY_DTYPE)
            q["u"], q["v"] = px, py
            q["ur"] = px - rng.uniform(0, 40, m)
            th = float(rng.choice([3.0, 7.0, 15.0, 40.0]))
            q["radius"] = np.where(rng.random(m) < 0.03, rng.choice([0.0, -1.0, np.nan]), th * SF[np.clip(lvl, 0, 7)]).astype(np.float32)
            mode = int(rng.integers(0, 3))                 # neutral / forward / backward level ranges (ORBmatcher.cc:1513-1521)
            q["min_level"] = lvl - 1 if mode == 0 else (lvl if mode == 1 else 0)
            q["max_level"] = lvl + 1 if mode == 0 else (-1 if mode == 1 else lvl)
            q["angle"] = rng.uniform(0, 360, m).astype(np.float32)
            fl = np.full(m, 1 | 4, np.int32)
            fl[rng.random(m) < 0.05] &= ~1
            fl[rng.random(m) < rng.uniform(0, 0.4)] &= ~4
            q["flags"] = fl
            use_ratio, ori = bool(rng.random() < 0.4), bool(rng.random() < 0.7)
            ratio = float(rng.choice([0.7, 0.9]))
            thd = int(rng.choice([0, 50, 100, 255]))
            ka = kp["angle"].copy()
            budget = int(rng.integers(1, 700)) if rng.random() < 0.3 else 0        # BUDGETING_FEATURE_MATCHING (ORBmatcher.cc:1547-1552)
            with O.feature_budget(budget):
                ref = O.search_by_projection_queries(kp, desc, u_right, ka, bounds, q, qd, use_ratio, ratio, thd, ori, taken)
            got = G.ORBmatcher(ratio, ori, extractor=ext).SearchByProjectionQueries(kp, desc, u_right, ka, bounds, q, qd, use_ratio, thd, taken, max_matches=budget)
        matches[kind] += int(ref[0])
        if got[0] != ref[0] or not (got[1] == ref[1]).all() or not (got[2] == ref[2]).all():
            mismatch(kind, it=it, n=n, m=m, wave=os.environ["GFO_PROJ_WAVE"], cap=os.environ["GFO_PROJ_SPILL_CAP"], got=got[0], ref=ref[0])
    elif kind == "bow":
        nk, nf = int(rng.choice([1, 40, 900, 2000, 5000])), int(rng.choice([1, 40, 900, 2000, 5000]))
        kk, kd = frame(nk, w, h)
        src = rng.integers(0, nk, nf)
        fd = flips(kd[src], int(rng.integers(0, 30)))
        fa = ((kk["angle"][src] + rng.normal(0, float(rng.choice([0.0, 3.0, 50.0])), nf)) % 360).astype(np.float32)
        nnodes = int(rng.choice([1, 3, 20, 100, 1000]))
        node_k = rng.integers(0, nnodes, nk).astype(np.int64) * 7
        node_f = np.where(rng.random(nf) < 0.8, node_k[src], rng.integers(0, nnodes, nf) * 7)
        node_k[rng.random(nk) < 0.03] = -1
        node_f[rng.random(nf) < 0.03] = -1
        valid = (rng.random(nk) >= rng.uniform(0, 0.5)).astype(np.uint8)
        ratio, ori = float(rng.choice([0.6, 0.75, 0.9])), bool(rng.random() < 0.7)
        kfv, ffv = O.make_feature_vector(node_k), O.make_feature_vector(node_f)
        budget = int(rng.integers(1, 700)) if rng.random() < 0.3 else 0            # BUDGETING_FEATURE_MATCHING (ORBmatcher.cc:360-365)
        with O.feature_budget(budget):
            ref = O.search_by_bow(kd, kk["angle"].copy(), valid, kfv, fd, fa, ffv, ratio, ori)
        got = G.ORBmatcher(ratio, ori, extractor=ext).SearchByBoW(kd, kk["angle"].copy(), valid, kfv, fd, fa, ffv, max_matches=budget)
        matches["bow"] += int(ref[0])
        if got[0] != ref[0] or not (got[1] == ref[1]).all():
            mismatch("bow", it=it, nk=nk, nf=nf, nnodes=nnodes, got=got[0], ref=ref[0])
        if rng.random() < 0.5:      # the keyframe-pair overload (ORBmatcher.cc:635-768) on the same arrays, a mask on the second side as well
            count["bow_kf"] = count.get("bow_kf", 0) + 1
            valid2 = (rng.random(nf) >= rng.uniform(0, 0.5)).astype(np.uint8)
            refk = O.search_by_bow_keyframes(kd, kk["angle"].copy(), valid, kfv, fd, fa, valid2, ffv, ratio, ori)
            gotk = G.ORBmatcher(ratio, ori, extractor=ext).SearchByBoWKeyFrames(kd, kk["angle"].copy(), valid, kfv, fd, fa, valid2, ffv)
            if gotk[0] != refk[0] or not (gotk[1] == refk[1]).all():
                mismatch("bow_kf", it=it, nk=nk, nf=nf, nnodes=nnodes, got=gotk[0], ref=refk[0])
    elif kind == "init":
        # SearchForInitialization (ORBmatcher.cc:520-633): windows around vbPrevMatched, thefts, two calls in a row on the updated vector
        n = int(rng.choice([1, 40, 900, 2000, 4000]))
        k1, d1 = frame(n, w, h)
        k1["octave"] = rng.choice([0, 0, 0, 1, 3], n)
        k2, d2, prev = gf_cases.initialization_case(O, k1, d1, rng, flips=int(rng.integers(0, 20)), sigma=float(rng.choice([0.0, 4.0, 20.0, 80.0])),
                                                    resample=bool(rng.random() < 0.7))
        n2 = int(rng.choice([n, max(1, n // 3)]))
        k2, d2 = k2[:n2], d2[:n2]
        win, ratio, ori = int(rng.choice([0, 10, 100, 100, 2000])), float(rng.choice([0.6, 0.9, 1.0])), bool(rng.random() < 0.6)
        pr, pg = prev.copy(), prev.copy()
        for rep in range(2):
            ref = O.search_for_initialization(k1, d1, pr, k2, d2, bounds, win, ratio, ori)
            got = G.ORBmatcher(ratio, ori, extractor=ext).SearchForInitialization(k1, d1, pg, k2, d2, bounds, win)
            matches["init"] = matches.get("init", 0) + int(ref[0])
            if got[0] != ref[0] or not (got[1] == ref[1]).all() or pr.tobytes() != pg.tobytes():
                mismatch("init", it=it, n=n, n2=n2, win=win, ratio=ratio, ori=ori, rep=rep, got=got[0], ref=ref[0])
                break
    elif kind == "tri":
        # SearchForTriangulation (ORBmatcher.cc:770-935): a second view of a synthetic keyframe under a random relative pose
        n = int(rng.choice([1, 40, 900, 2000, 4000]))
        k1, d1 = frame(n, w, h)
        c = gf_cases.triangulation_case(O, k1, d1, rng, flips=int(rng.integers(0, 16)), noise=float(rng.choice([0.0, 0.5, 2.0])), p_mp=float(rng.uniform(0, 0.8)),
                                        p_stereo=float(rng.choice([0.0, 0.5, 1.0])), node_shift=int(rng.choice([0, 2, 4, 6, 8])), forward=bool(rng.random() < 0.5))
        only, ori, mono = bool(rng.random() < 0.3), bool(rng.random() < 0.6), bool(rng.random() < 0.2)
        u1, u2 = (None, None) if mono else (c["ur1"], c["ur2"])
        sg = (SF * SF).astype(np.float32)
        a = (c["kp1"], c["desc1"], c["has1"], u1, c["fv1"], c["kp2"], c["desc2"], c["has2"], u2, c["fv2"], SF, sg, c["f12"], c["ex"], c["ey"])
        ref = O.search_for_triangulation(*a, only, ori)
        got = G.ORBmatcher(0.6, ori, extractor=ext).SearchForTriangulation(*a, only)
        matches["tri"] = matches.get("tri", 0) + int(ref[0])
        if got[0] != ref[0] or not (got[1] == ref[1]).all():
            mismatch("tri", it=it, n=n, only=only, ori=ori, mono=mono, got=got[0], ref=ref[0])
    else:
        k, depth = int(rng.integers(2, 12)), int(rng.integers(1, 6))
        while k ** depth > 200000:
            depth -= 1
        voc = O.make_vocabulary(k, depth, seed=int(rng.integers(0, 1 << 30)), p_stop=float(rng.choice([0.0, 0.1, 0.5])))
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, 2048, 2049, 8192, 8193, 20000]))
        leaves = voc["descriptors"][voc["n_children"] == 0]
        desc = flips(leaves[rng.integers(0, len(leaves), n)], int(rng.integers(0, 20)))
        wname, nname = str(rng.choice(["TF_IDF", "TF", "IDF", "BINARY"])), rng.choice([None, "L1", "L2"])
        W, Nn = {"TF_IDF": 0, "TF": 1, "IDF": 2, "BINARY": 3}[wname], {None: 0, "L1": 1, "L2": 2}[nname]
        levelsup = int(rng.integers(0, depth + 2))
        V = G.ORBVocabulary(voc, ext)
        (bw, bv), (fn, fs, fi) = V.compute_bow(desc, levelsup, wname, nname)
        rw, rv, rn, rs, ri = O.compute_bow(voc, desc, levelsup, W, Nn)
        matches["cbow_words"] += len(rw)
        if not (np.array_equal(bw, rw) and bv.tobytes() == rv.tobytes() and np.array_equal(fn, rn) and np.array_equal(fs, rs) and np.array_equal(fi, ri)):
            mismatch("cbow", it=it, k=k, depth=depth, n=n, weighting=wname, norm=nname, levelsup=levelsup)
    if (it + 1) % 50 == 0:
        print(f"{it + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s  {count}", flush=True)
print(f"fuzz_matchers: {cases} cases {count}, accepted matches / words in the oracle's answers {matches}, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
ext.close()
sys.exit(1 if bad else 0)
