"""A slice of the fuzz generators (tools/fuzz_parity.py, fuzz_matchers.py, fuzz_stereo.py) run against the ORACLE ALONE -- no GPU:
tests/test_sanitizers.py starts this file in a Python that has libasan preloaded and ORB_ORACLE_LIB pointing at the
-fsanitize=address,undefined build of oracle/orb_oracle.c.  Nothing is compared here; the sanitizers are the assertion (a report
aborts the process).   usage: python tests/san/oracle_fuzz_slice.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orb_oracle as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
SF = O.OracleExtractor(500, 1.2, 8, 20, 7).scale_factors
done = {"extract": 0, "stereo": 0, "stereo_frame": 0, "map": 0, "query": 0, "kf": 0, "bow": 0, "cbow": 0, "fold": 0, "area": 0}


def flips(d, kmax):
    d = d.copy()
    k = int(rng.integers(0, kmax + 1))
    if k and len(d):
        fl = rng.integers(0, 256, (len(d), k))
        for j in range(k):
            d[np.arange(len(d)), fl[:, j] >> 3] ^= (1 << (fl[:, j] & 7)).astype(np.uint8)
    return d


def frame(n, w, h):
    kp = np.zeros(n, O.KEYPOINT_DTYPE)
    nc = max(1, int(rng.integers(1, 40)))
    cx, cy = rng.uniform(0, w, nc), rng.uniform(0, h, nc)
    c = rng.integers(0, nc, n)
    spread = float(rng.choice([3.0, 15.0, 60.0, 400.0]))
    kp["x"] = np.clip(cx[c] + rng.normal(0, spread, n), -5, w + 5).astype(np.float32)
    kp["y"] = np.clip(cy[c] + rng.normal(0, spread, n), -5, h + 5).astype(np.float32)
    if rng.random() < 0.3:
        kp["x"], kp["y"] = np.round(kp["x"]), np.round(kp["y"])
    kp["octave"] = rng.integers(0, 8, n)
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    kp["size"], kp["response"], kp["class_id"] = 31.0, 50, -1
    protos = rng.integers(0, 256, (max(1, int(rng.integers(1, 200))), 32), dtype=np.uint8)
    return kp, flips(protos[rng.integers(0, len(protos), n)], int(rng.integers(0, 12)))


def image(w, h):
    kind = rng.integers(0, 4)
    if kind == 0:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1:
        img = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    else:
        img = (rng.normal(128, 20, (h, w))).clip(0, 255).astype(np.uint8)
        for _ in range(int(rng.integers(0, 30))):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            img[y0:y0 + int(rng.integers(3, 40)), x0:x0 + int(rng.integers(3, 40))] = int(rng.integers(0, 256))
    return np.ascontiguousarray(img)


for case in range(cases):
    what = case % 10
    if what == 0:                                   # extraction: odd sizes down to images with no cell at all, 1..8 levels
        w, h = int(rng.integers(20, 260)), int(rng.integers(20, 200))
        e = O.OracleExtractor(int(rng.integers(1, 800)), float(rng.choice([1.1, 1.2, 1.5, 2.0])), int(rng.integers(1, 9)), int(rng.integers(2, 60)),
                              int(rng.integers(1, 20)))
        img = image(w, h)
        kp, d = e(img)
        for l in range(e.nlevels):
            e.level(l); e.level(l, padded=True)
            if len(kp):
                e.level(l, blurred=True)
            e.level_candidates(l)
        done["extract"] += 1
    elif what in (1, 2):                            # stereo association, fresh and as member state over several calls
        nl, nr = int(rng.integers(0, 400)), int(rng.integers(0, 400))
        w, h = 752, int(rng.integers(8, 481))
        (kl, dl), (kr, dr) = frame(nl, w, h), frame(nr, w, h)
        kl["y"][rng.random(nl) < 0.05] += 2000      # rows outside the image
        win = rng.random() < 0.5
        md = rng.uniform(0, 60, nl).astype(np.float32) if win else None
        xd = (md + rng.uniform(-5, 120, nl)).astype(np.float32) if win else None
        if what == 1:
            O.stereo_match(kl, dl, kr, dr, SF, h, 47.9, 0.11, float(rng.choice([0.0, 30.0])), md, xd, online=bool(rng.integers(0, 2)))
            done["stereo"] += 1
        else:
            F = O.StereoFrame(kl, dl, kr, dr, SF, h, 47.9, 0.11, delayed=bool(rng.integers(0, 2)))
            for _ in range(int(rng.integers(1, 5))):
                if rng.random() < 0.2:
                    F.prepare()
                if rng.random() < 0.2:
                    F.clear_matched()
                F.match(md, xd, (rng.random(nl) < 0.5).astype(np.uint8), online=bool(rng.integers(0, 2)))
            done["stereo_frame"] += 1
    elif what in (3, 4, 5):                         # the three projection searches
        n, m = int(rng.integers(0, 500)), int(rng.integers(0, 700))
        w, h = float(rng.choice([752, 1920])), float(rng.choice([480, 1080]))
        kp, desc = frame(n, w, h)
        src = rng.integers(0, max(n, 1), m)
        ur = np.where(rng.random(n) < 0.5, kp["x"] - rng.uniform(0, 40, n), -1).astype(np.float32)
        taken = (rng.random(n) < 0.1).astype(np.uint8)
        bounds = (0.0, 0.0, w, h)
        if what == 3:
            mps = np.zeros(m, O.MAP_POINT_DTYPE)
            if n:
                mps["proj_x"] = kp["x"][src] + rng.normal(0, 3, m); mps["proj_y"] = kp["y"][src] + rng.normal(0, 3, m)
            mps["proj_xr"] = mps["proj_x"] - rng.uniform(0, 40, m)
            mps["level"] = rng.integers(-2, 11, m)                  # some outside the scale table: skipped
            mps["view_cos"] = rng.choice([1.0, 0.99, 0.5], m)
            mps["flags"] = rng.integers(0, 8, m)
            mpd = flips(desc[src], 10) if n else rng.integers(0, 256, (m, 32), dtype=np.uint8)
            O.search_by_projection(kp, desc, ur, SF, bounds, mps, mpd, float(rng.choice([1.0, 3.0, 7.0])), 0.8, taken)
            done["map"] += 1
        else:
            q = np.zeros(m, O.PROJ_QUERY_DTYPE)
            if n:
                q["u"] = kp["x"][src] + rng.normal(0, 3, m); q["v"] = kp["y"][src] + rng.normal(0, 3, m)
                lv = kp["octave"][src]
            else:
                lv = np.zeros(m, np.int32)
            q["ur"] = q["u"] - rng.uniform(0, 40, m)
            q["radius"] = rng.choice([0.0, 3.0, 7.0, 15.0, 200.0], m).astype(np.float32)
            q["min_level"] = lv - rng.integers(0, 3, m); q["max_level"] = np.where(rng.random(m) < 0.2, -1, lv + rng.integers(0, 3, m))
            q["angle"] = rng.uniform(0, 360, m).astype(np.float32)
            q["flags"] = rng.integers(0, 8, m)
            q["u"][rng.random(m) < 0.02] = np.nan
            q["radius"][rng.random(m) < 0.02] = np.inf
            qd = flips(desc[src], 10) if n else rng.integers(0, 256, (m, 32), dtype=np.uint8)
            if what == 4:
                O.search_by_projection_queries(kp, desc, ur if rng.random() < 0.7 else None, kp["angle"], bounds, q, qd, bool(rng.integers(0, 2)), 0.9,
                                               int(rng.integers(0, 256)), bool(rng.integers(0, 2)), taken)
                done["query"] += 1
            else:
                O.search_by_projection_kf(kp, desc, kp["angle"], bounds, q, qd, int(rng.integers(0, 256)), bool(rng.integers(0, 2)), taken)
                done["kf"] += 1
        if n and what == 5:                         # SearchForInitialization: the frame against a shuffled half of itself, two calls in a row
            n2 = max(1, n // 2)
            sel = rng.permutation(n)[:n2]
            prev = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
            prev[rng.random(n) < 0.02] = np.nan
            k1 = kp.copy()
            k1["octave"] = rng.choice([0, 0, 1], n)
            for _ in range(2):
                O.search_for_initialization(k1, desc, prev, kp[sel], flips(desc[sel], 8), bounds, int(rng.choice([0, 10, 100, 5000])),
                                            float(rng.choice([0.6, 0.9])), bool(rng.integers(0, 2)))
            done["init"] = done.get("init", 0) + 1
        if n:
            O.features_in_area(kp, bounds, float(rng.uniform(-50, w + 50)), float(rng.uniform(-50, h + 50)), float(rng.choice([0.0, 5.0, 1e9])),
                               int(rng.integers(-1, 8)), int(rng.integers(-1, 8)))
            done["area"] += 1
    elif what in (6, 7):                            # SearchByBoW over random CSRs (empty nodes, missing keypoints, disjoint vocabularies)
        nk, nf = int(rng.integers(0, 500)), int(rng.integers(0, 500))
        (kk, dk), (kf, df) = frame(nk, 752, 480), frame(nf, 752, 480)
        shift = int(rng.integers(0, 8))
        node_k = (dk[:, 0] >> shift).astype(np.int64) if nk else np.zeros(0, np.int64)
        node_f = (df[:, 0] >> shift).astype(np.int64) + int(rng.choice([0, 0, 0, 1000])) if nf else np.zeros(0, np.int64)
        node_k[rng.random(nk) < 0.05] = -1
        O.search_by_bow(dk, kk["angle"], (rng.random(nk) < 0.9).astype(np.uint8), O.make_feature_vector(node_k), df, kf["angle"],
                        O.make_feature_vector(node_f), float(rng.choice([0.6, 0.75, 0.9])), bool(rng.integers(0, 2)))
        done["bow"] += 1
        if nk and nf:                               # the keyframe-pair overload and SearchForTriangulation on the same two keyframes
            v1, v2 = (rng.random(nk) < 0.8).astype(np.uint8), (rng.random(nf) < 0.8).astype(np.uint8)
            O.search_by_bow_keyframes(dk, kk["angle"], v1, O.make_feature_vector(node_k), df, kf["angle"], v2, O.make_feature_vector(node_f),
                                      float(rng.choice([0.6, 0.9])), bool(rng.integers(0, 2)))
            mono = rng.random() < 0.3
            u1 = None if mono else np.where(rng.random(nk) < 0.5, kk["x"] - 5, -1).astype(np.float32)
            u2 = None if mono else np.where(rng.random(nf) < 0.5, kf["x"] - 5, -1).astype(np.float32)
            f12 = rng.normal(0, 1e-3, 9).astype(np.float32) if rng.random() < 0.9 else np.zeros(9, np.float32)
            O.search_for_triangulation(kk, dk, 1 - v1, u1, O.make_feature_vector(node_k), kf, df, 1 - v2, u2, O.make_feature_vector(node_f), SF,
                                       (SF * SF).astype(np.float32), f12, float(rng.uniform(-100, 900)), float(rng.uniform(-100, 600)),
                                       bool(rng.integers(0, 2)), bool(rng.integers(0, 2)))
            done["bow_kf_tri"] = done.get("bow_kf_tri", 0) + 1
    elif what == 8:                                 # ComputeBoW on ragged vocabularies of random shape
        voc = O.make_vocabulary(int(rng.integers(2, 11)), int(rng.integers(1, 5)), seed=int(rng.integers(0, 1 << 30)), p_stop=float(rng.choice([0, 0.1, 0.9])))
        n = int(rng.integers(0, 600))
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        O.compute_bow(voc, desc, int(rng.integers(0, 8)), int(rng.integers(0, 4)), int(rng.integers(0, 3)))
        O.bow_transform(voc, desc, int(rng.integers(0, 8)))
        done["cbow"] += 1
    else:                                           # the fold alone: streams with repeated words, zero weights, one element, none
        n = int(rng.integers(0, 3000))
        word = rng.integers(0, max(1, int(rng.integers(1, 5000))), n).astype(np.uint32)
        weight = np.where(rng.random(n) < 0.1, 0.0, rng.uniform(0.01, 9.0, n))
        node = rng.integers(0, max(1, int(rng.integers(1, 300))), n).astype(np.uint32)
        O.bow_fold(word, weight, node, int(rng.integers(0, 4)), int(rng.integers(0, 3)))
        done["fold"] += 1
print("ok " + " ".join(f"{k}={v}" for k, v in done.items()), flush=True)
