"""(No oracle in here: bench.py runs this file as a child for its `matcher_calls` block.)
Per-call latency of the host-array matcher calls Tracking makes on every frame (caller arrays in, caller arrays out, one
synchronisation): SearchByProjection(F, local map) for several map sizes, SearchByProjection(Cur, Last), SearchByBoW(KF, F),
ComputeBoW(F) -- on the EuRoC frame (2008 keypoints).  Every map point here imitates a random keypoint (8 flipped bits, N(0,2) px
away), so with M > N several points COMPETE for one keypoint: the ordered resolve runs its worst case, not a typical local map.
usage (through gpurun): [TH=1|3|5] [ONLY=map|gf|stereo|last|init|cbow|bow|tri] [JSON=1] python tools/matcher_call_latency.py [M ...]     (GFO_PROJ_STATS=1 prints
rounds / fallbacks per call; JSON=1: one JSON object on stdout instead of the text lines)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_vocabulary

JSON = os.environ.get("JSON") == "1"
out = {"frame": "EuRoC_l 752x480, extractor (2000, 1.2, 8, 20, 7)", "unit": "ms per call, median of 50", "calls": []}


def say(name, ms, detail, text):
    out["calls"].append(dict({"call": name, "ms": round(float(ms), 4)}, **detail))
    if not JSON:
        print(text, flush=True)

img = np.fromfile("tests/golden/EuRoC_l_752x480.u8", np.uint8).reshape(480, 752)
ext = G.ORBextractor(2000, 1.2, 8, 20, 7)
kp, desc = ext(img)
n = len(kp)
rng = np.random.default_rng(1)
m = G.ORBmatcher(0.8, True, extractor=ext)
sf = ext.GetScaleFactors()
bounds = (0.0, 0.0, 752.0, 480.0)
u_right = np.full(n, -1, np.float32)
ONLY = os.environ.get("ONLY", "")       # map | last | cbow | bow: time (and trace) one call only
TH = float(os.environ.get("TH", "1"))   # the call's `th`: 1 in TrackLocalMap (3 for RGB-D, 5 after a relocalisation)


def flipped(d, k):
    d = d.copy()
    fl = rng.integers(0, 256, (len(d), k))
    for j in range(k):
        d[np.arange(len(d)), fl[:, j] >> 3] ^= (1 << (fl[:, j] & 7)).astype(np.uint8)
    return d


def median_ms(call, reps=50):
    for _ in range(5):
        r = call()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = call()
        t.append(time.perf_counter() - t0)
    return np.median(t) * 1e3, r


for M in ([int(a) for a in sys.argv[1:]] or (1000, 2000, 4000)) if ONLY in ("", "map") else ():
    mps = np.zeros(M, G.MAP_POINT_DTYPE)
    src = rng.integers(0, n, M)
    mpd = flipped(desc[src], 8)
    mps["proj_x"] = kp["x"][src] + rng.normal(0, 2, M)
    mps["proj_y"] = kp["y"][src] + rng.normal(0, 2, M)
    mps["proj_xr"] = mps["proj_x"] - 5
    mps["level"] = kp["octave"][src]
    mps["view_cos"] = 1.0
    mps["flags"] = 1 | 4
    ms, r = median_ms(lambda: m.SearchByProjection(kp, desc, u_right, sf, bounds, mps, mpd, TH, None))
    say("SearchByProjection(F, MapPoints)", ms, {"map_points": M, "th": TH, "keypoints": n, "matches": int(r[0])},
        f"SearchByProjection(F, {M} map points, th {TH:g}), {n} keypoints: median {ms:.3f} ms, {r[0]} matches")

# the good-feature build's calls on a 2000-point map (GOOD_FEATURE_MAP_MATCHING, the reference's default): SearchByProjection_Budget's
# device call (what every point did at its turn) and the candidate table behind SearchByProjection_OnePoint, then 2000 picks from it
if ONLY in ("", "gf"):
    M = 2000
    mps = np.zeros(M, G.MAP_POINT_DTYPE)
    src = rng.integers(0, n, M)
    mpd = flipped(desc[src], 8)
    mps["proj_x"] = kp["x"][src] + rng.normal(0, 2, M)
    mps["proj_y"] = kp["y"][src] + rng.normal(0, 2, M)
    mps["proj_xr"] = mps["proj_x"] - 5
    mps["level"] = kp["octave"][src]
    mps["view_cos"] = 1.0
    mps["flags"] = 1 | 4
    ms, r = median_ms(lambda: m.SearchByProjectionPoints(kp, desc, u_right, sf, bounds, mps, mpd, TH, None))
    say("SearchByProjection_Budget (gfo_search_by_projection_points)", ms, {"map_points": M, "th": TH, "keypoints": n, "matches": int(r[0])},
        f"SearchByProjection_Budget's call ({M} map points, th {TH:g}, per-point outcomes): median {ms:.3f} ms, {r[0]} matches")
    ms, r = median_ms(lambda: m.GetCandidates(kp, desc, u_right, sf, bounds, mps, mpd, TH))
    say("GetCandidates for every map point (gfo_projection_candidates)", ms, {"map_points": M, "th": TH, "keypoints": n, "entries": int(len(r[1]))},
        f"candidate table ({M} map points, th {TH:g}): median {ms:.3f} ms, {len(r[1])} entries")

# Frame::ComputeStereoMatches_Undistorted on caller arrays (what the adapter falls back on when the rig cannot answer from the extraction)
if ONLY in ("", "stereo"):
    imr = np.fromfile("tests/golden/EuRoC_r_752x480.u8", np.uint8).reshape(480, 752)
    kr, dr = ext(imr)
    prm = G.StereoParams(480, 47.90639384423901, 47.90639384423901 / 435.2046959714599, 0.0)
    ms, r = median_ms(lambda: m.ComputeStereoMatches(kp, desc, kr, dr, sf, prm))
    say("ComputeStereoMatches(host arrays)", ms, {"left": n, "right": len(kr), "matches": int(r[0])},
        f"ComputeStereoMatches on host arrays: {n} x {len(kr)} keypoints: median {ms:.3f} ms, {r[0]} matches")

# SearchByProjection(CurrentFrame, LastFrame, th = 7 mono / 15 stereo): one query per tracked point of the last frame (ORBmatcher.cc:1440-1593)
nq = 1500
src = rng.choice(n, nq, replace=False)
q = np.zeros(nq, G.PROJ_QUERY_DTYPE)
q["u"] = kp["x"][src] + rng.normal(0, 2, nq)
q["v"] = kp["y"][src] + rng.normal(0, 2, nq)
q["ur"] = q["u"] - 5
q["radius"] = 7.0 * sf[kp["octave"][src]]
q["min_level"] = kp["octave"][src] - 1
q["max_level"] = kp["octave"][src] + 1
q["angle"] = kp["angle"][src]
q["flags"] = 1 | 4
qd = flipped(desc[src], 8)
if ONLY in ("", "last"):
    ms, r = median_ms(lambda: m.SearchByProjectionQueries(kp, desc, u_right, kp["angle"].copy(), bounds, q, qd))
    say("SearchByProjection(Cur, Last)", ms, {"tracked_points": nq, "th": 7, "keypoints": n, "matches": int(r[0])},
        f"SearchByProjection(Cur, Last): {nq} tracked points, th 7, rotation check: median {ms:.3f} ms, {r[0]} matches")

if ONLY in ("", "init"):
    # SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, 100) (monocular bootstrap): F2 = F1 moved by N(0, 10) px, 6 bits flipped
    kpi = kp.copy()
    kpi["x"] = (kp["x"] + rng.normal(0, 10, n)).astype(np.float32)
    kpi["y"] = (kp["y"] + rng.normal(0, 10, n)).astype(np.float32)
    fdi = flipped(desc, 6)
    prev0 = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
    mi = G.ORBmatcher(0.9, True, extractor=ext)
    ms, r = median_ms(lambda: mi.SearchForInitialization(kp, desc, prev0.copy(), kpi, fdi, bounds, 100))
    say("SearchForInitialization(F1, F2)", ms, {"keypoints": n, "level0": int((kp["octave"] == 0).sum()), "window": 100, "matches": int(r[0])},
        f"SearchForInitialization(F1, F2): {int((kp['octave'] == 0).sum())} level-0 keypoints of {n}, window 100: median {ms:.3f} ms, {r[0]} matches")

if ONLY not in ("", "cbow", "bow", "tri"):
    ext.close()
    if JSON:
        print(json.dumps(out), flush=True)
    sys.exit(0)
# Frame::ComputeBoW and SearchByBoW(KF, F): a random 10-ary, 4-level vocabulary (the ORBvoc shape is 10^6 leaves; the tree is not in the repo)
voc = synth_vocabulary(10, 4, seed=0)
V = G.ORBVocabulary(voc, ext)
fd = flipped(desc, 6)
ms, r = median_ms(lambda: V.compute_bow(desc, 2, "TF_IDF", "L1"), 50 if ONLY in ("", "cbow") else 1)
(bw, bv), kfv = r
if ONLY in ("", "cbow"):
    say("ComputeBoW", ms, {"descriptors": n, "k": 10, "L": 4, "levelsup": 2, "words": len(bw), "feature_vector_nodes": len(kfv[0])},
        f"ComputeBoW({n} descriptors, k=10 L=4, levelsup 2): median {ms:.3f} ms, {len(bw)} words, {len(kfv[0])} feature-vector nodes")
ffv = V.compute_bow(fd, 2, "TF_IDF", "L1")[1]
valid = np.ones(n, np.uint8)
mb = G.ORBmatcher(0.7, True, extractor=ext)
if ONLY in ("", "bow"):
    ms, r = median_ms(lambda: mb.SearchByBoW(desc, kp["angle"].copy(), valid, kfv, fd, kp["angle"].copy(), ffv))
    say("SearchByBoW(KF, F)", ms, {"keypoints": n, "nodes": len(kfv[0]), "matches": int(r[0])},
        f"SearchByBoW(KF, F): {n} x {n} keypoints over {len(kfv[0])} nodes: median {ms:.3f} ms, {r[0]} matches")
if ONLY in ("", "bow", "tri"):
    # SearchForTriangulation(KF1, KF2, F12, ...) (local mapping): the second keyframe = the first displaced along x (F12 of a pure
    # x translation: the epipolar line of a keypoint is its own row), rows disturbed by N(0, 0.7) px; half the keypoints of either side
    # have map points already
    kp2 = kp.copy()
    kp2["x"] = (kp["x"] - rng.uniform(2, 40, n)).astype(np.float32)
    kp2["y"] = (kp["y"] + rng.normal(0, 0.7, n)).astype(np.float32)
    has1, has2 = (rng.random(n) < 0.5).astype(np.uint8), (rng.random(n) < 0.5).astype(np.uint8)
    f12 = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32)
    sg = (sf * sf).astype(np.float32)
    mt = G.ORBmatcher(0.6, False, extractor=ext)
    ms, r = median_ms(lambda: mt.SearchForTriangulation(kp, desc, has1, u_right, kfv, kp2, fd, has2, u_right, ffv, sf, sg, f12, -1e5, 240.0))
    say("SearchForTriangulation(KF1, KF2)", ms, {"keypoints": n, "nodes": len(kfv[0]), "matches": int(r[0])},
        f"SearchForTriangulation(KF1, KF2): {n} x {n} keypoints over {len(kfv[0])} nodes: median {ms:.3f} ms, {r[0]} pairs")
# the same two calls with a vocabulary of the SIZE the reference loads (ORBvoc: k = 10, L = 6 -- 1 111 111 nodes, 35.5 MB of centres;
# test/test_Stereo.cpp:87), levelsup 4 as Frame.cc:666 passes it: the upload happens once per context, the call does not grow with it
if ONLY in ("", "cbow", "bow", "bigvoc", "tri"):
    import time
    big = synth_vocabulary(10, 6, seed=1)
    t0 = time.perf_counter()
    VB = G.ORBVocabulary(big, ext)
    up_ms = (time.perf_counter() - t0) * 1e3
    ms, r = median_ms(lambda: VB.compute_bow(desc, 4, "TF_IDF", "L1"), 50)
    (bwb, bvb), kfvb = r
    say("ComputeBoW (vocabulary of ORBvoc's size)", ms, {"descriptors": n, "k": 10, "L": 6, "levelsup": 4, "nodes": int(len(big["first_child"])),
                                                         "vocabulary_upload_ms": round(up_ms, 2), "words": len(bwb), "feature_vector_nodes": len(kfvb[0])},
        f"ComputeBoW({n} descriptors, k=10 L=6 = {len(big['first_child'])} nodes, levelsup 4): median {ms:.3f} ms, upload once {up_ms:.1f} ms, "
        f"{len(bwb)} words, {len(kfvb[0])} feature-vector nodes")
    ffvb = VB.compute_bow(fd, 4, "TF_IDF", "L1")[1]
    ms, r = median_ms(lambda: mb.SearchByBoW(desc, kp["angle"].copy(), valid, kfvb, fd, kp["angle"].copy(), ffvb))
    say("SearchByBoW(KF, F) (vocabulary of ORBvoc's size)", ms, {"keypoints": n, "nodes": len(kfvb[0]), "matches": int(r[0])},
        f"SearchByBoW(KF, F) over {len(kfvb[0])} level-2 nodes of that vocabulary: median {ms:.3f} ms, {r[0]} matches")
    ms, r = median_ms(lambda: mt.SearchForTriangulation(kp, desc, has1, u_right, kfvb, kp2, fd, has2, u_right, ffvb, sf, sg, f12, -1e5, 240.0))
    say("SearchForTriangulation(KF1, KF2) (vocabulary of ORBvoc's size)", ms, {"keypoints": n, "nodes": len(kfvb[0]), "matches": int(r[0])},
        f"SearchForTriangulation(KF1, KF2) over {len(kfvb[0])} level-2 nodes of that vocabulary: median {ms:.3f} ms, {r[0]} pairs")
ext.close()
if JSON:
    print(json.dumps(out), flush=True)
