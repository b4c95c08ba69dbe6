#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ from the CPU oracle (oracle/orb_oracle.c).

The reference holds no golden keypoints / descriptors / matches, and its own code
cannot be built here (no OpenCV), so these vectors pin THE ORACLE, variant
  [OCV] table : oracle/ocv_variants.json as committed (resize 11-bit fixed point; Gaussian taps {18,34,49,55,49,34,18},
                exact accumulation, (v + 2^15) >> 16; fastAtan2 un-fused) -- written into variants.json next to the
                vectors; tests/golden/check_against_cv2.py is what decides whether that table is the reference's
  sincos : correctly rounded (long double, rounded once), rotation un-fused
  quadtree tie-break : equal-sized nodes split newest first
against accidental change, and give the GPU tests a reference that needs no oracle run.
Inputs: the reference's own test images test/EuRoC_l.png / EuRoC_r.png (raw dumps).

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import orb_oracle as O  # noqa: E402

FX, BF = 435.2046959714599, 47.90639384423901


def make_projection_case(kp, desc, seed=7, m=5000):
    """S3 of SURVEY.md 8d, scaled down: a local map of m points, a 'visible' subset built from the
    frame's own descriptors with k~U{0..60} random bits flipped, projections = keypoint position
    + N(0, 2 px), the rest uniform random descriptors."""
    rng = np.random.default_rng(seed)
    n = len(kp)
    nvis = min(n, m // 4)
    vis = rng.choice(n, nvis, replace=False)
    mps = np.zeros(m, O.MAP_POINT_DTYPE)
    mpd = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    slots = rng.choice(m, nvis, replace=False)
    for s, i in zip(slots, vis):
        d = desc[i].copy()
        k = int(rng.integers(0, 61))
        bits = rng.choice(256, k, replace=False)
        for b in bits:
            d[b >> 3] ^= np.uint8(1 << (b & 7))
        mpd[s] = d
        mps["proj_x"][s] = kp["x"][i] + rng.normal(0, 2)
        mps["proj_y"][s] = kp["y"][i] + rng.normal(0, 2)
        mps["level"][s] = kp["octave"][i]
    rest = np.setdiff1d(np.arange(m), slots)
    mps["proj_x"][rest] = rng.uniform(0, 752, len(rest))
    mps["proj_y"][rest] = rng.uniform(0, 480, len(rest))
    mps["level"][rest] = rng.integers(0, 8, len(rest))
    mps["proj_xr"] = mps["proj_x"] - rng.uniform(0, 40, m).astype(np.float32)
    mps["view_cos"] = np.where(rng.random(m) < 0.5, 1.0, 0.99).astype(np.float32)
    flags = np.full(m, 1 | 4, np.int32)
    flags[rng.random(m) < 0.03] = 0          # not in view
    flags[rng.random(m) < 0.02] |= 2         # bad
    flags[rng.random(m) < 0.05] &= ~4        # no observations: does not block later points
    mps["flags"] = flags
    return mps, mpd


def main():
    import json
    oe = O.OracleExtractor(2000, 1.2, 8, 20, 7)
    with open(os.path.join(HERE, "variants.json"), "w") as f:      # the table the vectors below were made under
        json.dump({"ocv": O.get_ocv_variants(), "sincos": "correctly rounded", "rotation": "un-fused", "quadtree_ties": "newest first"}, f, indent=1)
    out = {}
    for side in ("l", "r"):
        img = np.fromfile(os.path.join(HERE, f"EuRoC_{side}_752x480.u8"), np.uint8).reshape(480, 752)
        kp, desc = oe(img)
        out[side] = (kp, desc)
        kp.tofile(os.path.join(HERE, f"EuRoC_{side}_kp.bin"))
        desc.tofile(os.path.join(HERE, f"EuRoC_{side}_desc.bin"))
        per_level = np.array([oe.level_keypoint_count(l) for l in range(8)], np.int32)
        ncand = np.array([len(oe.level_candidates(l)) for l in range(8)], np.int32)
        np.savez(os.path.join(HERE, f"EuRoC_{side}_levels.npz"), per_level=per_level, ncand=ncand)
        print(side, len(kp), per_level, ncand)
    (kl, dl), (kr, dr) = out["l"], out["r"]
    sf = oe.scale_factors
    nm, u, dp, bd, bi = O.stereo_match(kl, dl, kr, dr, sf, 480, BF, BF / FX, 0.0)
    np.savez(os.path.join(HERE, "EuRoC_stereo.npz"), nmatched=nm, u_right=u, depth=dp, best_dist=bd, best_idx=bi)
    print("stereo nmatched", nm, "with depth", int((dp > 0).sum()))
    mps, mpd = make_projection_case(kl, dl)
    bounds = (0.0, 0.0, 752.0, 480.0)
    taken = (np.random.default_rng(5).random(len(kl)) < 0.1).astype(np.uint8)
    nmm, out_mp, out_sc = O.search_by_projection(kl, dl, u, sf, bounds, mps, mpd, 3.0, 0.8, taken)
    np.savez(os.path.join(HERE, "EuRoC_projection.npz"), mps=mps, mp_desc=mpd, taken=taken, nmatches=nmm,
             out_mp=out_mp, out_score=out_sc)
    print("projection nmatches", nmm)


if __name__ == "__main__":
    main()
