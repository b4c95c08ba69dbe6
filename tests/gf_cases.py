"""Shared inputs of the good-feature matcher tests (tests/test_oracle.py, tests/test_gpu_gf_matchers.py): a frame from the golden
EuRoC extraction and a contended local map -- several map points per keypoint, blocking and non-blocking points mixed, a share of
points out of view / bad / with windows beside the image, some slots taken on entry."""
import os

import numpy as np

from conftest import GOLDEN

BOUNDS = (0.0, 0.0, 752.0, 480.0)


def frame(oracle):
    kl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_kp.bin"), oracle.KEYPOINT_DTYPE)
    dl = np.fromfile(os.path.join(GOLDEN, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    st = np.load(os.path.join(GOLDEN, "EuRoC_stereo.npz"))
    return kl, dl, st["u_right"], st["depth"] if "depth" in st.files else None


def contended_map(oracle, kl, dl, seed, m, sigma=3.0, nflip=6):
    rng = np.random.default_rng(seed)
    n = len(kl)
    mps = np.zeros(m, oracle.MAP_POINT_DTYPE)
    src = rng.integers(0, n, m)
    mpd = dl[src].copy()
    flips = rng.integers(0, 256, (m, nflip))
    for j in range(nflip):
        sel = rng.random(m) < 0.5
        mpd[sel, flips[sel, j] >> 3] ^= (1 << (flips[sel, j] & 7)).astype(np.uint8)
    far = rng.random(m) < 0.1                          # a tenth imitates nothing: candidates in the window, none within TH_HIGH
    mpd[far] = rng.integers(0, 256, (int(far.sum()), 32), dtype=np.uint8)
    mps["proj_x"] = kl["x"][src] + rng.normal(0, sigma, m)
    mps["proj_y"] = kl["y"][src] + rng.normal(0, sigma, m)
    off = rng.random(m) < 0.05                         # windows beside every keypoint: GetFeaturesInArea returns nothing
    mps["proj_x"][off] += 5000
    mps["proj_xr"] = mps["proj_x"] - rng.uniform(0, 30, m)
    mps["level"] = np.clip(kl["octave"][src] + rng.integers(-1, 2, m), 0, 7)
    mps["view_cos"] = rng.choice([1.0, 0.9985, 0.99], m)
    fl = np.full(m, 1 | 4, np.int32)
    fl[rng.random(m) < 0.05] = 4
    fl[rng.random(m) < 0.05] |= 2
    fl[rng.random(m) < 0.3] &= ~4
    mps["flags"] = fl
    taken = (rng.random(n) < 0.2).astype(np.uint8)
    return mps, mpd, taken


def clock_cut(out_point, k):
    """Index one past the point at which SearchByProjection_Budget's k-th clock reading happens (src/ORBmatcher.cc:96-102): the clock is
    read at the end of the loop body, which a point reaches when it matched or found nothing within TH_HIGH (-3)."""
    reads = np.flatnonzero((out_point >= 0) | (out_point == -3))
    return len(out_point) if k <= 0 or len(reads) < k else int(reads[k - 1]) + 1


def prefix_state(out_point, prefix, n):
    out_mp = np.full(n, -1, np.int32)
    out_sc = np.zeros(n, np.int32)
    cnt = 0
    for p in range(prefix):
        v = int(out_point[p])
        if v >= 0:
            out_mp[v & 0xFFFF] = p
            out_sc[v & 0xFFFF] = v >> 16
            cnt += 1
    return cnt, out_mp, out_sc
