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


def two_view_geometry(rng, fx=458.0, fy=457.0, cx=367.0, cy=248.0, baseline=0.3, rot_deg=4.0, forward=False):
    """A second camera a short way from the first: (R21, t21, K, F12 as ORB-SLAM's ComputeF12 lays it out: x1' F12 x2 = 0, the epipole
    of camera 1 in image 2).  float32 like the cv::Mat it stands for."""
    w = rng.normal(0, np.deg2rad(rot_deg), 3)
    th = np.linalg.norm(w)
    k = w / th if th > 0 else np.array([1.0, 0, 0])
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R21 = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t21 = rng.normal(0, baseline, 3)
    if forward:                                                    # motion along the optical axis: the epipole falls inside the image
        t21 = np.array([rng.normal(0, 0.03), rng.normal(0, 0.03), -0.5 - abs(rng.normal(0, baseline))])
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    R12, t12 = R21.T, -R21.T @ t21
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    F12 = np.linalg.inv(K).T @ tx @ R12 @ np.linalg.inv(K)
    c1_in_2 = t21                                                  # camera 1's centre in camera 2's frame
    ex, ey = fx * c1_in_2[0] / c1_in_2[2] + cx, fy * c1_in_2[1] / c1_in_2[2] + cy
    return R21, t21, K, F12.astype(np.float32), np.float32(ex), np.float32(ey)


def triangulation_case(oracle, kp1, desc1, rng, flips=6, noise=0.6, p_mp=0.3, p_stereo=0.5, node_shift=3, p_outlier=0.1, **cam):
    """The second keyframe of SearchForTriangulation from the first: every keypoint seen again at a random depth under a random
    relative pose, its pixel disturbed (some far off the epipolar line), descriptor bits flipped, the order shuffled; map-point and
    stereo flags on both sides; a toy vocabulary (the node of a descriptor = its leading bits)."""
    n = len(kp1)
    R21, t21, K, F12, ex, ey = two_view_geometry(rng, **cam)
    z = rng.uniform(1.0, 12.0, n)
    X1 = np.stack([(kp1["x"] - K[0, 2]) / K[0, 0] * z, (kp1["y"] - K[1, 2]) / K[1, 1] * z, z], 1)
    X2 = X1 @ R21.T + t21
    u2 = K[0, 0] * X2[:, 0] / X2[:, 2] + K[0, 2] + rng.normal(0, noise, n)
    v2 = K[1, 1] * X2[:, 1] / X2[:, 2] + K[1, 2] + rng.normal(0, noise, n)
    out = rng.random(n) < p_outlier
    v2[out] += rng.normal(0, 8.0, int(out.sum()))
    perm = rng.permutation(n)
    kp2 = kp1[perm].copy()
    kp2["x"], kp2["y"] = u2[perm].astype(np.float32), v2[perm].astype(np.float32)
    kp2["angle"] = ((kp1["angle"][perm] + rng.normal(0, 6.0, n)) % 360).astype(np.float32)
    d2 = desc1[perm].copy()
    for _ in range(flips):
        sel = rng.random(n) < 0.5
        bits = rng.integers(0, 256, n)
        d2[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    has1 = (rng.random(n) < p_mp).astype(np.uint8)
    has2 = (rng.random(n) < p_mp).astype(np.uint8)
    ur1 = np.where(rng.random(n) < p_stereo, kp1["x"] - rng.uniform(1, 40, n), -1).astype(np.float32)
    ur2 = np.where(rng.random(n) < p_stereo, kp2["x"] - rng.uniform(1, 40, n), -1).astype(np.float32)
    node1 = (desc1[:, 0] >> node_shift).astype(np.int64)
    node2 = (d2[:, 0] >> node_shift).astype(np.int64)
    node1[rng.random(n) < 0.02] = -1
    node2[rng.random(n) < 0.02] = -1
    return dict(kp1=kp1, desc1=desc1, has1=has1, ur1=ur1, fv1=oracle.make_feature_vector(node1), kp2=kp2, desc2=d2, has2=has2, ur2=ur2,
                fv2=oracle.make_feature_vector(node2), f12=F12, ex=ex, ey=ey, node1=node1, node2=node2, R21=R21, t21=t21)


def initialization_case(oracle, kp1, desc1, rng, flips=8, sigma=15.0, resample=True):
    """The second frame of SearchForInitialization: F1's keypoints drawn WITH replacement (some are seen twice, some not at all: several
    keypoints of F1 then want one keypoint of F2 and the closer one takes it), moved by N(0, sigma) px, bits flipped, shuffled."""
    n = len(kp1)
    src = rng.integers(0, n, n) if resample else rng.permutation(n)
    kp2 = kp1[src].copy()
    kp2["x"] = (kp2["x"] + rng.normal(0, sigma, n)).astype(np.float32)
    kp2["y"] = (kp2["y"] + rng.normal(0, sigma, n)).astype(np.float32)
    kp2["angle"] = ((kp2["angle"] + rng.normal(0, 5.0, n)) % 360).astype(np.float32)
    d2 = desc1[src].copy()
    for _ in range(flips):
        sel = rng.random(n) < 0.5
        bits = rng.integers(0, 256, n)
        d2[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    prev = np.stack([kp1["x"], kp1["y"]], 1).astype(np.float32).copy()      # Tracking.cc:1300-1302: where the keypoint is in F1
    return kp2, d2, prev
