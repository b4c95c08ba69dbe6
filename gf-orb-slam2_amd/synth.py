"""Synthetic inputs of SURVEY.md 8d (S2 frames, S3 local map) -- shared by tests and bench.py."""
import numpy as np


def synth_frame(w, h, idx=0):
    """Three octaves of value noise + 400 random dark/bright rectangles (side 6..40 px), seeded
    numpy.random.default_rng(20260403 + idx); fills every pyramid level's FAST quota."""
    rng = np.random.default_rng(20260403 + idx)
    img = np.zeros((h, w), np.float32)
    for o, amp in ((64, 60.0), (32, 30.0), (16, 15.0)):
        gh, gw = h // o + 2, w // o + 2
        g = rng.random((gh, gw), dtype=np.float32)
        ys = np.arange(h, dtype=np.float32) / o
        xs = np.arange(w, dtype=np.float32) / o
        y0 = ys.astype(np.int32)
        x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        img += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    img += 70.0
    for _ in range(400):
        rw, rh = rng.integers(6, 41, 2)
        rw, rh = min(int(rw), w - 1), min(int(rh), h - 1)   # tiny test images: keep the rectangle inside
        x = rng.integers(0, w - rw)
        y = rng.integers(0, h - rh)
        img[y:y + rh, x:x + rw] += rng.choice([-1.0, 1.0]) * rng.uniform(25, 90)
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_stereo_pair(w, h, idx=0):
    """Left = synth_frame; right = the same scene seen with a piecewise-constant disparity
    (three horizontal bands shifted by 6 / 14 / 27 px) plus +-2 grey levels of sensor noise."""
    left = synth_frame(w, h, idx)
    rng = np.random.default_rng(77 + idx)
    right = np.empty_like(left)
    bands = [(0, h // 3, 6), (h // 3, 2 * h // 3, 14), (2 * h // 3, h, 27)]
    for y0, y1, d in bands:
        right[y0:y1] = np.roll(left[y0:y1], -d, axis=1)
    noise = rng.integers(-2, 3, right.shape, dtype=np.int16)
    right = np.clip(right.astype(np.int16) + noise, 0, 255).astype(np.uint8)
    return left, right


def synth_stream(w, h, nframes, idx=0, max_shift=24):
    """`nframes` views of ONE synthetic scene, as a camera moving over it sees them: crops of a
    (w + 2 max_shift) x (h + 2 max_shift) synth_frame at per-frame integer offsets, each with its own +-2 grey
    levels of sensor noise.  Frame 0 sits at the centre offset.  Returns (frames, offsets[(ox, oy)])."""
    ms = int(max_shift)
    base = synth_frame(w + 2 * ms, h + 2 * ms, idx).astype(np.int16)
    rng = np.random.default_rng(991 + idx)
    offs = [(ms, ms)] + [(int(rng.integers(0, 2 * ms + 1)), int(rng.integers(0, 2 * ms + 1))) for _ in range(nframes - 1)]
    frames = []
    for ox, oy in offs:
        noise = rng.integers(-2, 3, (h, w), dtype=np.int16)
        frames.append(np.clip(base[oy:oy + h, ox:ox + w] + noise, 0, 255).astype(np.uint8))
    return frames, offs


MAP_POINT_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"),
                            ("view_cos", "<f4"), ("level", "<i4"), ("flags", "<i4")])


def synth_local_map(kp0, desc0, offs, w, h, m=50000, n_vis=4000, seed=7, nlevels=8):
    """SURVEY.md 8d input S3 for a stream: a local map of `m` descriptors -- a visible subset that imitates frame
    0's keypoints (their descriptors with k ~ U{0..60} random bits flipped), uniform random 256-bit strings for the
    rest -- and its projection into every frame: matching keypoint position (moved by the frame's offset against
    frame 0) + N(0, 2 px), levels copied, viewCos = 1, mbTrackInView and Observations() > 0 everywhere; the other
    points project uniformly over the image at random levels.  Returns (mp_desc [m, 32], mps [nframes, m])."""
    rng = np.random.default_rng(seed)
    n = len(kp0)
    nv = min(n, n_vis, m)
    mpd = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    vis = rng.choice(m, nv, replace=False)
    src = rng.choice(n, nv, replace=False)
    d = desc0[src].copy()
    kflip = rng.integers(0, 61, nv)
    for j in range(60):
        sel = j < kflip
        bits = rng.integers(0, 256, nv)
        d[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
    mpd[vis] = d
    level_rand = rng.integers(0, nlevels, m)
    mps = np.zeros((len(offs), m), MAP_POINT_DTYPE)
    ox0, oy0 = offs[0]
    for f, (ox, oy) in enumerate(offs):
        p = mps[f]
        p["proj_x"] = rng.uniform(0, w, m)
        p["proj_y"] = rng.uniform(0, h, m)
        p["proj_x"][vis] = kp0["x"][src] - (ox - ox0) + rng.normal(0, 2, nv)
        p["proj_y"][vis] = kp0["y"][src] - (oy - oy0) + rng.normal(0, 2, nv)
        p["level"] = level_rand
        p["level"][vis] = kp0["octave"][src]
        p["proj_xr"] = p["proj_x"] - 10
        p["view_cos"] = 1.0
        p["flags"] = 5
    return mpd, mps


def synth_vocabulary(k=10, depth=4, seed=0):
    """A full k-ary vocabulary tree of the given depth in the flattened form ORBVocabulary takes (breadth first, the children of a node
    contiguous, leaves = words numbered in tree order): node descriptors are their parent's with 128 >> level random bits flipped, word
    weights the logarithms DBoW2 stores (double) -- the SHAPE of ORBvoc (k = 10, L = 6; that file is not in the reference repository),
    not its content.  For bench.py and tools/matcher_call_latency.py; the parity tests use the oracle's ragged generator."""
    rng = np.random.default_rng(seed)
    first, nch, desc = [], [], [np.zeros((1, 32), np.uint8)]
    level_start, n_level = 0, 1
    for lvl in range(depth + 1):
        if lvl < depth:
            first.append(level_start + n_level + k * np.arange(n_level, dtype=np.int32))
            nch.append(np.full(n_level, k, np.int32))
            child = np.repeat(desc[-1], k, axis=0)
            nflip = max(1, 128 >> lvl)
            bits = rng.integers(0, 256, (len(child), nflip))
            for j in range(nflip):
                child[np.arange(len(child)), bits[:, j] >> 3] ^= (1 << (bits[:, j] & 7)).astype(np.uint8)
            desc.append(child)
        else:
            first.append(np.zeros(n_level, np.int32))
            nch.append(np.zeros(n_level, np.int32))
        level_start += n_level
        n_level *= k
    first, nch, desc = np.concatenate(first), np.concatenate(nch), np.concatenate(desc)
    n = len(first)
    leaves = np.nonzero(nch == 0)[0]
    word = np.full(n, -1, np.int32)
    word[leaves] = np.arange(len(leaves), dtype=np.int32)
    w64 = np.zeros(n, np.float64)
    w64[leaves] = np.log(rng.uniform(1.1, 400.0, len(leaves)))
    return {"first_child": first, "n_children": nch, "descriptors": desc, "word_id": word, "weight": w64.astype(np.float32), "weight64": w64,
            "depth": depth}
