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
