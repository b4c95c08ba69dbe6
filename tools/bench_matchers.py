"""Config 4 of BASELINE.json: one 1920x1080 frame @4000 features, SearchByProjection against a 50 000-point synthetic
local map (input S3).  Times the extraction (single frame, host buffers), the projection search on the
GPU (host arrays in and out) and the oracle's on one host core.  Not the headline metric: reported in DESIGN.md 6."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_frame
from oracle import orb_oracle as O

O.build()
ext = G.ORBextractor(4000, 1.2, 8, 20, 7)
img = synth_frame(1920, 1080, 3)
kp, desc = ext(img)
rng = np.random.default_rng(7)
n, m = len(kp), 50000
mps = np.zeros(m, O.MAP_POINT_DTYPE)
mpd = rng.integers(0, 256, (m, 32), dtype=np.uint8)
nv = min(n, 3500)
vis = rng.choice(m, nv, replace=False)
src = rng.choice(n, nv, replace=False)
d = desc[src].copy()
for j in range(60):
    sel = rng.random(nv) < rng.random(nv)
    bits = rng.integers(0, 256, nv)
    d[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
mpd[vis] = d
mps["proj_x"] = rng.uniform(0, 1920, m); mps["proj_y"] = rng.uniform(0, 1080, m)
mps["proj_x"][vis] = kp["x"][src] + rng.normal(0, 2, nv)
mps["proj_y"][vis] = kp["y"][src] + rng.normal(0, 2, nv)
mps["level"] = rng.integers(0, 8, m); mps["level"][vis] = kp["octave"][src]
mps["proj_xr"] = mps["proj_x"] - 10
mps["view_cos"] = 1.0
mps["flags"] = 5
bounds = (0.0, 0.0, 1920.0, 1080.0)
sf = ext.GetScaleFactors()
matcher = G.ORBmatcher(0.8, True, extractor=ext)


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps * 1e3, r


t_ext, _ = timeit(lambda: ext(img), 30)
t_gpu, got = timeit(lambda: matcher.SearchByProjection(kp, desc, None, sf, bounds, mps, mpd, 3.0, None), 30)
t_cpu, ref = timeit(lambda: O.search_by_projection(kp, desc, None, sf, bounds, mps, mpd, 3.0, 0.8, None), 3)
assert got[0] == ref[0] and (got[1] == ref[1]).all()
t_ocpu, _ = timeit(lambda: O.OracleExtractor(4000, 1.2, 8, 20, 7)(img), 3)
print(f"1920x1080 @4000: {n} keypoints, {m} map points, {got[0]} matches")
print(f"extract, one frame, host buffers: GPU {t_ext:.3f} ms   oracle (1 core) {t_ocpu:.1f} ms")
print(f"SearchByProjection, host arrays:  GPU {t_gpu:.3f} ms   oracle (1 core) {t_cpu:.1f} ms")
print(f"frame total: GPU {t_ext + t_gpu:.3f} ms = {1e3 / (t_ext + t_gpu):.0f} frames/s through the host-buffer calls; oracle {t_ocpu + t_cpu:.1f} ms")
