"""Latency of ONE stereo pair through the host-buffer entry points (what the C++ adapter calls per frame):
H2D of two 752x480 images, extraction of both, stereo association, D2H of keypoints / descriptors / matches."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_stereo_pair

l, r = synth_stereo_pair(752, 480, 3)
ext = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=2)
m = G.ORBmatcher(0.8, True, extractor=ext)
sp = G.StereoParams(480, 47.90639384423901, 47.90639384423901 / 435.2046959714599, 0.0)
imgs = np.stack([l, r])


def once():
    out = ext.extract_batch(imgs)
    m.stereo_match_batch(sp)
    return out, m.stereo_fetch(0, len(out[0][0]))


for _ in range(5):
    once()
ts = []
for _ in range(50):
    t0 = time.perf_counter()
    once()
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print(f"one stereo pair, host buffers in and out: median {np.median(ts):.3f} ms, min {ts.min():.3f} ms, p90 {np.percentile(ts, 90):.3f} ms")

# where the time goes: device time of each stage for this 2-image batch (HIP events), the rest is host-side
ext.profile_enable(True)
for _ in range(20):
    once()
prof = ext.profile_read()
ext.profile_enable(False)
print("device us per pair:", {k: round(ms / 20 * 1e3, 1) for k, (ms, n) in prof.items()}, "sum", round(sum(ms for ms, n in prof.values()) / 20 * 1e3, 1))
for name, fn in (("extract_batch", lambda: ext.extract_batch(imgs)), ("stereo_match_batch", lambda: m.stereo_match_batch(sp)),
                 ("stereo_fetch", lambda: m.stereo_fetch(0, 2064))):
    t0 = time.perf_counter()
    for _ in range(50):
        fn()
    print(f"{name}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
