"""Extraction throughput on the reference's own EuRoC test images (tests/golden/EuRoC_{l,r}_752x480.u8) instead of
the synthetic stream: real indoor imagery sends about half of the FAST cells into the second (minThFAST) round, which
the synthetic frames hardly exercise.  Prints frames/s and the per-kernel device times."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch

import gf_orb_slam2_amd as G

B = 128
l = np.fromfile("tests/golden/EuRoC_l_752x480.u8", np.uint8).reshape(480, 752)
r = np.fromfile("tests/golden/EuRoC_r_752x480.u8", np.uint8).reshape(480, 752)
# 64 "pairs": the two images under small integer shifts (np.roll), so the batch is not 64 identical copies
frames = []
for k in range(B // 2):
    frames += [np.roll(l, (3 * k) % 41, axis=1), np.roll(r, (3 * k) % 41, axis=1)]
d = torch.from_numpy(np.stack(frames)).cuda()
exts = []
NCTX = 2          # as bench.py runs stereo752: two contexts chained behind each other's pyramid
for k in range(NCTX):
    st = torch.cuda.Stream()
    e = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=B)
    e.set_stream(st.cuda_stream)
    e.extract_batch_device(d.data_ptr(), B, 752, 480)
    exts.append((e, st, G.ORBmatcher(0.8, True, extractor=e)))
for k in range(NCTX):
    exts[k][0].chain_after(exts[(k - 1) % NCTX][0], 1)
sp = G.StereoParams(480, 47.90639384423901, 47.90639384423901 / 435.2046959714599, 0.0)
torch.cuda.synchronize()
for steps in (30, 300):
    t0 = time.perf_counter()
    for i in range(steps):
        e, st, m = exts[i % NCTX]
        e.extract_batch_device(d.data_ptr(), B, 752, 480)
        m.stereo_match_batch(sp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"EuRoC images, stereo752 pipeline: {B * steps / dt:.0f} frames/s, {dt / steps * 1e3:.3f} ms per {B} images; "
      f"mean keypoints {exts[0][0].batch_counts(B).mean():.0f}")
e = exts[0][0]
for _ in range(30):
    e.extract_batch_device(d.data_ptr(), B, 752, 480)
e.profile_enable(True)
for _ in range(10):
    e.extract_batch_device(d.data_ptr(), B, 752, 480)
torch.cuda.synchronize()
print({k: round(v[0] / 10 * 1e3, 1) for k, v in e.profile_read().items() if v[1]})
