"""Extraction throughput on the reference's own EuRoC test images (tests/golden/EuRoC_{l,r}_752x480.u8) instead of
the synthetic stream: real indoor imagery sends about half of the FAST cells into the second (minThFAST) round, which
the synthetic frames hardly exercise.  Prints frames/s and the per-kernel device times."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import torch

import gf_orb_slam2_amd as G

B = int(os.environ.get("GFO_REAL_BATCH", "256"))   # as bench.py's default batch
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
value, ms_step, mean_kp = B * steps / dt, dt / steps * 1e3, float(exts[0][0].batch_counts(B).mean())
print(f"EuRoC images, stereo752 pipeline: {value:.0f} frames/s, {ms_step:.3f} ms per {B} images; mean keypoints {mean_kp:.0f}")
e = exts[0][0]
for _ in range(30):
    e.extract_batch_device(d.data_ptr(), B, 752, 480)
e.profile_enable(True)
for _ in range(10):
    e.extract_batch_device(d.data_ptr(), B, 752, 480)
torch.cuda.synchronize()
stage_us = {k: round(v[0] / 10 * 1e3, 1) for k, v in e.profile_read().items() if v[1]}
print(stage_us)
# last line: what bench.py folds into its own line as `real_image`
print(json.dumps({"workload": "stereo752 pipeline on the reference's EuRoC test pair (tests/golden/EuRoC_{l,r}_752x480.u8, 64 shifted copies of the pair per step)",
                  "value": round(value, 1), "unit": "frames/s", "ms_per_step": round(ms_step, 4), "images_per_step": B, "contexts": NCTX,
                  "mean_keypoints_per_image": round(mean_kp, 1), "stage_us_per_step_one_context": stage_us, "k_fast_us": stage_us.get("fast"),
                  "note": "about half of the FAST cells of real indoor imagery find no corner at iniThFAST and repeat the cascade at minThFAST "
                          "(ORBextractor.cc:811-818); the synthetic stream of the headline hardly exercises that"}))
