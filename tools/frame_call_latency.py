"""Per-call latency of ORBextractor::operator() on ONE image (host buffers in and out): python tools/frame_call_latency.py [W H NFEATURES]
(default 1920 1080 4000).  With tools/trace_cmd_tail.sh the kernel timeline of the last call."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_frame

w, h, nf = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 4000)
ext = G.ORBextractor(nf, 1.2, 8, 20, 7)
img = synth_frame(w, h, 3)
for _ in range(5):
    kp, desc = ext(img)
t = []
for _ in range(40):
    t0 = time.perf_counter()
    kp, desc = ext(img)
    t.append(time.perf_counter() - t0)
print(f"{w}x{h} @{nf}: {len(kp)} keypoints, median {np.median(t) * 1e3:.3f} ms per call (min {min(t) * 1e3:.3f})", flush=True)
ext.close()
