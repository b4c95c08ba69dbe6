"""Phase timestamps of one quadtree workgroup (level 0 of image 0) for a batch of B images: GFO_QT_TIMING=1 python tools/qt_timing.py [B]"""
import os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ["GFO_QT_TIMING"] = "1"
import torch
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_stereo_pair
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
fr = []
for p in range(B // 2):
    l, r = synth_stereo_pair(752, 480, p)
    fr += [l, r]
d = torch.from_numpy(np.stack(fr)).cuda()
e = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=B)
for _ in range(3):
    e.extract_batch_device(d.data_ptr(), B, 752, 480)
    e.synchronize()
