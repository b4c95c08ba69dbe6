import sys
sys.path.insert(0, ".")
import numpy as np
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_stereo_pair
l, r = synth_stereo_pair(752, 480, 3)
ext = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=2)
for _ in range(3):
    ext.extract_batch(np.stack([l, r]))
