import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_frame
e = G.ORBextractor(2000, 1.2, 8, 20, 7)
for name, img in (("synth", synth_frame(752, 480, 0)), ("euroc_l", np.fromfile("tests/golden/EuRoC_l_752x480.u8", np.uint8).reshape(480, 752))):
    print(name, file=sys.stderr)
    kp, d = e(img)
    cands = sum(len(e.debug_level_candidates(l)) for l in range(8))
    print(name, "keypoints", len(kp), "candidates", cands, file=sys.stderr)
