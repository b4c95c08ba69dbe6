"""Diagnostic: libgfo first, torch afterwards (the order a Python user gets when torch is imported lazily).
Dumps every thread's stack and exits if the sequence stalls for 90 s."""
import faulthandler
import sys
import time

sys.path.insert(0, ".")
faulthandler.dump_traceback_later(90, exit=True)
t0 = time.time()
import numpy as np
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_frame

img = synth_frame(752, 480, 1)
ext = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=128)
k, d = ext(img)
print("gfo first call ok", len(k), round(time.time() - t0, 2), flush=True)
import torch
print("import torch ok", round(time.time() - t0, 2), flush=True)
t = torch.from_numpy(np.stack([img] * 128)).cuda()
print("torch cuda init ok", round(time.time() - t0, 2), flush=True)
ext.set_stream(torch.cuda.current_stream().cuda_stream)
ext.extract_batch_device(t.data_ptr(), 128, 752, 480)
torch.cuda.synchronize()
n = ext.batch_counts(128)
assert (n == len(k)).all()
ext.set_stream(0)
ext.close()
print("batch ok", round(time.time() - t0, 2), flush=True)
