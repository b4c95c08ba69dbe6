"""Diagnostic of the batched projection search on the proj1080 stream: per-frame live points, fixed-point rounds,
matches; wall time of the chain alone (inputs resident).  Run on the GPU box."""
import ctypes, sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_local_map, synth_stream

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
w, h, M = 1920, 1080, 50000
frames, offs = synth_stream(w, h, B, idx=3)
ext = G.ORBextractor(4000, 1.2, 8, 20, 7, max_batch=B)
d = torch.from_numpy(np.stack(frames)).cuda()
ext.extract_batch_device(d.data_ptr(), B, w, h)
kp0, d0 = ext.batch_fetch(0)
mpd, mps = synth_local_map(kp0, d0, offs, w, h, M, 4000)
m = G.ORBmatcher(0.8, True, extractor=ext)
m.map_upload(mpd)
d_mps = torch.from_numpy(mps.view(np.uint8).reshape(B, -1)).cuda()
b = (0.0, 0.0, float(w), float(h))
m.search_by_projection_batch(d_mps.data_ptr(), b, device_ptrs=True)
ext.synchronize()
L = G.load_library()
p_mp, p_sc, p_ct, st, cs = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
L.gfo_projection_device_views(ext.handle, ctypes.byref(p_mp), ctypes.byref(p_sc), ctypes.byref(p_ct), ctypes.byref(st), ctypes.byref(cs))


class DA:
    def __init__(s, p, n): s.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (p, False), "version": 2}


cnt = torch.as_tensor(DA(p_ct.value, B * cs.value), device="cuda").cpu().numpy().reshape(B, cs.value)
print("nlive", cnt[:, 0].tolist())
print("rounds", cnt[:, 1].tolist())
print("nmatch", cnt[:, 2].tolist())
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        m.search_by_projection_batch(d_mps.data_ptr(), b, device_ptrs=True)
    ext.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"projection chain alone: {dt * 1e3:.3f} ms per batch of {B} = {dt / B * 1e6:.1f} us/frame")
