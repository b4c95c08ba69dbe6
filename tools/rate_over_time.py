"""Throughput of the resident pipeline in consecutive chunks of steps (does the rate drift after start-up?).
usage (through gpurun): python tools/rate_over_time.py [contexts] [chunks] [steps per chunk]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gf_orb_slam2_amd as G

nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 3
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 20
per = int(sys.argv[3]) if len(sys.argv) > 3 else 200
torch.cuda.set_device(0)
job = bench.Job(G, torch, "stereo752", 128, nctx, 0, 0, 1, None)
for _ in range(30):
    job.step()
torch.cuda.synchronize()
rates = []
for c in range(chunks):
    t0 = time.perf_counter()
    for _ in range(per):
        job.step()
    if os.environ.get("SYNC_EACH_CHUNK", "1") == "1":
        torch.cuda.synchronize()
    rates.append(128 * per / (time.perf_counter() - t0))
torch.cuda.synchronize()
print(f"contexts {nctx}:", " ".join(f"{r / 1e3:.0f}" for r in rates), flush=True)
job.close()
