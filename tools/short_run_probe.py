"""Where do the first steps of a short timed run go?  Completion time of every step of a K-step run (event per step)
after W warm-up steps and a synchronisation, as bench.py times it.   usage: python tools/short_run_probe.py [K] [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import gf_orb_slam2_amd as G

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.cuda.set_device(0)
job = bench.Job(G, torch, "stereo752", 128, bench.CONTEXTS["stereo752"], 0, 0, 1, None)
for rep in range(3):
    for _ in range(W):
        job.step()
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True)
    start.record(job.streams[job.step_no % job.nctx])
    evs = []
    t0 = time.perf_counter()
    for i in range(K):
        k = job.step_no % job.nctx
        job.step()
        e = torch.cuda.Event(enable_timing=True)
        e.record(job.streams[k])
        evs.append(e)
    host_submit = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    done = [start.elapsed_time(e) for e in evs]
    gaps = [done[0]] + [done[i] - done[i - 1] for i in range(1, K)]
    print(f"rep {rep}: wall {wall * 1e3:.2f} ms ({128 * K / wall:.0f} frames/s), host submitted everything after {host_submit * 1e3:.2f} ms; "
          "step completion gaps (ms): " + " ".join(f"{g:.2f}" for g in gaps), flush=True)
job.close()
