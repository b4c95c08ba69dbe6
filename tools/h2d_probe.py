"""Where does the PCIe-inclusive rate go?  Times the 46-MB batch copy alone, then while the extraction pipeline runs on
other streams (copy start/stop events on the copy stream), with one and with two copy streams.
usage (through gpurun): python tools/h2d_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

torch.cuda.set_device(0)
import gf_orb_slam2_amd as G
B = 128
job = bench.Job(G, torch, "stereo752", B, 3, 0, 0, 1, None)
pinned = torch.from_numpy(job.host_batches[0]).pin_memory()
nbytes = pinned.numel()
dst = [torch.empty_like(job.d_inputs[0]) for _ in range(2)]
cs = [torch.cuda.Stream(), torch.cuda.Stream()]


def copies(n, streams, chunks, with_compute):
    evs = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        if with_compute:
            job.step()
        s = cs[i % streams]
        with torch.cuda.stream(s):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s)
            step = nbytes // chunks
            flat_d, flat_s = dst[i % 2].view(-1), pinned.view(-1)
            for c in range(chunks):
                flat_d[c * step:(c + 1) * step].copy_(flat_s[c * step:(c + 1) * step], non_blocking=True)
            b.record(s)
            evs.append((a, b))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in evs]
    return float(np.median(ms)), wall / n * 1e3


def instream(n):
    """the batch copy on the context's OWN stream, in front of its kernels (no copy stream, no cross-stream event)"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = job.step_no % job.nctx
        d_in = job.d_inputs[job.step_no % len(job.d_inputs)]
        with torch.cuda.stream(job.streams[k]):
            d_in.copy_(pinned, non_blocking=True)
        job.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(20):
    job.step()
for n in (30, 100, 300):
    dth = job.timed(n, 5, h2d_from=pinned)
    print(f"bench.py step(h2d_from), {n} steps: {dth / n * 1e3:.3f} ms per step = {B * n / dth:.0f} frames/s", flush=True)
w = instream(60)
print(f"copy on the context's own stream: {w:.3f} ms per iteration = {B / w * 1e3:.0f} frames/s", flush=True)
for streams, chunks, comp in ((1, 1, False), (1, 1, True), (2, 1, True), (1, 4, True), (2, 2, True)):
    med, wall = copies(40, streams, chunks, comp)
    print(f"copy streams {streams}, chunks {chunks}, pipeline {'running' if comp else 'idle'}: copy {med:.3f} ms = {nbytes / med / 1e6:.1f} GB/s; "
          f"{wall:.3f} ms per iteration = {B / wall * 1e3:.0f} frames/s", flush=True)
job.close()
