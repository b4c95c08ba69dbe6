"""Which form of the batch blur wins at which image width (round 6): the bench's own Job (resident batches, two contexts, chained) on
workloads of several widths, GFO_BLUR_MFMA=0 / 1 alternating in ONE process (the library reads the variable per launch).
usage (through gpurun): python tools/ab_blur_width.py [w:h:nfeatures:batch ...]"""
import os
import sys

sys.path.insert(0, ".")
import bench

import torch
import gf_orb_slam2_amd as G

shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(752, 480, 2000, 256), (1024, 768, 2500, 128), (1241, 376, 2000, 192), (1280, 720, 3000, 96),
                                                                       (1600, 900, 3500, 64), (1920, 1080, 4000, 64)]
for w, h, nf, B in shapes:
    name = f"extract{w}x{h}"
    bench.WORKLOADS[name] = (w, h, nf, None, name)
    bench.CONTEXTS[name] = 2
    bench.CHAIN_STAGE[name] = 0
    job = bench.Job(G, torch, name, B, 2, 0, 0, 1, None)
    out = []
    for rep in range(2):
        for form in ("0", "1"):
            os.environ["GFO_BLUR_MFMA"] = form
            t = job.timed(40, 10, barrier=False)
            out.append((form, B / (t / 40) ))
    print(f"{w}x{h} @{nf}, batch {B}: " + "  ".join(f"{'matrix' if f == '1' else 'stream'} {v:9.0f}" for f, v in out) + " frames/s", flush=True)
    del job
    torch.cuda.empty_cache()
