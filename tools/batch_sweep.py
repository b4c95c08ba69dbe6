"""batch-size sweep 1 / 16 / 64 / 256 (plus 2, 128) of the headline workload on one MI355X:
frames/s with overlapping contexts, and the device time of every stage per image (HIP events, one context).
Writes gpurun_out/batch_sweep.json (copied to profiles/batch_sweep_<round>.json)."""
import json
import os
import sys

sys.path.insert(0, ".")
import torch

import bench
import gf_orb_slam2_amd as G

workload = sys.argv[1] if len(sys.argv) > 1 else "stereo752"
rows = []
for B in (2, 16, 64, 128, 256):
    job = bench.Job(G, torch, workload, B, bench.CONTEXTS[workload], 0, 0, 1, None, n_inputs=2)   # chained as bench.py runs them
    steps = max(20, min(400, 25600 // B))
    dt = job.timed(steps, 15)
    prof = job.profile(5)
    stage_us_per_img = {k: round(v[0] / 5 * 1e3 / B, 3) for k, v in prof.items() if v[1] > 0}
    rows.append({"images_per_step": B, "steps": steps, "frames_per_s": round(B * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4),
                 "stage_us_per_image": stage_us_per_img, "sum_us_per_image": round(sum(stage_us_per_img.values()), 3)})
    print(rows[-1], flush=True)
    job.close()
    del job
    torch.cuda.empty_cache()
# batch of ONE frame (one stereo pair = 2 images is the smallest stereo step; a single image for the extract workloads)
out = {"workload": workload, "contexts": bench.CONTEXTS[workload], "chain_stage": bench.CHAIN_STAGE.get(workload, 0), "rows": rows,
       "note": "frames/s: the contexts alternate steps as in bench.py; stage times: one context alone, HIP events per kernel"}
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/batch_sweep_%s.json" % workload, "w"), indent=1)
