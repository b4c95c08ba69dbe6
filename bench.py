#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X ORB front-end (extract + stereo match) on synthetic
752x480 grayscale frames at 2000 features, plus the roofline figure of the dominant kernel and
the CPU oracle timed beside it.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one resident batch of `--batch` images per GPU
(left/right interleaved: L0,R0,L1,R1,...): pyramid -> blur -> FAST cells -> quadtree ->
orientation+descriptor for every image, then the stereo Hamming association of every pair.
Inputs are already in HBM when the timed region starts.  Independent frames shard one batch
per GPU (weak scaling); the only collective is the RCCL all-gather of the per-image keypoint
counts.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (w, h, nfeatures, stereo, BASELINE.json config it corresponds to)
    "stereo752": (752, 480, 2000, True, "configs[2]: stereo 752x480 pair, 2000 features/image, extract L+R + stereo Hamming match"),
    "extract752": (752, 480, 2000, False, "configs[1]: 752x480 synthetic stream, 2000 features, extract-only"),
    "extract1080": (1920, 1080, 4000, False, "configs[3] (extract part): 1920x1080, 4000 features"),
}
FX, BF = 435.2046959714599, 47.90639384423901


class _DevArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, n, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def level_sizes(w, h, inv_scale):
    out = []
    for s in inv_scale:
        out.append((int(np.rint(np.float32(w) * np.float32(s))), int(np.rint(np.float32(h) * np.float32(s)))))
    return out


def algorithmic_bytes(w, h, sizes, n_kp_img, stereo):
    """Algorithmic HBM bytes per IMAGE of each stage (DESIGN.md 'bytes per unit')."""
    P = sum(a * b for a, b in sizes)
    px = [a * b for a, b in sizes]
    d = {
        "resize": sum(px[l - 1] + px[l] for l in range(1, len(px))),   # read level l-1, write level l
        "blur": 2 * P,                                                   # read level, write blurred level
        "fast": P,                                                       # read every level once
        "quadtree": 0,
        "orient_desc": n_kp_img * (749 + 1369 + 60),                     # disc + 37x37 window + kp/desc out
        "stereo_match": (2 * n_kp_img * 60 + n_kp_img * 8) / 2 if stereo else 0,  # per image = half a pair
        "stereo_bucket": 0,
        "stereo_cut": 0,
    }
    survey_total = w * h + 4 * P + n_kp_img * 2178 + ((2 * n_kp_img * 60 + n_kp_img * 8) / 2 if stereo else 0)
    return d, survey_total


def cpu_baseline(w, h, nfeatures, stereo, budget_s=10.0):
    """The CPU oracle (oracle/, kind 'port') on a bounded sample of the same synthetic stream, on this host:
    one thread (the headline `value`), two threads (the reference's own left/right arrangement, Frame.cc:84-87)
    and up to 32 threads with one stereo pair per thread.  ctypes releases the GIL, so
    the threads run the C oracle concurrently."""
    from concurrent.futures import ThreadPoolExecutor
    from gf_orb_slam2_amd.synth import synth_stereo_pair
    from oracle import orb_oracle as O
    O.build()

    def make():
        return O.OracleExtractor(nfeatures, 1.2, 8, 20, 7)

    pairs = [synth_stereo_pair(w, h, 1000 + i) for i in range(8)]
    sf = make().scale_factors

    def one_pair(oe_l, oe_r, idx, two_threads=None):
        l, r = pairs[idx % len(pairs)]
        if two_threads is not None:
            fl = two_threads.submit(oe_l, l)
            kr, dr = oe_r(r)
            kl, dl = fl.result()
        else:
            kl, dl = oe_l(l)
            kr, dr = oe_r(r)
        if stereo:
            O.stereo_match(kl, dl, kr, dr, sf, h, BF, BF / FX, 0.0)

    oe_l, oe_r = make(), make()
    t0 = time.perf_counter()
    one_pair(oe_l, oe_r, 0)
    t_first = time.perf_counter() - t0
    n1 = int(max(3, min(60, budget_s / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    for i in range(n1):
        one_pair(oe_l, oe_r, i)
    v1 = 2 * n1 / (time.perf_counter() - t0)
    # two threads: left and right image extracted concurrently
    n2 = max(3, n1 // 2)
    with ThreadPoolExecutor(1) as side:
        t0 = time.perf_counter()
        for i in range(n2):
            one_pair(oe_l, oe_r, i, side)
        v2 = 2 * n2 / (time.perf_counter() - t0)
    # all cores (capped at 32 threads): one pair per thread
    nt = max(1, min(32, os.cpu_count() or 1))
    exts = [(make(), make()) for _ in range(nt)]
    per = max(2, n1 // 4)

    def worker(k):
        for i in range(per):
            one_pair(exts[k][0], exts[k][1], k * per + i)
    with ThreadPoolExecutor(nt) as pool:
        t0 = time.perf_counter()
        list(pool.map(worker, range(nt)))
        vn = 2 * per * nt / (time.perf_counter() - t0)
    return {"value": round(v1, 2), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n1} stereo pairs ({2 * n1} images) of the same {w}x{h} synthetic stream, oracle/orb_oracle.c, "
                      f"single thread; {os.cpu_count()} host cores present",
            "threads_2": round(v2, 2), "threads_n": {"threads": nt, "value": round(vn, 2)},
            "reference_published": "13.7-22.2 ms per stereo frame on unstated hardware (README.md:7-16) = 90-146 images/s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="images per GPU per step (even)")
    ap.add_argument("--workload", default="stereo752", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--streams", type=int, default=3,
                    help="independent contexts (arena + HIP stream) the steps alternate between, so the tail of one "
                         "batch overlaps the head of the next")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.sharding import gather_counts, shard_pairs
    from gf_orb_slam2_amd.synth import synth_stereo_pair

    w, h, nfeat, stereo, cfg_name = WORKLOADS[args.workload]
    B = args.batch - (args.batch & 1)
    # distinct synthetic pairs per rank (frame index continues across ranks: independent streams)
    frames = []
    for p in shard_pairs(rank, world, B // 2):
        l, r = synth_stereo_pair(w, h, p)
        frames += [l, r]
    d_imgs = torch.from_numpy(np.stack(frames)).cuda()

    nctx = max(1, args.streams)
    L = G.load_library()
    sp = G.StereoParams(h, BF, BF / FX, 0.0)
    exts, matchers, streams, counts_ts, gathered = [], [], [], [], []
    for k in range(nctx):
        st = torch.cuda.Stream()
        e = G.ORBextractor(nfeat, 1.2, 8, 20, 7, device=local_rank, max_batch=B)
        e.set_stream(st.cuda_stream)
        # first pass plans the arena; then bind the device-side count vector for the collective
        e.extract_batch_device(d_imgs.data_ptr(), B, w, h)
        p_kp, p_desc, p_cnt, stride = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int()
        L.gfo_batch_device_views(e.handle, ctypes.byref(p_kp), ctypes.byref(p_desc), ctypes.byref(p_cnt), ctypes.byref(stride))
        exts.append(e)
        matchers.append(G.ORBmatcher(0.8, True, extractor=e))
        streams.append(st)
        counts_ts.append(torch.as_tensor(_DevArray(p_cnt.value, B), device="cuda"))
        gathered.append(torch.zeros(world * B, dtype=torch.int32, device="cuda") if world > 1 else None)
    ext = exts[0]
    step_no = [0]

    def step():
        k = step_no[0] % nctx
        step_no[0] += 1
        exts[k].extract_batch_device(d_imgs.data_ptr(), B, w, h)
        if stereo:
            matchers[k].stereo_match_batch(sp)
        if world > 1:
            with torch.cuda.stream(streams[k]):
                gather_counts(counts_ts[k], world, dist, gathered[k])

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    counts = ext.batch_counts(B)
    n_kp_img = float(counts.mean())

    # ---- per-kernel device time, HIP events on the launch stream (untimed extra steps) ----
    step_no[0] = 0
    nctx = 1                      # the profiled steps run alone on context 0: clean per-kernel durations
    ext.profile_enable(True)
    for _ in range(max(1, args.profile_steps)):
        step()
    torch.cuda.synchronize()
    prof = ext.profile_read()
    ext.profile_enable(False)

    if rank == 0:
        sizes = level_sizes(w, h, ext.GetInverseScaleFactors())
        per_img, survey_total = algorithmic_bytes(w, h, sizes, n_kp_img, stereo)
        stage_ms = {k: v[0] / max(1, args.profile_steps) for k, v in prof.items() if v[1] > 0}
        dom = max(stage_ms, key=stage_ms.get)
        dom_ms_total, dom_launches = prof[dom]
        avg_launch_ms = dom_ms_total / dom_launches
        launches_per_step = dom_launches / max(1, args.profile_steps)
        bytes_per_launch = per_img[dom] * B / launches_per_step
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and tj.get("batch") == B:
                    traffic = tj.get("hbm_bytes_per_launch", {}).get(dom)
            except Exception:
                traffic = None
        total_frames = world * B * args.steps
        value = total_frames / dt
        line = {
            "metric": "frames/sec ORB extract+match, 752x480 @2000 kp" if args.workload == "stereo752" else f"frames/sec ORB {args.workload}",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": cfg_name, "frame": "one camera image (a stereo pair = 2 frames + 1 association)",
                       "images_per_step_per_gpu": B, "stereo_pairs_per_s": round(value / 2, 1) if stereo else None,
                       "width": w, "height": h, "nfeatures": nfeat, "levels": 8, "scale_factor": 1.2, "fast_th": [20, 7],
                       "mean_keypoints_per_image": round(n_kp_img, 1), "contexts_per_gpu": max(1, args.streams), "sharding": f"{world} x independent streams, RCCL all-gather of counts" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                         "avg_launch_ms": round(avg_launch_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "pipeline_frac_hbm": round(value / world * survey_total / 8e12, 5),
                         "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()}},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(w, h, nfeat, stereo)
        print(json.dumps(line), flush=True)
    for e in exts:
        e.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
