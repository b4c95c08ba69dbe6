#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel-trace stats + FETCH_SIZE / WRITE_SIZE passes) into small
text/JSON summaries that are committed under profiles/.

usage: summarize_profile.py <gpurun_out/prof_TAG> <TAG>
HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B...
(rocprofv3 reports them in kilobytes); on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x,
so both the raw and the doubled read figure are listed.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pat):
    r = glob.glob(os.path.join(root, "**", pat), recursive=True)
    return r[0] if r else None


def short(name):
    n = name.split("(")[0]
    return n.replace("[clone .kd]", "").replace("void ", "").strip()


def main():
    src, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
    os.makedirs(dst, exist_ok=True)
    out = {"tag": tag}
    # ---- kernel trace: per-kernel count / avg / total ----
    kt = find(os.path.join(src, "trace"), "*kernel_trace.csv")
    per = defaultdict(list)
    if kt:
        for row in csv.DictReader(open(kt)):
            per[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us"]
    tot = sum(sum(v) for v in per.values()) or 1
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k},{len(v)},{sum(v) / 1e6:.3f},{sum(v) / len(v) / 1e3:.2f},{min(v) / 1e3:.2f},{max(v) / 1e3:.2f},{100 * sum(v) / tot:.1f}%")
    open(os.path.join(dst, f"kernel_stats_{tag}.csv"), "w").write("\n".join(lines) + "\n")
    out["kernels"] = {k: {"calls": len(v), "avg_us": sum(v) / len(v) / 1e3} for k, v in per.items()}
    st = find(os.path.join(src, "trace"), "*kernel_stats.csv")
    if st:
        open(os.path.join(dst, f"rocprof_kernel_stats_{tag}.csv"), "w").write(open(st).read())
    # ---- PMC passes ----
    for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        f = find(os.path.join(src, sub), "*counter_collection.csv")
        acc = defaultdict(list)
        if f:
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == cname:
                    acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
        out[cname] = {k: {"launches": len(v), "avg_per_launch": sum(v) / len(v)} for k, v in acc.items()}
    json.dump(out, open(os.path.join(dst, f"profile_{tag}.json"), "w"), indent=1, sort_keys=True)
    # per-launch HBM traffic of each pipeline stage, in bytes, for bench.py's roofline.traffic.
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  Calibration on this code's own access pattern
    # (4 B per lane): k_fast reads every level exactly once and FETCH_SIZE*1024 comes out at 1.00-1.03x
    # that byte count, so no 2x correction is applied (the gfx950 halving concerns 16 B/lane streams).
    # a stage may be several launches per step (the banded pyramid is two, the per-level fallback up to seven):
    # stage bytes per step = all bytes of the stage's kernels / steps, steps = launches of k_fast (one per step; the quadtree is up to three level-group launches)
    stage_of = (("k_resize", "resize"), ("k_pyramid_bands", "resize"), ("k_blur", "blur"), ("k_fast", "fast"), ("k_quadtree", "quadtree"),
                ("k_orient_desc", "orient_desc"), ("k_stereo_bucket", "stereo_bucket"), ("k_stereo_match", "stereo_match"),
                ("k_stereo_cut", "stereo_cut"))
    traffic, traffic_x2 = {}, {}
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        steps = max([v["launches"] for k, v in out[cname].items() if "k_fast" in k] or [1])
        for k, v in out[cname].items():
            for pat, st in stage_of:
                if pat in k and not (pat == "k_stereo_match" and "sad" in k):
                    b = int(v["avg_per_launch"] * v["launches"] * 1024 / steps)
                    traffic[st] = traffic.get(st, 0) + b
                    # MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies the 128-B requests of a 16-B-per-lane stream at
                    # 64 B -- the doubled figure is the upper bound for kernels that load 16 B per lane
                    traffic_x2[st] = traffic_x2.get(st, 0) + (2 * b if cname == "FETCH_SIZE" else b)
                    break
    # images per launch: what the traced bench line says (bench.py compares it with its own batch before it uses the file)
    batch = int(os.environ.get("GFO_PROF_BATCH", "256"))
    try:
        bl = [l for l in open(os.path.join(src, "bench_trace.log")).read().splitlines() if l.startswith("{")]
        batch = int(json.loads(bl[-1])["config"]["images_per_step_per_gpu"])
    except Exception:
        pass
    meta = {"tag": tag, "workload": os.environ.get("GFO_PROF_WORKLOAD", "stereo752"), "batch": batch,
            "hbm_bytes_per_launch": traffic, "hbm_bytes_per_launch_fetch_x2": traffic_x2,
            "note": "FETCH_SIZE+WRITE_SIZE (KiB) x 1024, separate --pmc passes, averaged per launch; *_fetch_x2 doubles FETCH_SIZE "
                    "(gfx950 counts a 16-B-per-lane streaming read at half its bytes: k_fast and k_orient_desc load 16 B per lane)"}
    json.dump(meta, open(os.path.join(dst, f"traffic_{tag}.json"), "w"), indent=1, sort_keys=True)
    json.dump(meta, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1, sort_keys=True)
    print(open(os.path.join(dst, f"kernel_stats_{tag}.csv")).read())
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        for k, v in sorted(out[cname].items()):
            print(cname, k, v)
    # copy the bench line measured under the tracer
    for name in ("bench_trace.log",):
        p = os.path.join(src, name)
        if os.path.exists(p):
            tail = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if tail:
                open(os.path.join(dst, f"bench_under_trace_{tag}.json"), "w").write(tail[-1] + "\n")


if __name__ == "__main__":
    main()
