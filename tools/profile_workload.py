#!/usr/bin/env python3
"""Per-workload profile records for the workloads that are not the headline (BASELINE configs[1], [3]: extract752, extract1080,
proj1080), on the GPU box:  tools/profile_workload.py <tag> <workload> <batch>
  1. rocprofv3 --kernel-trace --stats of `bench.py --workload W --batch B` (one context)  -> kernel_stats_<W>_<tag>.csv
  2. rocprofv3 --pmc FETCH_SIZE, then --pmc WRITE_SIZE (separate passes, gfx950)            -> traffic_<W>_<tag>.json
  3. rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... -> sq_counters_<W>_<tag>.json
All under gpurun_out/profiles_<tag>/ ; copy what is to be kept into profiles/ (and as *_<W>_latest.json, which bench.py reads for
`issue_frac` / `traffic` of that workload).  The profiled program is `python3 bench.py ...` itself, directly after `--`."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from bench import STAGE_OF_KERNEL, WIDE_READ_STAGES, pmc_bytes_per_step   # noqa: E402  (bench.py imports no torch at module level)


def short(name):
    return name.split("(")[0].replace("[clone .kd]", "").replace("void ", "").strip()


def run_pass(exe, flags, outdir, bench_args, log):
    shutil.rmtree(outdir, ignore_errors=True)
    cmd = [exe] + flags + ["--output-format", "csv", "-d", outdir, "--", "python3", os.path.join(R, "bench.py")] + bench_args
    with open(log, "wb") as fh:
        p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", GFO_BENCH_WATCHDOG="200"), stdout=fh, stderr=subprocess.STDOUT,
                             start_new_session=True)
        try:
            rc = p.wait(timeout=280)
        except subprocess.TimeoutExpired:
            import signal
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            raise SystemExit(f"pass {flags} did not finish in 280 s (log: {log}); stopping, no further GPU step")
    if rc != 0:
        raise SystemExit(f"pass {flags} failed rc {rc} (log: {log})")


def main():
    tag, workload, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    out = os.path.join(R, "gpurun_out", f"profiles_{tag}")
    os.makedirs(out, exist_ok=True)
    common = ["--workload", workload, "--batch", str(batch), "--streams", "1"]
    # 1. kernel trace
    d = os.path.join(out, f"trace_{workload}")
    run_pass(exe, ["--kernel-trace", "--stats"], d, common + ["--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-other-configs",
                                                            "--no-boundary", "--no-live-traffic"], os.path.join(out, f"trace_{workload}.log"))
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    per = defaultdict(list)
    for row in csv.DictReader(open(kt)):
        per[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    tot = sum(sum(v) for v in per.values()) or 1
    lines = [f"# rocprofv3 --kernel-trace --stats -- python3 bench.py {' '.join(common)} --steps 10 --warmup 2 (one context, {batch} images per launch)",
             "kernel,calls,total_ms,avg_us,min_us,max_us,share"]
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f"{k},{len(v)},{sum(v) / 1e6:.3f},{sum(v) / len(v) / 1e3:.2f},{min(v) / 1e3:.2f},{max(v) / 1e3:.2f},{100 * sum(v) / tot:.1f}%")
    open(os.path.join(out, f"kernel_stats_{workload}_{tag}.csv"), "w").write("\n".join(lines) + "\n")
    # 2. HBM traffic (two passes)
    raw = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, f"pmc_{counter}_{workload}")
        run_pass(exe, ["--pmc", counter], d, ["--pmc-child"] + common + ["--steps", "6", "--warmup", "2"], os.path.join(out, f"pmc_{counter}_{workload}.log"))
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
        raw[counter], steps = pmc_bytes_per_step(f, counter)
    rw, x2 = {}, {}
    for st in set(raw["FETCH_SIZE"]) | set(raw["WRITE_SIZE"]):
        f, w = raw["FETCH_SIZE"].get(st, 0.0), raw["WRITE_SIZE"].get(st, 0.0)
        rw[st] = int(f + w)
        x2[st] = int(2 * f + w)
    json.dump({"tag": tag, "workload": workload, "batch": batch, "steps_seen": steps, "hbm_bytes_per_launch": rw, "hbm_bytes_per_launch_fetch_x2": x2,
               "wide_read_stages": list(WIDE_READ_STAGES), "fetch_raw": {k: int(v) for k, v in raw["FETCH_SIZE"].items()},
               "write": {k: int(v) for k, v in raw["WRITE_SIZE"].items()},
               "note": "bytes per STEP and stage (a stage may be several launches per step: the pyramid, the projection rounds); rocprofv3 --pmc "
                       "FETCH_SIZE / WRITE_SIZE separate passes, KiB x 1024; FETCH doubled for the 16-B-per-lane stages per MI355X_MICROARCH.md"},
              open(os.path.join(out, f"traffic_{workload}_{tag}.json"), "w"), indent=1)
    # 3. SQ counters
    d = os.path.join(out, f"pmc_sq_{workload}")
    run_pass(exe, ["--pmc", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD"],
             d, ["--pmc-child"] + common + ["--steps", "3", "--warmup", "1"], os.path.join(out, f"pmc_sq_{workload}.log"))
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(lambda: defaultdict(float))
    nsteps = defaultdict(int)
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        for pat, st in STAGE_OF_KERNEL:
            if pat in kn and not (pat == "k_stereo_match" and "sad" in kn):
                acc[st][row["Counter_Name"]] += float(row["Counter_Value"])
                if st == "fast":
                    nsteps[row["Counter_Name"]] += 1
                break
    n = max(nsteps.values()) if nsteps else 1
    json.dump({"tag": tag, "workload": workload, "batch": batch, "per_stage_per_step": {st: {c: v / n for c, v in cs.items()} for st, cs in acc.items()},
               "note": f"rocprofv3 --pmc SQ_* pass of `bench.py --pmc-child {' '.join(common)} --steps 3 --warmup 1` (tools/profile_workload.py), "
                       f"summed per stage and divided by the {n} steps seen"},
              open(os.path.join(out, f"sq_counters_{workload}_{tag}.json"), "w"), indent=1)
    print(open(os.path.join(out, f"kernel_stats_{workload}_{tag}.csv")).read())
    print("profile_workload", workload, "done ->", out)


if __name__ == "__main__":
    main()
