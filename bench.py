#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X ORB front-end on synthetic frames, the roofline figure of the dominant
kernel and the CPU oracle timed beside it.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python bench.py --gpus 8 --steps 20 --warmup 3        (no launcher: this process starts the 8 ranks itself, spawn_ranks())
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json metric): `stereo752` = stereo 752x480 pairs at 2000 features, extract L+R + the
stereo Hamming association.  A "step" is one pass of the hot path over one resident batch of `--batch` images per
GPU: pyramid -> blur -> FAST cells -> quadtree -> orientation+descriptor for every image, then the matcher of the
workload (stereo association of every pair / SearchByProjection of every frame against the resident local map).
Inputs are already in HBM when the timed region starts; the steps walk over several distinct input batches so that
no step finds its input in the Infinity Cache.  Independent frames shard one batch per GPU (weak scaling); the only
collective is the RCCL all-gather of the per-image keypoint counts.  Rank 0 prints ONE JSON line; the other
BASELINE configs are measured after the headline (the same command with --workload, each in a process of its own) and
reported in the same line under "other_configs".
The steps alternate between a few independent contexts (arena + HIP stream; CONTEXTS per workload), for stereo752 two of
them chained behind each other's pyramid (gfo_ctx_chain).  Setup ends with PRIME_STEPS untimed batches (clock ramp of a
fresh process, config.priming_steps); --warmup is run as given, untimed, directly in front of the timed steps.
roofline.traffic: measured in the run itself at N = 1 (two short child passes of this command under `rocprofv3 --pmc
FETCH_SIZE` / `--pmc WRITE_SIZE`, started before the process initialises the GPU and bounded at 75 s each: live_traffic());
--no-live-traffic (and every spawned rank, and a run under a profiler) takes the committed summary of the latest such
measurement instead, profiles/traffic_latest.json, and says so (`traffic_measured: false`).
Beside `value` (inputs and results resident in HBM) the line carries `value_with_h2d` (every step's batch copied in from
pinned host memory), `value_delivered` (copied in AND every result -- counts, keypoints, descriptors, stereo outputs --
landed in pinned host memory: gfo_batch_deliver), `per_frame_boundary` (the reference's own call pattern, one stereo
frame per call from K host threads through the C ABI: tools/c/boundary_throughput.c) and `matcher_calls` (ms per call of the
host-array matcher calls Tracking makes per frame: tools/matcher_call_latency.py).
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
    # name: (w, h, nfeatures, matcher, BASELINE.json config it corresponds to)
    "stereo752": (752, 480, 2000, "stereo", "configs[2]: stereo 752x480 pair, 2000 features/image, extract L+R + stereo Hamming match"),
    "extract752": (752, 480, 2000, None, "configs[1]: 752x480 synthetic stream, 2000 features, extract-only"),
    "extract1080": (1920, 1080, 4000, None, "configs[3] (extract part): 1920x1080, 4000 features"),
    "proj1080": (1920, 1080, 4000, "project", "configs[3]: 1920x1080 @4000 features, extract + SearchByProjection against a 50k-descriptor synthetic local map"),
}
# independent contexts (arena + HIP stream) the steps alternate between, per workload: the measured best on MI355X
# (same-box A/B, tools/ab_args.sh: stereo752 2/3 contexts 214k/222k, extract752 239k/228k, extract1080 58.9k/60.0k,
#  proj1080 50.0k/47.3k frames/s) -- how many kernels may share the chip before they only take each other's wave slots
PRIME_STEPS = 48   # untimed batches at the end of Job setup (clock ramp of a fresh process, see Job.__init__)
CONTEXTS = {"stereo752": 2, "extract752": 2, "extract1080": 3, "proj1080": 2}
# gfo_ctx_chain stage per workload (0 = free-running): with two contexts the phase between their kernel chains settles at
# random after every synchronisation -- stereo752 then runs at 214k or 225k frames/s; each context chained after the other's
# pyramid gives 227k every time (3 free-running contexts: 221k).  extract752's free-running phase is the better one
# (241k against 232k chained); the 1080p workloads do not care.
CHAIN_STAGE = {"stereo752": 1, "extract752": 0, "extract1080": 0, "proj1080": 0}
FX, BF = 435.2046959714599, 47.90639384423901
ISSUE_CYCLES, N_SIMDS, CLOCK_HZ = 4, 1024, 2.4e9   # MI355X_MICROARCH.md: vector-instruction issue cost, 256 CUs x 4 SIMDs, max clock
HEADLINE_BATCH = 256   # == gf_orb_slam2_amd.HEADLINE_BATCH (checked in main(); the launcher process must not import the product)
MAP_POINTS = 50000


class _DevArray:
    """Exposes a raw device pointer to torch through __cuda_array_interface__."""

    def __init__(self, ptr, n, typestr="<i4"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def level_sizes(w, h, inv_scale):
    out = []
    for s in inv_scale:
        out.append((int(np.rint(np.float32(w) * np.float32(s))), int(np.rint(np.float32(h) * np.float32(s)))))
    return out


def algorithmic_bytes(w, h, sizes, n_kp_img, matcher, m_map=MAP_POINTS):
    """Algorithmic HBM bytes per IMAGE of each stage (DESIGN.md 'bytes per unit')."""
    P = sum(a * b for a, b in sizes)
    px = [a * b for a, b in sizes]
    stereo_b = (2 * n_kp_img * 60 + n_kp_img * 8) / 2 if matcher == "stereo" else 0
    project_b = m_map * (32 + 24) + n_kp_img * 60 + m_map * 8 if matcher == "project" else 0
    d = {
        "resize": sum(px[l - 1] + px[l] for l in range(1, len(px))),   # read level l-1, write level l
        "blur": 2 * P,                                                   # read level, write blurred level
        "fast": P,                                                       # read every level once
        "quadtree": 0,
        "orient_desc": n_kp_img * (749 + 1369 + 60),                     # disc + 37x37 window + kp/desc out
        "stereo_match": stereo_b,                                        # per image = half a pair
        "stereo_bucket": 0,
        "stereo_cut": 0,
        "project": project_b,                                            # M (desc + projection) + N (desc + kp) + M results
    }
    survey_total = w * h + 4 * P + n_kp_img * 2178 + stereo_b + project_b
    return d, survey_total


STAGE_OF_KERNEL = (("k_resize", "resize"), ("k_pyramid_bands", "resize"), ("k_blur", "blur"), ("k_fast", "fast"), ("k_quadtree", "quadtree"),
                   ("k_orient_desc", "orient_desc"), ("k_stereo_bucket", "stereo_bucket"), ("k_stereo_match", "stereo_match"),
                   ("k_stereo_cut", "stereo_cut"), ("k_proj_", "project"))
# stages whose kernels read 16 bytes per lane: MI355X_MICROARCH.md (HBM): gfx950 tallies such a stream's 128-byte requests
# at 64 bytes, so FETCH_SIZE is doubled for them; the other stages read 4 bytes per lane and were calibrated at 1.00-1.03x
# of a known byte count on this code's own access pattern (tools/summarize_profile.py)
WIDE_READ_STAGES = ("fast", "orient_desc")


def under_profiler():
    """rocprofv3 (or a tool built on rocprofiler) is wrapped around this process: child processes would inherit it"""
    return any("rocprof" in (os.environ.get(k) or "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY")) \
        or any(k.startswith("ROCPROF") for k in os.environ)


def pmc_bytes_per_step(csv_path, counter):
    """rocprofv3 counter_collection.csv -> ({stage: bytes per step}, steps).  FETCH_SIZE / WRITE_SIZE come in KiB per
    dispatch; a stage may be several launches per step (the pyramid, the quadtree's level groups), a step is one k_fast launch."""
    import csv
    per, launches = {}, {}
    for row in csv.DictReader(open(csv_path)):
        if row.get("Counter_Name") != counter:
            continue
        kn = row["Kernel_Name"]
        for pat, st in STAGE_OF_KERNEL:
            if pat in kn and not (pat == "k_stereo_match" and "sad" in kn):
                per[st] = per.get(st, 0.0) + float(row["Counter_Value"]) * 1024.0
                launches[st] = launches.get(st, 0) + 1
                break
    steps = launches.get("fast", 0)
    return ({st: v / steps for st, v in per.items()} if steps else {}), steps


LIVE_TRAFFIC_KILLED = None   # set when a counter pass had to be killed: the JSON line says so (roofline.traffic_pass_killed)


def live_traffic(workload, batch):
    """HBM bytes per step and stage from the PMC counters, measured NOW: two child passes of this script under
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (the two do not fit one pass on gfx950), run before this process
    touches the GPU.  Returns (per_stage dict, note) -- the dict is None when the profiler is missing, refuses or fails
    (then the committed profiles/traffic_latest.json is used and labelled as such)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if under_profiler():
        return None, "this run is itself under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    raw, steps_seen = {}, 0
    logdir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else "/tmp"
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        if raw:
            # both stalled passes of round 2 "followed another counter-collection session within a second"
            # (profiles/NOTEBOOK.md); the cause was never established, so the second session does not follow the first at once
            time.sleep(3.0)
        tmp = tempfile.mkdtemp(prefix="gfo_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", tmp, "--", "python3", os.path.join(ROOT, "bench.py"),
               "--pmc-child", "--workload", workload, "--batch", str(batch), "--steps", "6", "--warmup", "2", "--streams", "1"]
        # the child dumps every thread's stack after 50 s (faulthandler) and exits: if a pass ever stalls again, the log
        # says whether python was inside libgfo, inside the HIP runtime, or already gone (then it is the profiler's teardown)
        env = dict(os.environ, TMPDIR="/tmp", GFO_BENCH_WATCHDOG="50")
        log_path = os.path.join(logdir, f"pmc_pass_{counter}.log")
        try:
            # a pass takes ~5 s; it gets 75 s (a cold box pages torch in for a minute) and is killed as a GROUP -- rocprofv3
            # is a launcher, the process that holds the GPU is its child -- so that a stalled profiler can neither delay
            # the benchmark for long nor sit on the GPU while it is timed
            with open(log_path, "wb") as log:
                proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, start_new_session=True)
                try:
                    rc = proc.wait(timeout=75)
                except subprocess.TimeoutExpired:
                    import signal
                    try:      # who is still there, and where: evidence first, then the kill
                        ps = subprocess.run(["ps", "-o", "pid,ppid,stat,etimes,wchan:24,args", "-g", str(os.getpgid(proc.pid))],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=5).stdout
                        log.write(b"\n---- pass exceeded 75 s; process group at that moment ----\n" + ps)
                    except Exception:
                        pass
                    pgid = proc.pid
                    try:
                        os.killpg(pgid, signal.SIGKILL)
                    except OSError:
                        pass
                    proc.wait()
                    # nothing of that group may still hold the GPU when the headline is timed: wait until the group is empty
                    # (a killed process leaves the device once the kernel has reaped it), then a settling pause
                    gone = False
                    for _ in range(100):
                        left = subprocess.run(["ps", "-o", "pid=", "-g", str(pgid)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.strip()
                        if not left:
                            gone = True
                            break
                        time.sleep(0.1)
                    time.sleep(2.0)
                    global LIVE_TRAFFIC_KILLED
                    LIVE_TRAFFIC_KILLED = {"counter": counter, "process_group_gone": gone, "log": log_path}
                    return None, f"rocprofv3 --pmc {counter} pass did not finish in 75 s and was killed (evidence: {log_path})"
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} pass failed (rc {rc}; log: {log_path})"
            per_step, steps = pmc_bytes_per_step(files[0], counter)
            if steps == 0:
                return None, f"no kernels in the {counter} pass"
            raw[counter] = per_step
            steps_seen = steps
        except Exception as ex:   # unreadable output: the headline must not depend on the profiler
            return None, f"rocprofv3 --pmc {counter} pass: {ex!r}"
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for st in set(raw["FETCH_SIZE"]) | set(raw["WRITE_SIZE"]):
        f, w = raw["FETCH_SIZE"].get(st, 0.0), raw["WRITE_SIZE"].get(st, 0.0)
        out[st] = {"fetch_raw": int(f), "write": int(w), "bytes": int((2.0 * f if st in WIDE_READ_STAGES else f) + w)}
    return out, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes ({steps_seen} steps each, one context), "
                 "KiB x 1024, per step; FETCH doubled for the stages that read 16 B per lane "
                 f"({', '.join(WIDE_READ_STAGES)}) as MI355X_MICROARCH.md prescribes for gfx950")



def side_config(name, batch, streams):
    """One of the other BASELINE configs: THIS command with --workload <name> in a process of its own (400 steps), its
    line condensed.  A fresh process because the rate depends on which hardware queue the runtime binds every stream to
    on first use: a second job in a process that has already run one inherits a worse assignment (extract752 233k against
    249k, proj1080 46.5k against 51.2k frames/s, measured both ways)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", name, "--batch", str(batch), "--steps", "400", "--warmup", "30",
           "--streams", str(streams), "--no-other-configs", "--no-cpu-baseline", "--no-boundary", "--no-live-traffic", "--profile-steps", "5"]   # (each verifies its own last step)
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=240)
        lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
        if r.returncode not in (0, 4) or not lines:      # (4: the child printed its line and failed on verified.mismatches)
            return {"name": name, "error": f"child process failed (rc {r.returncode})"}
        j = json.loads(lines[-1])
        return {"workload": j["config"]["workload"], "name": name, "value": j["value"], "unit": "frames/s", "images_per_step": batch,
                "verified": j["config"].get("verified"),
                "contexts": j["config"]["contexts_per_gpu"], "steps": j["steps"], "ms_per_step": j["ms_per_step"],
                "process": "its own (same command with --workload)", "roofline": j["roofline"]}
    except Exception as ex:      # a failing side measurement must not lose the headline line
        return {"name": name, "error": repr(ex)}


def per_frame_boundary(seconds=1.0):
    """The reference's own call pattern through the C ABI: one stereo frame per call, K host threads (K camera streams),
    tools/c/boundary_throughput.c built with the host compiler and run as a child process (it initialises the GPU itself).
    gfo_extract_stereo at K = 1, 4, 8, 16 and the adapter's two-context pattern at K = 1 (the one configuration GF-ORB-SLAM2
    itself runs: one stereo camera), 4, 8, contexts combining as the adapter sets them (gfo_ctx_set_combining)."""
    import shutil
    import subprocess
    cc = shutil.which("gcc") or shutil.which("cc")
    if not cc:
        return {"error": "no C compiler on this host: tools/c/boundary_throughput.c not built"}
    exe = "/tmp/gfo_boundary_throughput"
    try:
        subprocess.run([cc, "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "c", "boundary_throughput.c"), "-o", exe,
                        "-ldl", "-lpthread", "-lm"], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        from gf_orb_slam2_amd._lib import lib_path
        pts = []
        for mode, ks in (("stereo", "1,4,8,16"), ("adapter", "1,4,8")):
            r = subprocess.run([exe, lib_path(), os.path.join(ROOT, "tests", "golden"), str(seconds), mode, ks, "1"], stdout=subprocess.PIPE,
                               stderr=subprocess.DEVNULL, timeout=180)
            j = json.loads(r.stdout.decode(errors="replace"))
            for p in j["points"]:
                pts.append({k: p[k] for k in ("path", "combining", "streams", "host_threads", "images_per_s", "stereo_frames_per_s",
                                              "frames_per_device_batch", "latency_ms", "stereo_rig", "result_mismatches", "errors",
                                              "contexts_created_in_timed_region", "arenas_planned_in_timed_region")})
            if r.returncode != 0:
                return {"error": f"harness rc {r.returncode}", "points": pts}
        best = max((p for p in pts if p["path"] == "gfo_extract_stereo"), key=lambda p: p["images_per_s"])
        return {"workload": j["workload"], "unit": "images/s", "harness": "tools/c/boundary_throughput.c (C, dlopen of libgfo.so, no Python in the loop)",
                "seconds_per_point": seconds, "best_images_per_s": best["images_per_s"], "best_at_streams": best["streams"], "points": pts}
    except Exception as ex:      # a failing side measurement must not lose the headline line
        return {"error": repr(ex)}


def matcher_calls():
    """The matcher calls Tracking makes on every frame, caller arrays in and out, ms per call (tools/matcher_call_latency.py as a
    child process: SearchByProjection(F, MapPoints) for three map sizes, SearchByProjection(Cur, Last), ComputeBoW, SearchByBoW on the
    EuRoC frame).  A side measurement like per_frame_boundary: never the metric."""
    import subprocess
    try:
        env = dict(os.environ, JSON="1")
        for k in ("TH", "ONLY", "GFO_PROJ_STATS"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "matcher_call_latency.py")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           timeout=180, env=env, cwd=ROOT)
        if r.returncode != 0:
            return {"error": f"tools/matcher_call_latency.py rc {r.returncode}"}
        return json.loads(r.stdout.decode(errors="replace").strip().splitlines()[-1])
    except Exception as ex:      # a failing side measurement must not lose the headline line
        return {"error": repr(ex)}


def real_image():
    """The headline pipeline on the reference's own EuRoC images instead of the synthetic stream (tools/bench_real_image.py,
    a process of its own): real imagery sends half of the FAST cells through the second, minThFAST, round."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_real_image.py")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=240)
        lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{\"")]
        if r.returncode != 0 or not lines:
            return {"error": f"child process failed (rc {r.returncode})"}
        return json.loads(lines[-1])
    except Exception as ex:      # a failing side measurement must not lose the headline line
        return {"error": repr(ex)}


def host_cores():
    """(cores this process may be scheduled on, cores' worth of CPU time its cgroup grants, or None).  A GPU box shows all 256
    hardware threads in the affinity mask while the container's cgroup grants a fraction of them (cpu.max): a pool sized by the
    mask alone is throttled by the quota -- round 3's "256 threads = 12.5 x one thread"."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        cg = open("/proc/self/cgroup").read().strip().splitlines()
        rel = [l.split(":", 2)[2] for l in cg if l.startswith("0::")]
        paths = (["/sys/fs/cgroup" + rel[0].rstrip("/") + "/cpu.max"] if rel else []) + ["/sys/fs/cgroup/cpu.max"]
        for path in paths:
            if os.path.exists(path):
                q, per = open(path).read().split()[:2]
                if q != "max":
                    quota = float(q) / float(per)
                break
        if quota is None and os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
    except Exception:
        quota = None
    return aff, quota


def cpu_model():
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                return l.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _cpu_inputs(w, h, euroc):
    from gf_orb_slam2_amd.synth import synth_stereo_pair
    if euroc:
        gold = os.path.join(ROOT, "tests", "golden")
        l = np.fromfile(os.path.join(gold, "EuRoC_l_752x480.u8"), np.uint8).reshape(480, 752)
        r = np.fromfile(os.path.join(gold, "EuRoC_r_752x480.u8"), np.uint8).reshape(480, 752)
        return [(l, r)]
    return [synth_stereo_pair(w, h, 1000 + i) for i in range(8)]


def cpu_worker(argv):
    """One CPU-baseline worker process (bench.py --cpu-worker w h nfeatures stereo euroc n_pairs start_epoch): a process of its
    own per core -- no GIL, no shared allocator, never touches the GPU -- that waits for the common start time, runs n_pairs
    stereo pairs through the oracle and prints its start and end times."""
    w, h, nfeat, stereo, euroc, n_pairs = (int(x) for x in argv[:6])
    t_start = float(argv[6])
    from oracle import orb_oracle as O
    pairs = _cpu_inputs(w, h, euroc)
    oe_l, oe_r = O.OracleExtractor(nfeat, 1.2, 8, 20, 7), O.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    sf = oe_l.scale_factors
    oe_l(pairs[0][0])                      # first touch of the library and its buffers, untimed
    while time.time() < t_start:
        time.sleep(0.001)
    t0 = time.time()
    for i in range(n_pairs):
        l, r = pairs[i % len(pairs)]
        kl, dl = oe_l(l)
        kr, dr = oe_r(r)
        if stereo:
            O.stereo_match(kl, dl, kr, dr, sf, h, BF, BF / FX, 0.0)
    print(json.dumps({"t0": t0, "t1": time.time(), "late": t0 - t_start}), flush=True)


def cpu_processes(w, h, nfeat, stereo, euroc, nproc, n_pairs):
    """nproc worker processes, one stereo pair stream each, started together; images/s over the span first start .. last end"""
    import subprocess
    t_start = time.time() + 4.0 + 0.02 * nproc        # interpreter + numpy + inputs of every worker are up by then (checked: `late`)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker"] + [str(x) for x in (w, h, nfeat, int(stereo), int(euroc), n_pairs, t_start)]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL) for _ in range(nproc)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
            outs.append(json.loads(o.decode().strip().splitlines()[-1]))
        except Exception:
            p.kill()
    if len(outs) != nproc:
        return None
    span = max(o["t1"] for o in outs) - min(o["t0"] for o in outs)
    return {"processes": nproc, "value": round(2 * n_pairs * nproc / span, 2), "pairs_per_process": n_pairs,
            "max_start_delay_s": round(max(o["late"] for o in outs), 3)}


def cpu_baseline(w, h, nfeatures, stereo, budget_s=10.0):
    """The CPU oracle (oracle/, kind 'port') on a bounded sample of the same synthetic stream, on this host:
    one thread (the headline `value`), two threads (the reference's own left/right arrangement, Frame.cc:84-87)
    and every core this process is GRANTED -- min(affinity mask, cgroup quota) -- as one process per core with one stereo pair
    stream each.  The same three on the reference's own EuRoC pair ("euroc": BASELINE configs[0]: the CPU
    extractor on test/EuRoC_l.png).  ctypes releases the GIL, so the two threads run the C oracle concurrently."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import orb_oracle as O
    O.build()

    def make():
        return O.OracleExtractor(nfeatures, 1.2, 8, 20, 7)

    sf = make().scale_factors

    def one_pair(pairs, oe_l, oe_r, idx, two_threads=None, match=stereo):
        l, r = pairs[idx % len(pairs)]
        if two_threads is not None:
            fl = two_threads.submit(oe_l, l)
            kr, dr = oe_r(r)
            kl, dl = fl.result()
        else:
            kl, dl = oe_l(l)
            kr, dr = oe_r(r)
        if match:
            O.stereo_match(kl, dl, kr, dr, sf, h, BF, BF / FX, 0.0)

    def one_and_two(pairs, budget):
        oe_l, oe_r = make(), make()
        t0 = time.perf_counter()
        one_pair(pairs, oe_l, oe_r, 0)
        t_first = time.perf_counter() - t0
        n1 = int(max(3, min(60, budget / max(t_first, 1e-3))))
        t0 = time.perf_counter()
        for i in range(n1):
            one_pair(pairs, oe_l, oe_r, i)
        v1 = 2 * n1 / (time.perf_counter() - t0)
        n2 = max(3, n1 // 2)       # two threads: left and right image extracted concurrently
        with ThreadPoolExecutor(1) as side:
            t0 = time.perf_counter()
            for i in range(n2):
                one_pair(pairs, oe_l, oe_r, i, side)
            v2 = 2 * n2 / (time.perf_counter() - t0)
        return n1, v1, v2

    aff, quota = host_cores()
    granted = max(1, min(aff, int(quota + 0.5)) if quota else aff)
    pairs = _cpu_inputs(w, h, False)
    n1, v1, v2 = one_and_two(pairs, budget_s * 0.35)
    per = max(2, n1 // 4)
    vn = cpu_processes(w, h, nfeatures, stereo, False, granted, per)
    out = {"value": round(v1, 2), "unit": "frames/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
           "sample": f"{n1} stereo pairs ({2 * n1} images) of the same {w}x{h} synthetic stream, oracle/orb_oracle.c, "
                     f"single thread; {os.cpu_count()} hardware threads present, {aff} in this process's affinity mask, "
                     f"cgroup CPU quota {('%.1f cores' % quota) if quota else 'none'} -> {granted} granted",
           "threads_2": round(v2, 2), "all_granted_cores": vn,
           "scaling_vs_one_thread": round(vn["value"] / v1 / granted, 3) if vn else None,
           "reference_published": "13.7-22.2 ms per stereo frame on unstated hardware (README.md:7-16) = 90-146 images/s"}
    if (w, h) == (752, 480):
        # BASELINE configs[0]: the CPU ORBextractor on the reference's test/EuRoC_l.png (752x480, 8 levels, 2000 features);
        # committed here as tests/golden/EuRoC_l_752x480.u8 (raw pixels of that PNG)
        ep = _cpu_inputs(w, h, True)
        oe = make()
        t0 = time.perf_counter()
        oe(ep[0][0])
        ne = int(max(5, min(100, 0.1 * budget_s / max(time.perf_counter() - t0, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(ne):
            oe(ep[0][0])
        v_l = ne / (time.perf_counter() - t0)
        ne1, ve1, ve2 = one_and_two(ep, budget_s * 0.15)
        ven = cpu_processes(w, h, nfeatures, stereo, True, granted, max(2, ne1 // 4))
        out["euroc"] = {"config": "BASELINE configs[0]: CPU ORBextractor on test/EuRoC_l.png, 752x480, 8 levels, 2000 features",
                        "extract_only_left_image": {"value": round(v_l, 2), "unit": "frames/s", "cores": 1, "images": ne},
                        "stereo_pair_extract_and_match": {"value": round(ve1, 2), "threads_2": round(ve2, 2), "unit": "frames/s", "pairs": ne1,
                                                          "all_granted_cores": ven}}
    return out


def spawn_ranks(n, argv, selftest=False):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): this process -- which has not
    imported torch and never touches the GPU -- starts N fresh child processes of this same command, one per GPU, with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would have set), relays
    rank 0's single JSON line, and exits non-zero if any rank does.  No exec anywhere: the children are ordinary
    subprocesses, each a session of its own so that the whole rank can be killed as a group when a bound is hit
    (GFO_BENCH_SPAWN_TIMEOUT seconds, default 900) or when a sibling has failed (the survivors would otherwise wait in
    a collective forever)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    bound = float(os.environ.get("GFO_BENCH_SPAWN_TIMEOUT", "900"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GFO_BENCH_SPAWNED="1")
        # (HSA_ENABLE_IPC_MODE_LEGACY: see main() -- every rank sets it for itself, whichever launcher started it)
        cmd = [sys.executable, os.path.abspath(__file__)] + list(argv) + (["--spawn-selftest"] if selftest else [])
        # rank 0's stdout carries the line; the other ranks print nothing on stdout by contract, and if they do it goes to stderr
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                pass

    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().decode(errors="replace").splitlines()), daemon=True)
    reader.start()
    t0, failed = time.monotonic(), None
    try:
        while any(p.poll() is None for p in procs):
            bad = [(r, p.returncode) for r, p in enumerate(procs) if p.poll() is not None and p.returncode != 0]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if time.monotonic() - t0 > bound:
                failed = f"ranks still running after {bound:.0f} s (GFO_BENCH_SPAWN_TIMEOUT)"
                break
            time.sleep(0.05)
    finally:
        if failed or any(p.poll() is None for p in procs):
            kill_all()
    reader.join(timeout=10)
    bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
    if failed is None and bad:
        failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
    lines = [l for l in out0 if l.startswith("{")]
    for l in lines:
        print(l, flush=True)
    if os.environ.get("GFO_BENCH_PARENT_REPORT"):      # tests: what this launcher process itself has loaded
        print("parent_modules " + json.dumps(sorted(m for m in sys.modules if m.split(".")[0] in ("torch", "gf_orb_slam2_amd", "oracle"))),
              file=sys.stderr, flush=True)
    if failed:
        print(f"bench.py --gpus {n}: {failed}", file=sys.stderr, flush=True)
        return 1
    if len(lines) != 1:
        print(f"bench.py --gpus {n}: rank 0 printed {len(lines)} JSON lines, expected 1", file=sys.stderr, flush=True)
        return 1
    return 0


class Job:
    """One workload on this rank: input batches resident in HBM, `nctx` independent contexts (arena + HIP stream)
    the steps alternate between, so the tail of one batch overlaps the head of the next."""

    def __init__(self, G, torch, name, B, nctx, local_rank, rank, world, dist, n_inputs=4, prime=True, gather_every=1, one_stream=False):
        from gf_orb_slam2_amd.sharding import shard_pairs
        from gf_orb_slam2_amd.synth import synth_local_map, synth_stereo_pair, synth_stream
        self.G, self.torch, self.name, self.B, self.world, self.dist = G, torch, name, B, world, dist
        self.w, self.h, self.nfeat, self.matcher, self.cfg_name = WORKLOADS[name]
        w, h = self.w, self.h
        offs = None
        if self.matcher == "project":
            frames, offs = synth_stream(w, h, B, idx=3 + (0 if one_stream else rank))     # one scene per rank, a moving camera
        else:
            frames = []
            # weak scaling: distinct pairs per rank (independent camera streams); strong scaling: every rank holds the ONE stream
            for p in shard_pairs(0 if one_stream else rank, 1 if one_stream else world, B // 2):
                l, r = synth_stereo_pair(w, h, p)
                frames += [l, r]
        base = np.stack(frames)
        # several distinct input batches (vertical rolls of the first: new images for FAST, same disparities) so that
        # consecutive steps do not re-read one batch out of the 256 MB Infinity Cache
        self.host_batches = [base] + [np.roll(base, 37 * k, axis=1) for k in range(1, n_inputs)] if self.matcher != "project" else [base]
        self.d_inputs = [torch.from_numpy(b).cuda() for b in self.host_batches]
        self.L = G.load_library()
        self.sp = G.StereoParams(h, BF, BF / FX, 0.0)
        self.exts, self.matchers, self.streams, self.counts_ts, self.gathered = [], [], [], [], []
        for k in range(nctx):
            st = torch.cuda.Stream()
            e = G.ORBextractor(self.nfeat, 1.2, 8, 20, 7, device=local_rank, max_batch=B)
            e.set_stream(st.cuda_stream)
            # first pass plans the arena; then bind the device-side count vector for the collective
            e.extract_batch_device(self.d_inputs[0].data_ptr(), B, w, h)
            p_kp, p_desc, p_cnt, stride = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int()
            self.L.gfo_batch_device_views(e.handle, ctypes.byref(p_kp), ctypes.byref(p_desc), ctypes.byref(p_cnt), ctypes.byref(stride))
            self.exts.append(e)
            self.matchers.append(G.ORBmatcher(0.8, True, extractor=e))
            self.streams.append(st)
            self.counts_ts.append(torch.as_tensor(_DevArray(p_cnt.value, B), device="cuda"))
            self.gathered.append(torch.zeros(world * B, dtype=torch.int32, device="cuda") if world > 1 else None)
        # (GFO_BENCH_CHAIN=<stage> overrides the table for experiments; 0 = free-running)
        self.chain_stage = int(os.environ["GFO_BENCH_CHAIN"]) if os.environ.get("GFO_BENCH_CHAIN") else CHAIN_STAGE.get(name, 0)
        self.chained = nctx > 1 and self.chain_stage > 0
        self.chain(True)
        self.bounds = (0.0, 0.0, float(w), float(h))
        self.d_mps = None
        if self.matcher == "project":
            kp0, desc0 = self.exts[0].batch_fetch(0)
            mpd, mps = synth_local_map(kp0, desc0, offs, w, h, MAP_POINTS, 4000, seed=7)
            self.mpd, self.mps = mpd, mps                                              # host copies: verify() hands them to the oracle
            self.d_mps = torch.from_numpy(mps.view(np.uint8).reshape(B, -1)).cuda()   # [B][M] projections, resident
            for m in self.matchers:
                m.map_upload(mpd)                                                      # one resident map per context
        self.step_no = 0
        self.nctx = nctx
        self.gather_every = max(0, int(gather_every))   # 0 = never: separates straggler coupling from kernel time on a real node
        self.deliver_blocks = None
        # Setup ends with PRIME_STEPS untimed batches through the whole pipeline: a freshly started process finds the chip
        # at its idle clock and the first ~25 batches run at 0.60-0.70 ms instead of 0.55 (tools/short_run_probe.py) -- with
        # the driver's --steps 20 --warmup 3 that start-up transient is most of the measurement (212k against 236k
        # frames/s sustained).  It is part of setup like the arena-planning pass above, reported as config.priming_steps,
        # and independent of --warmup, which is still run, untimed, right in front of the timed steps.
        for _ in range(PRIME_STEPS if prime else 0):
            self.step()
        torch.cuda.synchronize()

    def chain(self, on):
        """(un)chain the contexts in a ring: context k starts behind context k-1's stage"""
        if not self.chained:
            return
        n = len(self.exts)
        for k in range(n):
            self.exts[k].chain_after(self.exts[(k - 1) % n] if on else None, self.chain_stage)

    def step(self, ctx=None, h2d_from=None, deliver=False, item=None, idle=False):
        """one batch through the pipeline.  item: the batch's index in ONE stream dealt over the ranks (strong scaling: it picks the
        input batch); idle: this rank has no item in the current round of such a stream -- it only takes part in the count exchange."""
        k = self.step_no % self.nctx if ctx is None else ctx
        in_idx = (self.step_no if item is None else item) % len(self.d_inputs)
        d_in = self.d_inputs[in_idx]
        self.step_no += 1
        if not idle:
            self.last_in, self.last_ctx = in_idx, k      # what verify() compares with the oracle
        if idle:
            if self.world > 1 and self.gather_every and self.step_no % self.gather_every == 0:
                from gf_orb_slam2_amd.sharding import gather_counts
                with self.torch.cuda.stream(self.streams[k]):
                    gather_counts(self.counts_ts[k], self.world, self.dist, self.gathered[k])
            return
        if h2d_from is not None:
            # PCIe-inclusive variant: the batch comes from pinned host memory first.  ONE copy stream feeds all contexts
            # (concurrent H2D copies from several streams share the link badly: 35 GB/s aggregate against 56 GB/s for one
            # stream, measured), double-buffered through the input batches: a buffer is overwritten only after the step
            # that last read it, and a step starts only after its copy.
            torch = self.torch
            if not hasattr(self, "copy_stream"):
                self.copy_stream = torch.cuda.Stream()
                self.buf_done = [None] * len(self.d_inputs)
            b = (self.step_no - 1) % len(self.d_inputs)
            if self.buf_done[b] is not None:
                self.copy_stream.wait_event(self.buf_done[b])
            with torch.cuda.stream(self.copy_stream):
                d_in.copy_(h2d_from, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
            self.streams[k].wait_event(ev)
        self.exts[k].extract_batch_device(d_in.data_ptr(), self.B, self.w, self.h)
        if self.matcher == "stereo":
            self.matchers[k].stereo_match_batch(self.sp)
        elif self.matcher == "project":
            self.matchers[k].search_by_projection_batch(self.d_mps.data_ptr(), self.bounds, th=3.0, device_ptrs=True)
        if h2d_from is not None:
            done = self.torch.cuda.Event()
            done.record(self.streams[k])
            self.buf_done[(self.step_no - 1) % len(self.d_inputs)] = done
        if deliver:
            # every result of the batch -> pinned host memory (one block per context), on the context's copy stream: the D2H
            # runs beside the next step's kernels and opposite to its H2D (the link is full duplex)
            if self.deliver_blocks is None:
                lay = self.exts[k].batch_deliver()
                self.deliver_layout = lay
                self.deliver_blocks = [self.torch.empty(lay.bytes, dtype=self.torch.uint8).pin_memory() for _ in range(self.nctx)]
            self.exts[k].batch_deliver(self.deliver_blocks[k].data_ptr(), self.deliver_layout.bytes)
        if self.world > 1 and self.gather_every and self.step_no % self.gather_every == 0:
            from gf_orb_slam2_amd.sharding import gather_counts
            with self.torch.cuda.stream(self.streams[k]):
                gather_counts(self.counts_ts[k], self.world, self.dist, self.gathered[k])

    def timed(self, steps, warmup, barrier=True, **kw):
        torch, dist, world = self.torch, self.dist, self.world
        for _ in range(warmup):
            self.step(**kw)
        if world > 1 and barrier:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(**kw)
        torch.cuda.synchronize()
        if kw.get("deliver"):
            for e in self.exts:
                e.deliver_wait()
        if world > 1 and barrier:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1 and barrier:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    def timed_strong(self, total_steps, warmup):
        """STRONG scaling ("for single-stream scaling, round-robin batches of frames across ranks"): ONE stream of
        total_steps batches, batch i goes to rank i % world (sharding.shard_round_robin); the work is fixed, every rank does its share,
        the time is the slowest rank's between two barriers.  The count all-gather stays one per round of `world` batches; a rank
        without a batch in the last, partial round only takes part in the exchange.  Returns (seconds, batches this rank processed)."""
        from gf_orb_slam2_amd.sharding import shard_round_robin
        torch, dist, world = self.torch, self.dist, self.world
        rank = dist.get_rank() if (dist is not None and world > 1) else 0
        mine = list(shard_round_robin(total_steps, rank, world))
        rounds = (total_steps + world - 1) // world
        for _ in range(warmup):
            self.step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for j in range(rounds):
            i = j * world + rank
            if i < total_steps:
                self.step(item=i)
            else:
                self.step(idle=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, len(mine)

    def verify(self, n_images=4, n_pairs=2):
        """The checker role of the oracle, outside every timed region: results of the LAST step submitted (still in that context's
        arena) -- n_images images (first pair and last pair of the batch) and n_pairs matcher units (stereo pairs / projected
        frames) -- compared with the CPU oracle on the same host images, bit for bit.  Call right after timed()."""
        from oracle import orb_oracle as O
        O.build()
        self.torch.cuda.synchronize()
        k = self.last_ctx
        host = self.host_batches[self.last_in]
        ext, m = self.exts[k], self.matchers[k]
        B = self.B
        slots = ([0, 1] + [B - 2, B - 1] + list(range(2, B - 2)))[:max(2, n_images)]
        oe = O.OracleExtractor(self.nfeat, 1.2, 8, 20, 7)
        sf = oe.scale_factors
        bad, ref = 0, {}
        for i in slots:
            ok, od = oe(host[i])
            gk, gd = ext.batch_fetch(i)
            ref[i] = (ok, od)
            bad += int(not (gk.tobytes() == ok.tobytes() and gd.tobytes() == od.tobytes()))
        pairs = 0
        if self.matcher == "stereo":
            for pr in [s_ // 2 for s_ in slots[::2]][:n_pairs]:
                (kl, dl), (kr, dr) = ref[2 * pr], ref[2 * pr + 1]
                o = O.stereo_match(kl, dl, kr, dr, sf, self.h, BF, BF / FX, 0.0)
                g = m.stereo_fetch(pr, len(kl))
                bad += int(not (g[0] == o[0] and all(a.tobytes() == b.tobytes() for a, b in zip(g[1:], o[1:]))))
                pairs += 1
        elif self.matcher == "project":
            for f in slots[:n_pairs]:
                kp, desc = ref[f]
                o = O.search_by_projection(kp, desc, None, sf, self.bounds, self.mps[f], self.mpd, 3.0, 0.8, None)
                nm, out_mp, out_sc = m.projection_fetch(f, len(kp))
                bad += int(not (nm == o[0] and (out_mp[:len(kp)] == o[1]).all() and (out_sc[:len(kp)] == o[2]).all()))
                pairs += 1
        if os.environ.get("GFO_BENCH_INJECT_MISMATCH"):      # tests only: the self-policing exit must be reachable
            bad += 1
        return {"images": len(slots), "pairs": pairs, "mismatches": bad, "context": k, "slots": slots,
                "against": "oracle/orb_oracle.c on the same host images, bit for bit; the last step of the timed region, outside it"}

    def profile(self, nsteps):
        """per-kernel device time from HIP events on the launch stream, context 0 alone (clean per-kernel durations)"""
        ext = self.exts[0]
        # the chip lowers its clocks while idle and takes ~15 launches to come back (k_fast: 186 us right after a
        # synchronisation, 167 us in a running stream -- rocprofv3 shows the same ramp): warm up first and switch the
        # per-kernel events on without a synchronisation in between
        for _ in range(30):
            self.step(ctx=0)
        ext.profile_enable(True)
        for _ in range(max(1, nsteps)):
            self.step(ctx=0)
        self.torch.cuda.synchronize()
        prof = ext.profile_read()
        ext.profile_enable(False)
        return prof

    def roofline(self, prof, nsteps, value_per_gpu, traffic_workload, live=None, live_note=None):
        counts = self.exts[0].batch_counts(self.B)
        n_kp_img = float(counts.mean())
        sizes = level_sizes(self.w, self.h, self.exts[0].GetInverseScaleFactors())
        per_img, survey_total = algorithmic_bytes(self.w, self.h, sizes, n_kp_img, self.matcher)
        stage_ms = {k: v[0] / max(1, nsteps) for k, v in prof.items() if v[1] > 0}
        dom = max(stage_ms, key=stage_ms.get)
        dom_ms_total, dom_launches = prof[dom]
        avg_launch_ms = dom_ms_total / dom_launches
        launches_per_step = dom_launches / max(1, nsteps)
        bytes_per_launch = per_img[dom] * self.B / launches_per_step
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        # HBM traffic per step and stage: measured in this run (live_traffic) when the profiler was available, else the
        # committed summary of an earlier run of the same command, labelled as such
        traffic, traffic_source, tj_all, raw_all = None, None, {}, {}
        if live:
            tj_all = {k: v["bytes"] for k, v in live.items()}
            raw_all = {k: {"fetch_raw": v["fetch_raw"], "write": v["write"]} for k, v in live.items()}
            traffic_source = live_note
        else:
            for cand in (f"traffic_{traffic_workload}_latest.json", "traffic_latest.json"):
                try:
                    tj = json.load(open(os.path.join(ROOT, "profiles", cand)))
                    if tj.get("workload") == traffic_workload and tj.get("batch") == self.B:
                        x2, rw = tj.get("hbm_bytes_per_launch_fetch_x2", {}), tj.get("hbm_bytes_per_launch", {})
                        tj_all = {k: (x2.get(k, v) if k in WIDE_READ_STAGES else v) for k, v in rw.items()}
                        traffic_source = (f"profiles/{cand} = tag {tj.get('tag')} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of "
                                          f"this command; not measured in this run: {live_note or 'live measurement off'})")
                        break
                except Exception:
                    pass
        if dom in tj_all:
            traffic = int(tj_all[dom] / max(launches_per_step, 1e-9))
        # The roofline that BINDS.  These kernels are integer / byte work whose wall is vector-instruction issue, not HBM: a
        # wave-instruction occupies its SIMD for ISSUE_CYCLES cycles (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost': 4;
        # tools/c/valu_rate.hip measured 4.2 for the packed-integer mix these kernels are made of) and the chip has 1024 SIMDs at
        # <= 2.4 GHz, so issue_frac = SQ_INSTS_VALU x 4 / (1024 x 2.4e9 x duration).  SQ_INSTS_VALU per stage and step comes from
        # the committed counter pass of this command (profiles/sq_counters_latest.json, tools/pmc_sq.sh) -- instruction counts
        # do not vary from run to run; the duration is this run's.
        issue_all, issue_src = {}, None
        for cand in (f"sq_counters_{traffic_workload}_latest.json", "sq_counters_latest.json"):
            try:
                sj = json.load(open(os.path.join(ROOT, "profiles", cand)))
                if sj.get("workload") == traffic_workload and sj.get("batch") == self.B:
                    issue_all = {k: v.get("SQ_INSTS_VALU") for k, v in sj["per_stage_per_step"].items()}
                    issue_src = (f"profiles/{cand} = sq_counters_{sj.get('tag')} (rocprofv3 --pmc SQ_INSTS_VALU pass of this command, "
                                 "tools/pmc_sq.sh); duration measured in this run")
                    break
            except Exception:
                pass

        def issue_frac(stage, ms_per_step):
            n = issue_all.get(stage)
            return round(n * ISSUE_CYCLES / (N_SIMDS * CLOCK_HZ * ms_per_step * 1e-3), 4) if n and ms_per_step > 0 else None
        # Round 6: the flat 4 cycles replaced by what THIS kernel's instructions cost.  tools/opcode_histogram.py disassembles the
        # gfx950 code object of every kernel, takes the opcode mix of its loops and prices it with the measured per-opcode issue rates
        # (profiles/valu_rate_r02.txt: plain 32-bit logic / add / move ~2.4 cycles, everything packed, min / max, compares, v_perm,
        # dot products ~4.2): a MEAN cost per vector instruction, multiplied here with the dynamic SQ_INSTS_VALU.  Two more pipes
        # from the same counter pass: valu_busy (SQ_ACTIVE_INST_VALU, quad-cycles the vector ALUs were executing: the hardware's own
        # figure) and the LDS pipe (SQ_LDS_IDX_ACTIVE cycles per CU, of which SQ_LDS_BANK_CONFLICT are replays).
        mix, mix_src = {}, None
        try:
            mj = json.load(open(os.path.join(ROOT, "profiles", "opcode_mix_latest.json")))
            mix = {k: v.get("mean_cycles_per_valu") for k, v in mj["kernels"].items()}
            mix_src = "profiles/opcode_mix_latest.json (tools/opcode_histogram.py: opcode mix of the kernels' loops x profiles/valu_rate_r02.txt)"
        except Exception:
            pass
        MIX_OF = {"stereo_match": "stereo_match"}
        sq_all = {}
        try:
            sq_all = sj["per_stage_per_step"] if issue_all else {}
        except Exception:
            sq_all = {}

        def pipes(stage, ms_per_step):
            n, c = issue_all.get(stage), mix.get(MIX_OF.get(stage, stage))
            sec = ms_per_step * 1e-3
            q = sq_all.get(stage, {})
            out = {"issue_frac_weighted": round(n * c / (N_SIMDS * CLOCK_HZ * sec), 4) if n and c and sec > 0 else None,
                   "mean_cycles_per_valu": c,
                   "valu_busy_frac": round(q["SQ_ACTIVE_INST_VALU"] * 4 / (N_SIMDS * CLOCK_HZ * sec), 4) if q.get("SQ_ACTIVE_INST_VALU") and sec > 0 else None,
                   "lds_busy_frac": round(q["SQ_LDS_IDX_ACTIVE"] / (N_SIMDS / 4 * CLOCK_HZ * sec), 4) if q.get("SQ_LDS_IDX_ACTIVE") and sec > 0 else None,
                   "lds_conflict_frac": round(q["SQ_LDS_BANK_CONFLICT"] / (N_SIMDS / 4 * CLOCK_HZ * sec), 4) if q.get("SQ_LDS_BANK_CONFLICT") and sec > 0 else None}
            return out
        per_stage = {}
        for k, ms in stage_ms.items():
            lps = prof[k][1] / max(1, nsteps)
            bpl = per_img[k] * self.B / max(lps, 1e-9)
            al = prof[k][0] / prof[k][1]
            per_stage[k] = {"avg_launch_ms": round(al, 4), "launches_per_step": round(lps, 2), "algorithmic_bytes_per_launch": int(bpl),
                            "achieved_GBps": round(bpl / (al * 1e-3) / 1e9, 1), "frac": round(bpl / (al * 1e-3) / 8e12, 5),
                            "issue_frac": issue_frac(k, ms),
                            "traffic_per_step": tj_all.get(k), "traffic_counters_per_step": raw_all.get(k)}
            per_stage[k].update(pipes(k, ms))
        dom_issue = issue_frac(dom, stage_ms[dom])
        dom_pipes = pipes(dom, stage_ms[dom])
        hbm_frac = round(achieved / 8000.0, 5)
        def bound_of(v, hbm):
            """the pipe with the largest utilisation among HBM (the contract's fraction), vector issue (the weighted model; the flat
            one where no opcode mix exists) and the LDS pipe; None without an instruction count: no claim about which wall binds"""
            issue = v.get("issue_frac_weighted") if v.get("issue_frac_weighted") is not None else v.get("issue_frac")
            if issue is None:
                return None
            cand = {"hbm": hbm, "valu-issue": issue, "lds": v.get("lds_busy_frac") or 0.0}
            return max(cand, key=cand.get)
        for k, v in per_stage.items():
            # HBM utilisation for the purpose of naming the wall: the bytes that really crossed the memory controllers where the PMC
            # passes measured them (k_orient_desc's overlapping windows are served by L2: 0.55 x its algorithmic bytes), else `frac`
            t = v.get("traffic_per_step")
            v["hbm_traffic_frac"] = round(t / (stage_ms[k] * 1e-3) / 8e12, 5) if t and stage_ms.get(k, 0) > 0 else None
            v["bound"] = bound_of(v, v["hbm_traffic_frac"] if v["hbm_traffic_frac"] is not None else v["frac"])
        return n_kp_img, {
            # `bound` names the wall the dominant kernel actually sits against: whichever of the two fractions is larger.
            # achieved / peak / frac stay the HBM figures the contract defines; issue_frac is the other roofline.
            "bound": bound_of(dict(dom_pipes, issue_frac=dom_issue), hbm_frac), "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
            "frac": hbm_frac, "issue_frac": dom_issue, "issue_frac_weighted": dom_pipes["issue_frac_weighted"],
            "valu_busy_frac": dom_pipes["valu_busy_frac"], "lds_busy_frac": dom_pipes["lds_busy_frac"], "lds_conflict_frac": dom_pipes["lds_conflict_frac"],
            "issue_model": {"valu_instructions_per_launch": int(issue_all[dom] / max(launches_per_step, 1e-9)) if issue_all.get(dom) else None,
                            "cycles_per_wave_instruction_flat": ISSUE_CYCLES, "cycles_per_wave_instruction_weighted": dom_pipes["mean_cycles_per_valu"],
                            "simds": N_SIMDS, "clock_ghz": CLOCK_HZ / 1e9, "source": issue_src, "opcode_mix": mix_src,
                            "bound_rule": "largest of HBM (measured traffic where a PMC pass gave it, else frac), issue_frac_weighted and lds_busy_frac"},
            "traffic": traffic, "traffic_measured": bool(live), "traffic_source": traffic_source, "traffic_pass_killed": LIVE_TRAFFIC_KILLED,
            "avg_launch_ms": round(avg_launch_ms, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch),
            "pipeline_frac_hbm": round(value_per_gpu * survey_total / 8e12, 5),
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
            "all_kernels": per_stage}

    def close(self):
        for e in self.exts:
            e.close()
        self.d_inputs = None
        self.d_mps = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=HEADLINE_BATCH, help="images per GPU per step (even); 256 is where the rate levels off on MI355X (64 / 128 / 256 / 384 / 1024: 237k / 260k / 274k / 275k / 275k frames/s, profiles/batch_sweep_r03.txt)")
    ap.add_argument("--workload", default="stereo752", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short passes over the other BASELINE configs")
    ap.add_argument("--profile-steps", type=int, default=10)
    ap.add_argument("--streams", type=int, default=0,
                    help="independent contexts (arena + HIP stream) the steps alternate between, so the tail of one "
                         "batch overlaps the head of the next (0 = the workload's measured best, CONTEXTS)")
    ap.add_argument("--live-traffic", action="store_true", help=argparse.SUPPRESS)   # (the default since round 5)
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run.  Default (N = 1, rocprofv3 present, not under a profiler): two "
                         "rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) of this command, started before this process touches the "
                         "GPU, 75 s bound each (+10..15 s); if they cannot run the committed profiles/traffic_latest.json is used and labelled")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): every GPU its own camera stream, --steps batches EACH; strong: ONE stream of --steps batches dealt "
                         "round robin over the GPUs (sharding.shard_round_robin), value = its frames / the slowest rank's time.  A weak "
                         "run on N > 1 GPUs carries a short strong pass beside it (strong_scaling)")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1 GPUs: all-gather the keypoint counts every this many steps (1 = every step, as north_star names it; "
                         "0 = never -- separates straggler coupling between ranks from kernel time on a real node)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the last timed step (4 images + 2 matcher units)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the per-frame boundary harness (per_frame_boundary)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # internal: the pass live_traffic() profiles
    ap.add_argument("--spawn-selftest", action="store_true", help=argparse.SUPPRESS)   # tests: ranks report their environment and exit (no GPU)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and args.gpus > 1:
        # no launcher around this command: be the launcher (before torch is imported, before anything touches the GPU)
        argv = [a for a in sys.argv[1:] if a != "--spawn-selftest"]
        sys.exit(spawn_ranks(args.gpus, argv, selftest=args.spawn_selftest))
    if args.spawn_selftest:
        info = {"spawn_selftest": True, "rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                "world": int(os.environ.get("WORLD_SIZE", "1")), "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}",
                "gpus_arg": args.gpus, "torch_loaded": "torch" in sys.modules}
        if os.environ.get("GFO_BENCH_SELFTEST_FAIL_RANK") == os.environ.get("RANK", "0"):
            sys.exit(7)
        if info["rank"] == 0:
            print(json.dumps(info), flush=True)
        else:
            print(json.dumps(info), file=sys.stderr, flush=True)
        return
    if os.environ.get("GFO_BENCH_WATCHDOG"):      # diagnosis of a stalled run: dump every thread's stack after N seconds and exit
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["GFO_BENCH_WATCHDOG"]), exit=True)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # Multi-process GPU work on this pool: the image exports HSA_ENABLE_IPC_MODE_LEGACY=0 here and on the GPU boxes, and its
    # environment notes say why -- the host driver only supports dmabuf IPC; without it RCCL across processes fails with
    # `hipIpcGetMemHandle: invalid argument`.  Kept (never overridden) in EVERY rank, before the HIP runtime loads, so that the
    # spawned form (`bench.py --gpus N`) and the torch.distributed.run form are one and the same on this point.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment: running {world} ranks (n_gpus reports {world})", file=sys.stderr)
    # the PMC passes run first, as children, while this process has not touched the GPU yet
    live, live_note = None, "live measurement off (--no-live-traffic)" if args.no_live_traffic else "N > 1: the counter passes are a single-GPU measurement"
    if world == 1 and not args.no_live_traffic and not args.pmc_child and "GFO_BENCH_SPAWNED" not in os.environ:
        live, live_note = live_traffic(args.workload, args.batch - (args.batch & 1))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # rehearsal of the N > 1 control flow on a one-GPU box (tools/rehearse_multirank.sh): every rank on device 0 and
    # gloo instead of RCCL (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("GFO_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import gf_orb_slam2_amd as G
    assert G.HEADLINE_BATCH == HEADLINE_BATCH, "bench.py and the package disagree on the headline batch size"

    B = args.batch - (args.batch & 1)
    nctx = args.streams if args.streams >= 1 else CONTEXTS[args.workload]   # 0: the workload's default
    strong = args.scaling == "strong"
    job = Job(G, torch, args.workload, B, nctx, local_rank, rank, world, dist, prime=not args.pmc_child, gather_every=args.gather_every,
              one_stream=strong)
    if args.pmc_child:      # only the kernels are wanted (counters are read per dispatch by the profiler around this process)
        job.timed(args.steps, args.warmup)
        job.close()
        return
    if strong:
        dt, my_steps = job.timed_strong(args.steps, args.warmup)     # EXACTLY --steps batches in all, dealt over the ranks
        total_frames = B * args.steps
    else:
        dt = job.timed(args.steps, args.warmup)
        total_frames = world * B * args.steps
    value = total_frames / dt
    verified = job.verify() if rank == 0 and not args.no_verify else None

    # a timed region shorter than half a second says little about a sustained rate: repeat for >= 1 s
    sustained = None
    if dt < 0.5 and strong:
        n_s = int(min(20000 * world, max(args.steps, 1.2 * args.steps / max(dt, 1e-6))))
        dts, _ = job.timed_strong(n_s, 0)
        sustained = {"value": round(B * n_s / dts, 1), "seconds": round(dts, 3), "steps": n_s}
    elif dt < 0.5:
        n_s = int(min(20000, max(args.steps, 1.2 / max(dt / args.steps, 1e-6))))
        dts = job.timed(n_s, 0)
        sustained = {"value": round(world * B * n_s / dts, 1), "seconds": round(dts, 3), "steps": n_s}
    # N > 1, weak headline: the strong-scaling figure beside it -- one stream of world x 40 batches dealt round robin (this rank's
    # resident batches stand in for its share of that stream: same sizes, same kernels), fixed total work
    strong_beside = None
    if world > 1 and not strong:
        n_b = 40 * world
        dtb, mine_b = job.timed_strong(n_b, 2)
        strong_beside = {"scaling": "strong", "value": round(B * n_b / dtb, 1), "unit": "frames/s", "stream_batches": n_b, "batches_this_rank": mine_b,
                         "images_per_batch": B, "seconds": round(dtb, 4), "partition": "batch i -> rank i % N (sharding.shard_round_robin)",
                         "note": "N = 1 value of the same figure is the headline's (one rank takes every batch)"}

    extra = {}
    if world == 1:
        # PCIe-inclusive rate (never `value`): every step first copies its batch from pinned host memory
        pinned = torch.from_numpy(job.host_batches[0]).pin_memory()
        n_h = max(20, min(args.steps, 60))    # (30 steps = 33 ms were too short a region: one hiccup halved the delivered rate once)
        job.chain(False)      # the copies already pace the contexts; chained on top of that they serialise (77k against 111k)
        # (three repetitions, the median: these are 0.1-s regions on a shared PCIe link and one hiccup -- a page of the pinned blocks
        #  touched for the first time, a neighbour's transfer -- used to decide the figure: 94 k next to 137 k on one box)
        dth = sorted(job.timed(n_h, 4 if r == 0 else 1, h2d_from=pinned) for r in range(3))[1]
        job.chain(True)
        extra["value_with_h2d"] = round(B * n_h / dth, 1)
        # ... and the results out: every step also lands counts, keypoints, descriptors (and the stereo outputs) of its batch in
        # pinned host memory, where ORBextractor::operator() leaves them (ORBextractor.cc:1137-1173).  Opposite directions of
        # a full-duplex link: the ceiling stays the input copy
        dtd = sorted(job.timed(n_h, 6 if r == 0 else 1, h2d_from=pinned, deliver=True) for r in range(3))[1]    # (the first deliveries touch the pinned result blocks for the first time)
        lay = job.deliver_layout
        extra["value_delivered"] = round(B * n_h / dtd, 1)
        extra["delivery"] = {"host_bytes_in_per_step": int(B * job.w * job.h), "host_bytes_out_per_step": int(lay.bytes),
                             "pcie_ceiling_frames_per_s": round(56.5e9 / (job.w * job.h), 0),
                             "note": "in: one H2D per step from pinned memory on a copy stream; out: gfo_batch_deliver, D2H on the context's "
                                     "copy stream, full duplex with the next step's H2D; ceiling = 56.5 GB/s measured H2D rate / bytes per frame in",
                             "repetitions": "median of 3 regions of %d steps each, for both figures" % n_h,
                             "ratio_to_value_with_h2d": round((B * n_h / dtd) / (B * n_h / dth), 3)}
        del pinned
        # median of 20 single batches, one context, each batch synchronised
        lat = []
        for _ in range(20):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            job.step(ctx=0)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
        extra["median_of_20"] = {"ms_per_batch": round(float(np.median(lat)) * 1e3, 4), "value": round(B / float(np.median(lat)), 1),
                                 "note": "one context, every batch synchronised (no overlap between batches)"}

    prof = job.profile(args.profile_steps)
    line = None
    if rank == 0:
        n_kp_img, roof = job.roofline(prof, args.profile_steps, value / world, args.workload, live, live_note)
        stereo = job.matcher == "stereo"
        # did the collective see N ranks?  the output of the LAST count all-gather of the timed region, cut into B-sized
        # segments (one per rank, zero-initialised): a rank that did not take part leaves its segment at zero
        coll = None
        if world > 1:
            segs = [int((g.view(world, B) != 0).any(dim=1).sum().item()) for g in job.gathered if g is not None]
            coll = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "gathered_elements": int(job.gathered[0].numel()),
                    "elements_per_rank": B, "ranks_with_counts": min(segs) if args.gather_every else None,
                    "launcher": "bench.py spawn_ranks" if os.environ.get("GFO_BENCH_SPAWNED") else "external (torch.distributed.run)"}
        line = {
            "metric": "frames/sec ORB extract+match, 752x480 @2000 kp" if args.workload == "stereo752" else f"frames/sec ORB {args.workload}",
            "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            # ranks whose counts are in the last all-gather's output: `collective_ranks` whatever the backend, `rccl_ranks` only when
            # the backend IS RCCL ("nccl"); a gloo rehearsal on one GPU reports rccl_ranks null
            "collective_ranks": (coll["ranks_with_counts"] if coll else 1),
            "rccl_ranks": ((coll["ranks_with_counts"] if coll["backend"] == "nccl" else None) if coll else 1), "collective": coll, "verified": verified,
            "config": {"workload": job.cfg_name,
                       "verified": ({k: verified[k] for k in ("images", "pairs", "mismatches")} if verified else None), "frame": "one camera image (a stereo pair = 2 frames + 1 association)",
                       "images_per_step_per_gpu": B, "stereo_pairs_per_s": round(value / 2, 1) if stereo else None,
                       "width": job.w, "height": job.h, "nfeatures": job.nfeat, "levels": 8, "scale_factor": 1.2, "fast_th": [20, 7],
                       "map_points": MAP_POINTS if job.matcher == "project" else None,
                       "mean_keypoints_per_image": round(n_kp_img, 1), "contexts_per_gpu": nctx,
                       "contexts_chained_after_stage": job.chain_stage if job.chained else None,
                       "priming_steps": PRIME_STEPS, "gather_every": args.gather_every if world > 1 else None,
                       "distinct_input_batches": len(job.d_inputs),
                       "sharding": ("single GPU" if world == 1 else
                                    f"ONE stream of {args.steps} batches, batch i -> rank i % {world} (shard_round_robin), all-gather of counts per round" if strong else
                                    f"{world} x independent streams, RCCL all-gather of counts")},
            "roofline": roof,
        }
        if sustained:
            line["sustained"] = sustained
        if strong_beside:
            line["strong_scaling"] = strong_beside
        line.update(extra)
    job.close()
    del job
    torch.cuda.empty_cache()

    # ---- N > 1: BASELINE configs[4] in the same line -- one 1920x1080 @4000 stream per GPU (extract + SearchByProjection against the
    #      50 000-point map), 64 images per rank and step, same ranks, same all-gather of the keypoint counts
    if world > 1 and not args.no_other_configs and args.workload != "proj1080":
        b4 = min(64, B)
        job4 = Job(G, torch, "proj1080", b4, CONTEXTS["proj1080"], local_rank, rank, world, dist, gather_every=args.gather_every)
        s4 = max(10, min(60, args.steps))
        dt4 = job4.timed(s4, 5)
        ver4 = job4.verify(n_images=2, n_pairs=1) if rank == 0 and not args.no_verify else None
        if rank == 0:
            segs = [int((g.view(world, b4) != 0).any(dim=1).sum().item()) for g in job4.gathered if g is not None]
            ranks4 = min(segs) if args.gather_every else None
            line["other_configs"] = [{"workload": "configs[4]: " + job4.cfg_name + f", one stream per GPU x {world}", "name": "proj1080",
                                      "value": round(world * b4 * s4 / dt4, 1), "unit": "frames/s", "n_gpus": world, "images_per_step_per_gpu": b4,
                                      "steps": s4, "ms_per_step": round(dt4 / s4 * 1e3, 4), "scaling": "weak", "collective_ranks": ranks4,
                                      "rccl_ranks": ranks4 if dist.get_backend() == "nccl" else None,
                                      "verified": ({k: ver4[k] for k in ("images", "pairs", "mismatches")} if ver4 else None)}]
            if ver4 and ver4["mismatches"]:
                verified = dict(verified or {"images": 0, "pairs": 0, "mismatches": 0})
                verified["mismatches"] += ver4["mismatches"]
        job4.close()
        del job4
        torch.cuda.empty_cache()

    # ---- the other BASELINE configs, short passes, same line ----
    if world == 1 and not args.no_other_configs and under_profiler():
        line["other_configs"] = "skipped: this run is under a profiler (they run as child processes of this command)"
    elif world == 1 and not args.no_other_configs:
        others = []
        for name in ("extract752", "extract1080", "proj1080", "stereo752"):
            if name == args.workload:
                continue
            ob = 64 if WORKLOADS[name][0] > 1000 else B
            others.append(side_config(name, ob, args.streams))
        line["other_configs"] = others
        line["real_image"] = real_image()
        side_bad = sum((o.get("verified") or {}).get("mismatches", 0) for o in others)
        if side_bad:
            verified = dict(verified or {"images": 0, "pairs": 0, "mismatches": 0})
            verified["mismatches"] += side_bad

    if rank == 0 and world == 1 and not args.no_boundary and not under_profiler():
        line["per_frame_boundary"] = per_frame_boundary()
        line["matcher_calls"] = matcher_calls()
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            w, h, nfeat, matcher, _ = WORKLOADS[args.workload]
            line["cpu_baseline"] = cpu_baseline(w, h, nfeat, matcher == "stereo")
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # self-policing: a run whose checked results differ from the oracle's has printed its line (the evidence) and FAILS
    if rank == 0 and verified and verified["mismatches"] > 0:
        print(f"bench.py: {verified['mismatches']} of the verified results differ from the oracle: the measurement is void", file=sys.stderr, flush=True)
        sys.exit(4)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker(sys.argv[2:])
    else:
        main()
