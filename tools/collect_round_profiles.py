#!/usr/bin/env python3
"""Copies what tools/round_profiles.sh <tag> left under gpurun_out/ into profiles/ (run here, after the gpurun call):
kernel stats + PMC traffic (tools/summarize_profile.py on the newest CSVs only), SQ counters, boundary throughput (three runs
merged into one file), latency pair, boundary trace, default bench line.  usage: tools/collect_round_profiles.py <tag> [batch]"""
import glob, json, os, shutil, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]; batch = sys.argv[2] if len(sys.argv) > 2 else "256"
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
# gpurun merges into existing directories: keep only the newest run of each profiler output directory
for sub in ("trace", "pmc_fetch", "pmc_write"):
    runs = sorted(glob.glob(os.path.join(G, f"prof_{tag}", sub, "*", "*_agent_info.csv")), key=os.path.getmtime)
    for old in runs[:-1]:
        stem = old[:-len("agent_info.csv")]
        for f in glob.glob(stem + "*"):
            os.remove(f)
subprocess.check_call([sys.executable, os.path.join(R, "tools", "summarize_profile.py"), os.path.join(G, f"prof_{tag}"), tag],
                      env=dict(os.environ, GFO_PROF_BATCH=batch), stdout=subprocess.DEVNULL)
shutil.copy(os.path.join(P, f"traffic_{tag}.json"), os.path.join(P, "traffic_latest.json"))
tmpl = os.path.join(P, f"boundary_throughput_{tag}.json")
if not os.path.exists(tmpl):      # a new round: the descriptions of the three runs come from the newest earlier file
    tmpl = sorted(glob.glob(os.path.join(P, "boundary_throughput_r*.json")))[-1]
old = json.load(open(tmpl))
runs = {}
for k in ("pageable", "pinned", "nocombine"):
    j = json.load(open(os.path.join(G, f"boundary_throughput_{tag}_{k}.json")))
    runs[k] = {"what": old["runs"][k]["what"], "points": j["points"]}
json.dump({"workload": old["workload"], "runs": runs}, open(os.path.join(P, f"boundary_throughput_{tag}.json"), "w"), indent=1)
for src, dst in ((f"bench_default_{tag}.json", f"bench_default_{tag}.json"), (f"latency_pair_{tag}.json", f"latency_pair_{tag}.json"),
                 (f"sq_counters_{tag}.txt", f"sq_counters_{tag}.txt"), (f"trace_boundary_{tag}_k8_combined.txt", f"boundary_trace_{tag}_k8_combined.txt")):
    shutil.copy(os.path.join(G, src), os.path.join(P, dst))
# the per-stage, per-step SQ counters bench.py's issue / LDS model reads (tools/pmc_sq.sh)
sq = os.path.join(G, "pmc_sq", "sq_per_step.json")
if os.path.exists(sq):
    j = json.load(open(sq))
    j["tag"] = tag
    for name in (f"sq_counters_{tag}.json", "sq_counters_latest.json"):
        json.dump(j, open(os.path.join(P, name), "w"), indent=1)
print("profiles/", tag, "refreshed")
