"""bench.py's N > 1 control flow, rehearsed on the one GPU a test box has: two ranks share device 0 and talk over gloo
(GFO_BENCH_REHEARSAL=1; RCCL refuses two ranks on one device).  What it checks is what the driver's multi-GPU run
relies on: the rendezvous on 127.0.0.1, one context set per rank, the count all-gather in every step, the barriers,
max-over-ranks timing and rank 0 printing exactly one JSON line with the whole-job aggregate."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, GFO_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "16",
           "--profile-steps", "2", "--no-other-configs"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6 and d["warmup"] == 2
    assert d["value"] > 0 and d["unit"] == "frames/s"
    # whole-job aggregate: 2 ranks x 16 images x 6 steps over the max-over-ranks time
    assert abs(d["value"] - 2 * 16 * 6 / (d["ms_per_step"] * 6 * 1e-3)) / d["value"] < 0.01
    assert "roofline" in d and "cpu_baseline" not in d and "other_configs" not in d
    assert d["collective_ranks"] == 2 and d["collective"]["launcher"].startswith("external")
    assert d["collective"]["backend"] == "gloo" and d["rccl_ranks"] is None      # a gloo rehearsal does not claim RCCL ranks
    assert d["config"]["verified"] == {"images": 4, "pairs": 2, "mismatches": 0}
    # the strong-scaling figure beside the weak line: one stream of 2 x 40 batches, batch i -> rank i % 2, fixed total work
    ss = d["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["stream_batches"] == 80 and ss["batches_this_rank"] == 40 and ss["images_per_batch"] == 16
    assert abs(ss["value"] - 16 * 80 / ss["seconds"]) / ss["value"] < 0.01


def test_strong_scaling_deals_one_stream_over_the_ranks():
    """`--scaling strong` (single-stream scaling, round-robin batches of frames across ranks): --steps is the length of
    ONE stream, batch i goes to rank i % N (sharding.shard_round_robin), value = the stream's frames over the slowest rank's time.
    7 batches over 2 ranks: the last round is partial -- rank 1 has no batch in it and still takes part in the count exchange (a
    rank that skipped it would hang the other one: the command would time out)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["GFO_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scaling", "strong", "--steps", "7", "--warmup", "2", "--batch", "16",
           "--profile-steps", "2", "--no-other-configs"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 7
    # the WHOLE job is 7 batches of 16 images, whatever the number of ranks
    assert abs(d["value"] - 16 * 7 / (d["ms_per_step"] * 7 * 1e-3)) / d["value"] < 0.01
    assert d["collective_ranks"] == 2 and "shard_round_robin" in d["config"]["sharding"]
    assert d["config"]["verified"]["mismatches"] == 0 and "strong_scaling" not in d
    # one rank alone: the same command form, the whole stream on rank 0
    cmd1 = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--scaling", "strong", "--steps", "7", "--warmup", "2", "--batch", "16",
            "--profile-steps", "2", "--no-other-configs", "--no-live-traffic", "--no-boundary", "--no-cpu-baseline"]
    r1 = subprocess.run(cmd1, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r1.returncode == 0, r1.stderr[-3000:]
    d1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["scaling"] == "strong" and abs(d1["value"] - 16 * 7 / (d1["ms_per_step"] * 7 * 1e-3)) / d1["value"] < 0.01


def test_plain_gpus_2_launches_two_ranks_by_itself():
    """The driver's recorded command form, `python3 bench.py --gpus N ...` with no launcher: bench.py starts the N ranks
    itself (spawn_ranks) and the count all-gather of the timed region holds counts of N ranks.  Without --no-other-configs
    the N > 1 line also carries BASELINE configs[4] (one 1920x1080 @4000 stream per GPU, extract + SearchByProjection against the
    50 000-point map, same ranks, same all-gather): here 4 images per rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["GFO_BENCH_REHEARSAL"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "4", "--profile-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=400)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 == d["collective_ranks"], (d["n_gpus"], d["collective_ranks"])
    assert d["collective"]["world_size"] == 2 and d["collective"]["gathered_elements"] == 2 * 4
    assert d["collective"]["launcher"] == "bench.py spawn_ranks"
    assert abs(d["value"] - 2 * 4 * 6 / (d["ms_per_step"] * 6 * 1e-3)) / d["value"] < 0.01
    (c4,) = d["other_configs"]
    assert c4["name"] == "proj1080" and c4["workload"].startswith("configs[4]") and c4["n_gpus"] == 2 and c4["images_per_step_per_gpu"] == 4
    assert c4["collective_ranks"] == 2 and c4["value"] > 0 and c4["verified"]["mismatches"] == 0 and c4["verified"]["pairs"] == 1
    assert abs(c4["value"] - 2 * 4 * c4["steps"] / (c4["ms_per_step"] * c4["steps"] * 1e-3)) / c4["value"] < 0.01
