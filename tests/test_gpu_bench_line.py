"""bench.py polices its own line: the oracle check of the last timed step is part of `config` (what the
driver's record keeps) and a mismatch fails the command AFTER the line -- the evidence -- has been printed."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CMD = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8", "--profile-steps", "1",
       "--no-other-configs", "--no-cpu-baseline", "--no-boundary", "--no-live-traffic"]


def _run(extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    r = subprocess.run(CMD, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout[-1000:], r.stderr[-2000:])
    return r, json.loads(lines[0])


def test_verified_is_in_config_and_a_clean_run_passes():
    r, d = _run({})
    assert r.returncode == 0, r.stderr[-2000:]
    assert d["config"]["verified"] == {"images": 4, "pairs": 2, "mismatches": 0}
    assert d["roofline"]["bound"] in ("valu-issue", "hbm", None)
    # no instruction count is committed for a batch of 8: no claim about the binding wall (never "hbm" by default)
    assert d["roofline"]["issue_frac"] is None and d["roofline"]["bound"] is None
    assert all(v["bound"] is None for v in d["roofline"]["all_kernels"].values())


def test_a_mismatch_prints_the_line_and_fails_the_command():
    r, d = _run({"GFO_BENCH_INJECT_MISMATCH": "1"})
    assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
    assert d["config"]["verified"]["mismatches"] == 1 and d["value"] > 0
    assert "differ from the oracle" in r.stderr


def test_matcher_calls_block_of_the_line():
    """`matcher_calls` of the default line = tools/matcher_call_latency.py with JSON=1 (no oracle inside: it runs under bench.py): fourteen
    calls, each with a time and matches found."""
    env = {k: v for k, v in os.environ.items() if k not in ("TH", "ONLY", "GFO_PROJ_STATS")}
    env["JSON"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "matcher_call_latency.py")], capture_output=True, text=True, env=env, cwd=ROOT,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    names = [c["call"] for c in d["calls"]]
    assert names == ["SearchByProjection(F, MapPoints)"] * 3 + ["SearchByProjection_Budget (gfo_search_by_projection_points)",
                     "GetCandidates for every map point (gfo_projection_candidates)",
                     "ComputeStereoMatches(host arrays)", "SearchByProjection(Cur, Last)", "SearchForInitialization(F1, F2)", "ComputeBoW",
                     "SearchByBoW(KF, F)", "SearchForTriangulation(KF1, KF2)", "ComputeBoW (vocabulary of ORBvoc's size)",
                     "SearchByBoW(KF, F) (vocabulary of ORBvoc's size)", "SearchForTriangulation(KF1, KF2) (vocabulary of ORBvoc's size)"]
    big = d["calls"][-3]
    assert big["nodes"] == 1111111 and big["L"] == 6 and 0 < big["vocabulary_upload_ms"] < 2000
    assert big["ms"] < 3 * [c for c in d["calls"] if c["call"] == "ComputeBoW"][0]["ms"] + 0.1            # a call does not grow with the vocabulary (the descent reads 6 x 10 centres)
    assert all(0 < c["ms"] < 50 for c in d["calls"])
    assert all(c.get("matches", c.get("words", c.get("entries"))) > 100 for c in d["calls"])
    src = open(os.path.join(ROOT, "tools", "matcher_call_latency.py")).read()
    assert "import orb_oracle" not in src and "from oracle" not in src
