"""CPU tests of the host-side pieces: import shim, synthetic streams, bench accounting, sharding
(including a real 2-process gloo run of the count all-gather)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_shim_exposes_reference_names():
    import gf_orb_slam2_amd as G
    for name in ("ORBextractor", "ORBmatcher", "StereoParams", "FrameBounds"):
        assert hasattr(G, name)
    assert G.ORBmatcher.TH_HIGH == 100 and G.ORBmatcher.TH_LOW == 50 and G.ORBmatcher.HISTO_LENGTH == 30
    for m in ("GetLevels", "GetScaleFactor", "GetScaleFactors", "GetInverseScaleFactors", "GetScaleSigmaSquares",
              "GetInverseScaleSigmaSquares", "ComputePyramid", "__call__"):
        assert hasattr(G.ORBextractor, m)


def test_synthetic_stream_is_seeded_and_textured():
    from gf_orb_slam2_amd.synth import synth_frame, synth_stereo_pair
    a, b = synth_frame(752, 480, 3), synth_frame(752, 480, 3)
    assert a.dtype == np.uint8 and a.shape == (480, 752) and (a == b).all()
    assert (synth_frame(752, 480, 4) != a).any()
    assert a.std() > 20
    l, r = synth_stereo_pair(752, 480, 0)
    assert (l == synth_frame(752, 480, 0)).all() and r.shape == l.shape and (r != l).any()


def test_bench_byte_accounting_matches_survey():
    sys.path.insert(0, ROOT)
    import bench
    inv = np.float32(1.0) / np.cumprod(np.concatenate([[np.float32(1.0)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    sizes = bench.level_sizes(752, 480, inv)
    assert sum(a * b for a, b in sizes) == 1117367
    per, total = bench.algorithmic_bytes(752, 480, sizes, 2000, None)
    assert total == 9186428
    assert per["blur"] == 2 * 1117367 and per["fast"] == 1117367 and per["orient_desc"] == 2000 * 2178
    _, total_st = bench.algorithmic_bytes(752, 480, sizes, 2000, "stereo")
    assert total_st == 9186428 + 256000 / 2              # half a pair's association per image
    sizes_b = bench.level_sizes(1920, 1080, inv)
    assert bench.algorithmic_bytes(1920, 1080, sizes_b, 4000, None)[1] == 36462884
    per_p, total_p = bench.algorithmic_bytes(1920, 1080, sizes_b, 4000, "project", 50000)
    assert per_p["project"] == 3440000 and total_p == 36462884 + 3440000


def test_synthetic_stream_and_local_map():
    """input S3 as a stream: one scene, shifted crops; the map imitates frame 0's keypoints"""
    from gf_orb_slam2_amd.synth import MAP_POINT_DTYPE, synth_local_map, synth_stream
    import gf_orb_slam2_amd as G
    assert MAP_POINT_DTYPE == G.MAP_POINT_DTYPE
    frames, offs = synth_stream(160, 120, 3, idx=1, max_shift=8)
    assert len(frames) == 3 and frames[0].shape == (120, 160) and offs[0] == (8, 8)
    ox, oy = offs[1]
    a, b = frames[0][20:100, 20:140].astype(int), frames[1][20 + 8 - oy:100 + 8 - oy, 20 + 8 - ox:140 + 8 - ox].astype(int)
    assert np.abs(a - b).max() <= 4                      # same scene, +-2 grey levels of noise each
    kp = np.zeros(50, G.KEYPOINT_DTYPE)
    kp["x"] = np.linspace(20, 140, 50); kp["y"] = 60; kp["octave"] = np.arange(50) % 8
    desc = np.random.default_rng(0).integers(0, 256, (50, 32), dtype=np.uint8)
    mpd, mps = synth_local_map(kp, desc, offs, 160, 120, m=500, n_vis=40, seed=7)
    assert mpd.shape == (500, 32) and mps.shape == (3, 500) and (mps["flags"] == 5).all()
    d = np.unpackbits(mpd[:, None, :] ^ desc[None, :, :], axis=2).sum(2).min(1)
    assert (d <= 60).sum() >= 40                         # the visible subset stays within 60 flipped bits of a keypoint


def test_sharding_partitions():
    from gf_orb_slam2_amd.sharding import shard_pairs, shard_round_robin
    world, per = 8, 64
    allp = [p for r in range(world) for p in shard_pairs(r, world, per)]
    assert allp == list(range(world * per))
    rr = sorted(i for r in range(3) for i in shard_round_robin(10, r, 3))
    assert rr == list(range(10))
    with pytest.raises(ValueError):
        shard_pairs(8, 8, 4)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gf_orb_slam2_amd.sharding import gather_counts, shard_pairs
    pairs = list(shard_pairs(rank, world, 3))
    counts = torch.tensor([2000 + 10 * p + s for p in pairs for s in (0, 1)], dtype=torch.int32)
    got = gather_counts(counts, world, dist)
    # timing reduction as bench.py does it: MAX over ranks
    t = torch.tensor([0.5 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, got.tolist(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_count_all_gather_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [2000 + 10 * p + s for p in range(6) for s in (0, 1)]
    for rank, got, tmax in res:
        assert got == expect
        assert tmax == 1.5


def test_matcher_on_closed_extractor_raises():
    """ORBmatcher / ORBVocabulary borrow the extractor's context: once it is closed every call must raise GfoError(-5)
    on the host side instead of handing a freed gfo_ctx to the library (no GPU needed: the handle is null)."""
    import ctypes
    import pytest
    import gf_orb_slam2_amd as G

    class Closed:
        handle = ctypes.c_void_p()

    m = G.ORBmatcher(0.8, True, extractor=Closed())
    for call in (lambda: m.stereo_match_batch(G.StereoParams(480, 1.0, 1.0, 0.0)),
                 lambda: m.map_upload(np.zeros((4, 32), np.uint8)),
                 lambda: m.projection_fetch(0, 4)):
        with pytest.raises(G.GfoError) as e:
            call()
        assert e.value.code == -5


def test_bench_reads_the_pmc_csv_per_stage_and_step(tmp_path):
    """bench.py's reader of rocprofv3's counter_collection.csv: KiB per dispatch -> bytes per stage and step (a step is one
    k_fast launch; the pyramid and the quadtree are several launches per step; the SAD matcher is not the stereo stage)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
    for step in range(3):
        rows += [f"{10 * step},k_resize(GfoGeom const*),FETCH_SIZE,100", f"{10 * step + 1},k_resize(GfoGeom const*),FETCH_SIZE,50",
                 f"{10 * step + 2},k_resize_tail(x),FETCH_SIZE,10", f"{10 * step + 3},\"void k_fast<48, 44, true>(x)\",FETCH_SIZE,64",
                 f"{10 * step + 4},k_quadtree(x),FETCH_SIZE,3", f"{10 * step + 8},k_quadtree(x),FETCH_SIZE,1", f"{10 * step + 5},k_stereo_match_sad(x),FETCH_SIZE,999",
                 f"{10 * step + 6},k_stereo_match(x),FETCH_SIZE,8", f"{10 * step + 7},k_fast(x),WRITE_SIZE,7"]
    p = tmp_path / "counter_collection.csv"
    p.write_text("\n".join(rows) + "\n")
    per, steps = bench.pmc_bytes_per_step(str(p), "FETCH_SIZE")
    assert steps == 3
    assert per["resize"] == 160 * 1024 and per["fast"] == 64 * 1024 and per["quadtree"] == 4 * 1024 and per["stereo_match"] == 8 * 1024
    per_w, _ = bench.pmc_bytes_per_step(str(p), "WRITE_SIZE")
    assert per_w == {"fast": 7 * 1024}


def test_adapter_context_table_keeps_live_extractors_and_retires_dead_ones():
    """adapter/gfo_context_table.h (what ORBextractor_gfo.cc keeps its contexts in) against a counting stand-in of
    gfo_ctx_create / gfo_ctx_destroy: 32 live extractors called round-robin never lose a context (no re-creation in
    steady state), an extractor re-created at the address of a deleted one
    (Tracking::updateORBExtractor, src/Tracking.cc:298-320) gets a NEW context and the old one is destroyed at once
    (nothing keyed by the pointer may survive), dead objects at other addresses are reclaimed when idle."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ missing")
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "context_table_check")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "gf-orb-slam2_amd", "adapter"),
                        os.path.join(ROOT, "tests", "host", "context_table_check.cc"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_bench_gpus_n_is_its_own_launcher_and_stays_off_the_gpu():
    """`python bench.py --gpus N` with no launcher around it (the driver's recorded command form) starts N ranks itself
    (bench.spawn_ranks): every rank gets RANK / LOCAL_RANK / WORLD_SIZE and one shared MASTER_ADDR:PORT on 127.0.0.1, rank 0's
    single JSON line is relayed, and the launcher process never imports torch (so it cannot have initialised a GPU) nor the
    product nor the oracle.  A failing rank fails the command."""
    import json
    import subprocess
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["GFO_BENCH_PARENT_REPORT"] = "1"
    r = subprocess.run([sys.executable, bench, "--gpus", "3", "--steps", "2", "--spawn-selftest"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(out) == 1 and out[0]["rank"] == 0 and out[0]["world"] == 3 and out[0]["gpus_arg"] == 3
    others = [json.loads(l) for l in r.stderr.splitlines() if l.startswith("{")]
    assert sorted(o["rank"] for o in others) == [1, 2] and all(o["local_rank"] == o["rank"] and o["world"] == 3 for o in others)
    assert {o["master"] for o in others + out} == {out[0]["master"]} and out[0]["master"].startswith("127.0.0.1:")
    assert not any(o["torch_loaded"] for o in others + out)
    parent = [l for l in r.stderr.splitlines() if l.startswith("parent_modules ")]
    assert parent == ["parent_modules []"], parent
    # one rank dies: the command fails and says which
    env["GFO_BENCH_SELFTEST_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, bench, "--gpus", "3", "--spawn-selftest"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "rank 1 exited with code 7" in r.stderr
    # under a launcher (WORLD_SIZE set) the command is a rank, not a launcher
    env.pop("GFO_BENCH_SELFTEST_FAIL_RANK")
    env.update(RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--spawn-selftest"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "" and '"rank": 1' in r.stderr


def test_combiner_under_thread_sanitizer(tmp_path):
    """gf-orb-slam2_amd/csrc/gfo_combine.hip -- engine slots, forming batches, the GfoPair rendezvous with its
    lock-free state mirror, dormancy / wake counts -- compiled UNMODIFIED for the CPU with -fsanitize=thread against a fake backend
    (tests/host/tsan/fake_gfo_internal.h: host memory for the device, 30-200 us per batch, results that are a function of the image
    bytes) and driven by tests/host/combine_tsan.cc: 1-6 cameras in the adapter's pattern with declared rigs, stereo + monocular
    callers on one engine with injected batch failures, a partner that never shows up, a rig re-declared by a third thread
    mid-stream, the partner context destroyed while the other side waits, host-array associations batched across threads.
    >= 50 000 frames, every result checked against the image it belongs to, and not one ThreadSanitizer report.
    (Round 5: the first run found gfo_ctx::pair read in gfo_extract while gfo_ctx_pair / gfo_ctx_destroy of the partner wrote it;
    it is now only touched through the atomic shared_ptr functions.)"""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ missing")
    exe = str(tmp_path / "combine_tsan")
    probe = subprocess.run(["g++", "-fsanitize=thread", "-x", "c++", "-", "-o", str(tmp_path / "probe")], input="int main(){return 0;}", capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("-fsanitize=thread is not available with this g++")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-x", "c++", os.path.join(ROOT, "tests", "host", "combine_tsan.cc"), "-o", exe,
                        "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "4"], capture_output=True, text=True, timeout=600)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:6000]
    assert r.returncode == 0, r.stderr[-3000:]
    frames = int(r.stdout.split("ok:")[1].split()[0])
    assert frames >= 50000, r.stdout


def test_synthetic_vocabulary_has_the_flattened_form_the_library_takes():
    """synth_vocabulary (bench.py's matcher_calls, tools/matcher_call_latency.py): breadth first, the children of a node one contiguous
    range BEHIND it (what gfo_vocabulary_upload checks), leaves = words numbered in tree order, every leaf at the stated depth; and the
    oracle's descent over it ends in a leaf for any descriptor (the same tree both sides walk in the GPU tests)."""
    from gf_orb_slam2_amd.synth import synth_vocabulary
    from oracle import orb_oracle as O
    O.build()
    v = synth_vocabulary(5, 3, seed=2)
    n = len(v["first_child"])
    assert n == 1 + 5 + 25 + 125 and v["depth"] == 3
    inner = v["n_children"] > 0
    assert (v["n_children"][inner] == 5).all() and (v["first_child"][inner] > np.nonzero(inner)[0]).all()
    kids = np.concatenate([np.arange(f, f + c) for f, c in zip(v["first_child"][inner], v["n_children"][inner])])
    assert sorted(kids.tolist()) == list(range(1, n))                         # every node but the root is the child of exactly one node
    leaves = np.nonzero(~inner)[0]
    assert (v["word_id"][leaves] == np.arange(len(leaves))).all() and (v["word_id"][inner] == -1).all()
    assert (v["weight64"][leaves] > 0).all() and v["descriptors"].shape == (n, 32) and v["descriptors"].dtype == np.uint8
    rng = np.random.default_rng(0)
    desc = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    word, weight, node = O.bow_transform(v, desc, levelsup=1)
    assert ((word >= 0) & (word < len(leaves))).all() and (node >= 1).all()
