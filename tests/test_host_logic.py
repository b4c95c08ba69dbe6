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
    per, total = bench.algorithmic_bytes(752, 480, sizes, 2000, stereo=False)
    assert total == 9186428
    assert per["blur"] == 2 * 1117367 and per["fast"] == 1117367 and per["orient_desc"] == 2000 * 2178
    _, total_st = bench.algorithmic_bytes(752, 480, sizes, 2000, stereo=True)
    assert total_st == 9186428 + 256000 / 2              # half a pair's association per image
    sizes_b = bench.level_sizes(1920, 1080, inv)
    assert bench.algorithmic_bytes(1920, 1080, sizes_b, 4000, False)[1] == 36462884


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
