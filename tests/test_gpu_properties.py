"""Size-independent properties at the bench's full batch size (gf_orb_slam2_amd.HEADLINE_BATCH = bench.py's --batch default:
256 images, 752x480, 2000 features): the oracle
is too slow to check every image of every batch, so these pin the batched device path against itself and
against invariants of the reference algorithm."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from gf_orb_slam2_amd import HEADLINE_BATCH as B      # the constant bench.py's --batch defaults to


@pytest.fixture(scope="module")
def batch():
    import torch
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.synth import synth_stereo_pair
    frames = []
    for p in range(8):                       # 8 distinct pairs tiled to B images
        l, r = synth_stereo_pair(752, 480, 500 + p)
        frames += [l, r]
    imgs = np.stack([frames[i % 16] for i in range(B)])
    t = torch.from_numpy(imgs).cuda()
    ext = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=B)
    ext.set_stream(torch.cuda.current_stream().cuda_stream)
    yield ext, t, imgs
    ext.set_stream(0)
    ext.close()


def _run(ext, t):
    import torch
    ext.extract_batch_device(t.data_ptr(), B, 752, 480)
    torch.cuda.synchronize()
    n, per_level = ext.batch_counts(B, per_level=True)
    return n, per_level


def test_full_batch_is_deterministic_and_position_independent(batch, oracle):
    ext, t, imgs = batch
    n1, pl1 = _run(ext, t)
    first = [ext.batch_fetch(i) for i in range(16)]
    n2, pl2 = _run(ext, t)
    np.testing.assert_array_equal(n1, n2)                     # idempotent
    np.testing.assert_array_equal(pl1, pl2)
    for i in range(B):                                        # image i == its copy at i % 16, wherever it sits
        k, d = ext.batch_fetch(i)
        fk, fd = first[i % 16]
        assert k.tobytes() == fk.tobytes() and (d == fd).all()
    # two of the batch against the oracle
    for i in (3, B - 51):
        ok, od = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)(imgs[i])
        k, d = ext.batch_fetch(i)
        assert k.tobytes() == ok.tobytes() and (d == od).all()


def test_full_batch_invariants(batch):
    ext, t, imgs = batch
    n, per_level = _run(ext, t)
    quota = ext.mnFeaturesPerLevel
    assert (per_level.sum(axis=1) == n).all()
    assert (per_level <= quota[None, :] + 3).all()            # quadtree overshoot <= 3 nodes per level
    sf = ext.GetScaleFactors()
    for i in range(0, B, 17):
        k, d = ext.batch_fetch(i)
        assert (np.diff(k["octave"]) >= 0).all()              # rows level by level
        assert ((k["angle"] >= 0) & (k["angle"] < 360)).all()
        assert (k["size"] == np.floor(31 * sf[k["octave"]])).all()
        lv_w = np.rint(np.float32(752) / sf[k["octave"]]); lv_h = np.rint(np.float32(480) / sf[k["octave"]])
        x = k["x"] / sf[k["octave"]]; y = k["y"] / sf[k["octave"]]
        assert (x > 18.9).all() and (x < lv_w - 18.9).all() and (y > 18.9).all() and (y < lv_h - 18.9).all()
        assert len(np.unique(np.stack([k["x"], k["y"], k["octave"]], 1), axis=0)) == len(k)   # one keypoint per pixel
        assert d.any(axis=1).all()


def test_full_batch_stereo_properties(batch, oracle):
    import torch
    import gf_orb_slam2_amd as G
    ext, t, imgs = batch
    _run(ext, t)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    p = G.StereoParams(480, 47.90639384423901, 47.90639384423901 / 435.2046959714599, 0.0)
    m.stereo_match_batch(p)
    torch.cuda.synchronize()
    ref = {}
    for pair in range(B // 2):
        kl, dl = ext.batch_fetch(2 * pair)
        kr, dr = ext.batch_fetch(2 * pair + 1)
        nm, u, dp, bd, bi = m.stereo_fetch(pair, len(kl))
        matched = u >= 0
        # every accepted match is a real candidate: right index valid, distance recomputes, disparity in range
        assert (bi[matched] >= 0).all() and (bi[matched] < len(kr)).all()
        sel = np.nonzero(matched)[0][:50]
        for i in sel:
            assert bd[i] == int(np.unpackbits(dl[i] ^ dr[bi[i]]).sum()) and bd[i] < 75
            assert abs(int(kr["octave"][bi[i]]) - int(kl["octave"][i])) <= 1
            assert dp[i] > 0 and kl["x"][i] - u[i] >= -1e-3
        assert matched.sum() > 200                          # the synthetic pairs really match
        key = pair % 8
        sig = (nm, u.tobytes(), bd.tobytes())
        assert ref.setdefault(key, sig) == sig                # same pair -> same answer in every slot
    # one pair against the oracle
    kl, dl = ext.batch_fetch(10); kr, dr = ext.batch_fetch(11)
    o = oracle.stereo_match(kl, dl, kr, dr, ext.GetScaleFactors(), 480, p.mbf, p.mb, 0.0)
    g = m.stereo_fetch(5, len(kl))
    assert g[0] == o[0]
    for a, b in zip(g[1:], o[1:]):
        assert a.tobytes() == b.tobytes()


@pytest.fixture(scope="module")
def oracle_refs(batch, oracle):
    """the 16 distinct images (8 pairs) of the batch through the oracle, once per session"""
    _, _, imgs = batch
    oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
    ext_refs = [oe(imgs[i]) for i in range(16)]
    sf = oe.scale_factors
    st_refs = [oracle.stereo_match(ext_refs[2 * p][0], ext_refs[2 * p][1], ext_refs[2 * p + 1][0], ext_refs[2 * p + 1][1], sf, 480,
                                   47.90639384423901, 47.90639384423901 / 435.2046959714599, 0.0) for p in range(8)]
    return ext_refs, st_refs


def test_headline_configuration_against_the_oracle(batch, oracle_refs):
    """bench.py's exact headline shape: HEADLINE_BATCH = 256 images per step (the XCD8 grids with
    32 image groups, 128-pair stereo launches, the 512- / 256-thread bucket / cut kernels, the 256-image delivery layout), TWO
    contexts chained behind each other's pyramid (gfo_ctx_chain), four alternating steps submitted without a synchronisation in
    between, stereo association of the B/2 pairs -- then, per context, 16 images (every distinct image once, at positions spread over
    the batch) and 8 pairs compared with the oracle bit for bit, read back through gfo_batch_deliver (the delivery
    path bench.py's `value_delivered` uses) and cross-checked against gfo_batch_fetch / gfo_stereo_fetch."""
    import torch
    import gf_orb_slam2_amd as G
    _, t, imgs = batch
    ext_refs, st_refs = oracle_refs
    # a second input batch with the 8 pairs in another order, so that a frame's slot differs between the inputs
    NP = B // 2
    perm = np.concatenate([[2 * ((p * 5 + 3) % NP), 2 * ((p * 5 + 3) % NP) + 1] for p in range(NP)])
    t2 = t[torch.from_numpy(perm).cuda()].contiguous()
    inputs, ident = [t, t2], [np.arange(B) % 16, perm % 16]
    prm = G.StereoParams(480, 47.90639384423901, 47.90639384423901 / 435.2046959714599, 0.0)
    exts = [G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=B) for _ in range(2)]
    ms = [G.ORBmatcher(0.8, True, extractor=e) for e in exts]
    exts[0].chain_after(exts[1], exts[0].STAGE_PYRAMID)
    exts[1].chain_after(exts[0], exts[1].STAGE_PYRAMID)
    lay = None
    blocks = []
    for step in range(4):                          # ctx 0: inputs 0, 0 ; ctx 1: inputs 1, 1 -- then swapped roles below
        k = step & 1
        src = (step + (step >> 1)) & 1             # 0, 1, 1, 0: each context sees both inputs
        exts[k].extract_batch_device(inputs[src].data_ptr(), B, 752, 480)
        ms[k].stereo_match_batch(prm)
        if step >= 2:
            if lay is None:
                lay = exts[k].batch_deliver()
            blk = torch.empty(lay.bytes, dtype=torch.uint8).pin_memory()
            exts[k].batch_deliver(blk.data_ptr(), lay.bytes)
            blocks.append((k, src, blk))
    for k, src, blk in blocks:
        exts[k].deliver_wait()
        v = G.ORBextractor.delivered_views(blk.numpy(), lay)
        assert v["flags"][0] == 0 and lay.stereo == 1 and lay.nimg == B
        slots = [next(i for i in range((7 * j) % B, B + (7 * j) % B) if ident[src][i % B] == j) % B for j in range(16)]   # image j somewhere past 7j
        for j, i in enumerate(slots):
            n = int(v["counts"][i])
            ok, od = ext_refs[j]
            assert n == len(ok) and v["kp"][i, :n].tobytes() == ok.tobytes() and (v["desc"][i, :n] == od).all(), f"ctx {k} image slot {i}"
        pairs = sorted({i // 2 for i in slots})[:8] if len({i // 2 for i in slots}) >= 8 else list(range(0, NP, NP // 8))
        seen = set()
        for pr in pairs + list(range(NP)):
            j = int(ident[src][2 * pr]) // 2
            if j in seen:
                continue
            seen.add(j)
            nl = int(v["counts"][2 * pr])
            nm, u, dp, bd, bi = st_refs[j]
            assert int(v["nmatched"][pr]) == nm
            for a, b_ in ((v["u_right"][pr, :nl], u), (v["depth"][pr, :nl], dp), (v["best_dist"][pr, :nl], bd), (v["best_idx"][pr, :nl], bi)):
                assert a.tobytes() == b_.tobytes(), f"ctx {k} pair {pr}"
        assert len(seen) == 8
        # the delivered block equals what the fetch entry points hand out
        kf, df = exts[k].batch_fetch(slots[5])
        assert kf.tobytes() == v["kp"][slots[5], :len(kf)].tobytes() and (df == v["desc"][slots[5], :len(kf)]).all()
        g = ms[k].stereo_fetch(3, int(v["counts"][6]))
        assert g[0] == int(v["nmatched"][3]) and g[1].tobytes() == v["u_right"][3, :len(g[1])].tobytes()
    for e in exts:
        e.close()


def test_count_all_gather_over_rccl_single_rank():
    """bench.py's only collective, on the real backend: torch.distributed 'nccl' (= RCCL) all-gathering the per-image
    keypoint counts straight out of the arena (the int32 vector is aliased through __cuda_array_interface__, no
    copy).  One rank is all a 1-GPU box offers; it still exercises init_process_group('nccl'), the aliased device
    pointer as an RCCL send buffer and the stream ordering against the extraction (the N > 1 logic is covered by the
    2-rank gloo test in tests/test_host_logic.py)."""
    import ctypes
    import os
    import torch
    import torch.distributed as dist
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.sharding import gather_counts
    from gf_orb_slam2_amd.synth import synth_frame
    import bench

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B = 8
        imgs = np.stack([synth_frame(320, 240, i) for i in range(B)])
        d = torch.from_numpy(imgs).cuda()
        st = torch.cuda.Stream()
        ext = G.ORBextractor(300, 1.2, 8, 20, 7, max_batch=B)
        ext.set_stream(st.cuda_stream)
        ext.extract_batch_device(d.data_ptr(), B, 320, 240)
        L = G.load_library()
        p_cnt, stride = ctypes.c_void_p(), ctypes.c_int()
        L.gfo_batch_device_views(ext.handle, None, None, ctypes.byref(p_cnt), ctypes.byref(stride))
        counts = torch.as_tensor(bench._DevArray(p_cnt.value, B), device="cuda")
        out = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        with torch.cuda.stream(st):
            # gather_counts short-cuts world == 1; call the collective itself
            dist.all_gather_into_tensor(out, counts)
        torch.cuda.synchronize()
        ref = ext.batch_counts(B)
        assert (out.cpu().numpy() == ref).all() and (ref > 100).all()
        assert gather_counts(counts, 1, dist) is counts
        ext.set_stream(0)
        ext.close()
    finally:
        if created:
            dist.destroy_process_group()
