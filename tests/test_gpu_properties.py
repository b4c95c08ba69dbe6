"""Size-independent properties at the bench's full batch size (128 images, 752x480, 2000 features): the oracle
is too slow to check every image of every batch, so these pin the batched device path against itself and
against invariants of the reference algorithm."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B = 128


@pytest.fixture(scope="module")
def batch():
    import torch
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.synth import synth_stereo_pair
    frames = []
    for p in range(8):                       # 8 distinct pairs tiled to 128 images
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
    # two of the 128 against the oracle
    for i in (3, 77):
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
