"""GPU parity tests of the device-resident chain extract_batch_device -> [stereo_match_batch] ->
search_by_projection_batch (gfo_map_upload / gfo_search_by_projection_batch / gfo_projection_fetch) against the
oracle's serial ORBmatcher::SearchByProjection(Frame&, MapPoints, th) (ORBmatcher.cc:155-241), frame by frame.
Indices and scores bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FX, BF = 435.2046959714599, 47.90639384423901


def _extract_stream(G, frames, nfeat, torch):
    h, w = frames[0].shape
    B = len(frames)
    ext = G.ORBextractor(nfeat, 1.2, 8, 20, 7, max_batch=B)
    d_imgs = torch.from_numpy(np.stack(frames)).cuda()
    ext.extract_batch_device(d_imgs.data_ptr(), B, w, h)
    ext.synchronize()
    return ext, d_imgs


def test_batch_of_1080p_frames_against_50k_map(oracle):
    """config 4 as a stream: 16 frames 1920x1080 @4000 of one scene, ONE resident 50 000-point S3 map, one launch
    chain for the whole batch -- every frame equals the oracle's serial search on that frame's keypoints"""
    import torch
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.synth import synth_local_map, synth_stream
    w, h, B, M = 1920, 1080, 16, 50000
    frames, offs = synth_stream(w, h, B, idx=3)
    ext, d_imgs = _extract_stream(G, frames, 4000, torch)
    kd = [ext.batch_fetch(i) for i in range(B)]
    mpd, mps = synth_local_map(kd[0][0], kd[0][1], offs, w, h, M, 4000, seed=7)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    m.map_upload(mpd)
    bounds = (0.0, 0.0, float(w), float(h))
    m.search_by_projection_batch(mps, bounds, th=3.0)
    sf = ext.GetScaleFactors()
    total = 0
    for f in range(B):
        kp, desc = kd[f]
        ref = oracle.search_by_projection(kp, desc, None, sf, bounds, mps[f], mpd, 3.0, 0.8, None)
        nm, out_mp, out_sc = m.projection_fetch(f, len(kp))
        assert nm == ref[0], f"frame {f}"
        np.testing.assert_array_equal(out_mp[:len(kp)], ref[1])
        np.testing.assert_array_equal(out_sc[:len(kp)], ref[2])
        total += nm
    assert total > B * 1000          # the stream really is tracked against the map
    # same search with the projections already resident on the device
    d_mps = torch.from_numpy(mps.view(np.uint8).reshape(B, -1)).cuda()
    ext.extract_batch_device(d_imgs.data_ptr(), B, w, h)
    m.search_by_projection_batch(d_mps.data_ptr(), bounds, th=3.0, device_ptrs=True)
    for f in (0, B - 1):
        kp, desc = kd[f]
        ref = oracle.search_by_projection(kp, desc, None, sf, bounds, mps[f], mpd, 3.0, 0.8, None)
        nm, out_mp, _ = m.projection_fetch(f, len(kp))
        assert nm == ref[0]
        np.testing.assert_array_equal(out_mp[:len(kp)], ref[1])
    ext.close()


@pytest.mark.parametrize("M", [12000, 3500])
def test_stereo_chain_feeds_the_projection(oracle, M):
    """extract (L, R interleaved) -> stereo association -> projection search with mvuRight gating (:201-206) and
    keypoints taken on entry, contended map (blocking and non-blocking points mixed).  M = 3500: 4 frames x 3500 points is a call of
    at most 16 384 points, so round 0 is the wavefront-per-point kernel with several frames in one launch (one candidate-list pool for
    all of them); M = 12 000: the thread-per-point kernels."""
    import torch
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.synth import synth_stereo_pair
    w, h, P = 752, 480, 4
    frames = []
    for p in range(P):
        l, r = synth_stereo_pair(w, h, 40 + p)
        frames += [l, r]
    ext, d_imgs = _extract_stream(G, frames, 2000, torch)
    m = G.ORBmatcher(0.7, True, extractor=ext)
    sp = G.StereoParams(h, BF, BF / FX, 0.0)
    m.stereo_match_batch(sp)
    cap = ext.max_keypoints()
    rng = np.random.default_rng(5)
    kd = [ext.batch_fetch(2 * p) for p in range(P)]
    ur = [m.stereo_fetch(p, cap)[1] for p in range(P)]
    # the map imitates pair 0's left keypoints; every pair sees it at slightly different projections
    kp0, d0 = kd[0]
    src = rng.integers(0, len(kp0), M)
    mpd = d0[src].copy()
    flips = rng.integers(0, 256, (M, 8))
    for j in range(8):
        sel = rng.random(M) < 0.5
        mpd[sel, flips[sel, j] >> 3] ^= (1 << (flips[sel, j] & 7)).astype(np.uint8)
    mps = np.zeros((P, M), G.MAP_POINT_DTYPE)
    taken = np.zeros((P, cap), np.uint8)
    for p in range(P):
        q = mps[p]
        q["proj_x"] = kp0["x"][src] + rng.normal(0, 3, M)
        q["proj_y"] = kp0["y"][src] + rng.normal(0, 3, M)
        q["proj_xr"] = q["proj_x"] - rng.uniform(0, 30, M)
        q["level"] = np.clip(kp0["octave"][src] + rng.integers(-1, 2, M), -1, 8)   # -1 and 8: out-of-range levels are skipped
        q["view_cos"] = rng.choice([1.0, 0.9985, 0.99], M)
        fl = np.full(M, 5, np.int32)
        fl[rng.random(M) < 0.05] = 4
        fl[rng.random(M) < 0.05] |= 2
        fl[rng.random(M) < 0.3] &= ~4
        q["flags"] = fl
        taken[p] = rng.random(cap) < 0.15
    mps["level"] = np.where((mps["level"] < 0) | (mps["level"] > 7), np.where(rng.random((P, M)) < 0.5, mps["level"], 3), mps["level"])
    bounds = (0.0, 0.0, float(w), float(h))
    m.map_upload(mpd)
    m.search_by_projection_batch(mps, bounds, th=5.0, kp_taken=taken, stereo=True)
    sf = ext.GetScaleFactors()
    for p in range(P):
        kp, desc = kd[p]
        n = len(kp)
        # (levels -1 and 8 -- outside the scale table -- are skipped by the library and by the oracle alike)
        ref = oracle.search_by_projection(kp, desc, ur[p][:n], sf, bounds, mps[p], mpd, 5.0, 0.7, taken[p][:n])
        nm, out_mp, out_sc = m.projection_fetch(p, n)
        assert nm == ref[0] and nm > 200, f"pair {p}"
        np.testing.assert_array_equal(out_mp[:n], ref[1])
        np.testing.assert_array_equal(out_sc[:n], ref[2])
    ext.close()


def test_call_order_and_capacity_errors():
    import torch
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.synth import synth_frame
    img = synth_frame(320, 240, 1)
    ext, d_imgs = _extract_stream(G, [img, img], 300, torch)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    mps = np.zeros((2, 10), G.MAP_POINT_DTYPE)
    b = (0.0, 0.0, 320.0, 240.0)
    with pytest.raises(G.GfoError) as e:
        m.search_by_projection_batch(mps, b)          # no map yet
    assert e.value.code == -5
    with pytest.raises(G.GfoError) as e:
        m.projection_fetch(0, 10)                     # nothing searched yet
    assert e.value.code == -5
    m.map_upload(np.zeros((10, 32), np.uint8))
    with pytest.raises(G.GfoError) as e:
        m.search_by_projection_batch(mps, b, stereo=True)   # no stereo batch matched
    assert e.value.code == -5
    m.search_by_projection_batch(mps, b)              # all points inactive (flags 0): zero matches everywhere
    nm, out_mp, _ = m.projection_fetch(1, ext.max_keypoints())
    assert nm == 0 and (out_mp == -1).all()
    with pytest.raises(G.GfoError) as e:
        m.projection_fetch(0, 1)                      # more keypoints than the caller's capacity
    assert e.value.code == -3
    with pytest.raises(G.GfoError):
        m.projection_fetch(2, 10)
    ext.close()
    with pytest.raises(G.GfoError) as e:              # the matcher sees its extractor closed (no dangling context)
        m.projection_fetch(0, 10)
    assert e.value.code == -5
