"""A short, seeded slice of the randomised sweeps in tools/fuzz_parity.py / tools/fuzz_stereo.py: random sizes,
parameters and image statistics, GPU vs oracle, bit for bit."""
import numpy as np
import pytest

from conftest import synth_frame

pytestmark = pytest.mark.gpu


def _random_image(rng, w, h, kind):
    if kind == 0:
        return synth_frame(w, h, int(rng.integers(0, 1 << 20)))
    if kind == 1:        # white noise: FAST over-fires, both polarities everywhere
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 2:        # low-contrast noise around a level: cells that need the second threshold
        return (rng.integers(0, 24, (h, w)) + int(rng.integers(0, 230))).astype(np.uint8)
    if kind == 3:        # saturated blocks and stripes: ties, flat plateaus
        img = np.zeros((h, w), np.uint8)
        for _ in range(60):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            img[y0:y0 + int(rng.integers(2, 80)), x0:x0 + int(rng.integers(2, 80))] = int(rng.choice([0, 255, 128, 64]))
        img[:: int(rng.integers(3, 17))] ^= 255
        return img
    yy, xx = np.mgrid[0:h, 0:w]   # smooth gradient + sparse impulses
    img = ((xx * 255 // max(w - 1, 1) + yy * 255 // max(h - 1, 1)) // 2).astype(np.uint8)
    idx = rng.integers(0, h * w, max(h * w // 200, 1))
    img.reshape(-1)[idx] = rng.integers(0, 256, len(idx), dtype=np.uint8)
    return img


@pytest.mark.parametrize("seed", [101, 202])
def test_random_extractions_match_oracle(oracle, monkeypatch, seed):
    import gf_orb_slam2_amd as G
    rng = np.random.default_rng(seed)
    done = 0
    for it in range(30):
        w, h = int(rng.integers(64, 1000)), int(rng.integers(48, 700))
        nf = int(rng.choice([50, 300, 1000, 2000]))
        sf = float(rng.choice([1.1, 1.2, 1.2, 1.3, 1.5, 2.0]))
        nl = int(rng.integers(2, 11 if sf < 1.4 else 5))
        ini = int(rng.integers(5, 60))
        mn = int(rng.integers(1, ini + 1))
        img = _random_image(rng, w, h, int(rng.integers(0, 5)))
        monkeypatch.setenv("GFO_PYR_BAND_MIN_WG", "1" if it % 2 == 0 else "100000000")   # banded (forced: a single image takes the per-level form by default) / per-level
        monkeypatch.setenv("GFO_PYR_LDS_KB", str(int(rng.choice([8, 16, 32, 64]))))
        monkeypatch.setenv("GFO_PYR_MAX_W", "100000")
        monkeypatch.setenv("GFO_PYR_MAX_OVERHEAD", "100")
        monkeypatch.setenv("GFO_PYR_GROUP", str(int(rng.integers(2, 6))))
        ext = G.ORBextractor(nf, sf, nl, ini, mn)
        try:
            gk, gd = ext(img)
        except G.GfoError as e:
            assert e.code == -1, e          # only the documented refusals (oversized level quota)
            continue
        finally:
            ext.close()
        ok, od = oracle.OracleExtractor(nf, sf, nl, ini, mn)(img)
        assert gk.tobytes() == ok.tobytes() and (gd == od).all(), f"case {it}: {w}x{h} nf={nf} sf={sf} nl={nl} th={ini}/{mn}"
        done += 1
    assert done >= 20


def test_random_stereo_pairs_match_oracle(oracle):
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd.synth import synth_stereo_pair
    rng = np.random.default_rng(303)
    for it in range(25):
        w, h = int(rng.integers(200, 900)), int(rng.integers(150, 600))
        nf = int(rng.choice([200, 1000, 2000]))
        l, r = synth_stereo_pair(w, h, int(rng.integers(0, 1 << 20)))
        if rng.random() < 0.2:      # unrelated right image: few / no matches, empty buckets
            r = rng.integers(0, 256, (h, w), dtype=np.uint8)
        bf, fx = float(rng.uniform(5.0, 80.0)), float(rng.uniform(200.0, 600.0))
        ext = G.ORBextractor(nf, 1.2, 8, 20, 7, max_batch=2)
        m = G.ORBmatcher(0.8, True, extractor=ext)
        (kl, kr), (dl, dr) = ext.extract_batch([l, r])
        sf = ext.GetScaleFactors()
        prm = G.StereoParams(h, bf, bf / fx, float(rng.choice([0.0, -20.0, 15.0])))
        ref = oracle.stereo_match(kl, dl, kr, dr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
        m.stereo_match_batch(prm)
        for got in (m.stereo_fetch(0, max(len(kl), 1)), m.ComputeStereoMatches(kl, dl, kr, dr, sf, prm)):
            assert got[0] == ref[0], f"case {it}"
            for a, b in zip(got[1:], ref[1:]):
                assert a.tobytes() == b.tobytes(), f"case {it}"
        ext.close()


def test_projection_queries_fuzz(oracle):
    """random frames, bounds (non-zero origin included), query sets, modes: the query-form projection search against the
    oracle's serial statement.  Sizes straddle the switch between the two round-0 paths (grid in LDS from 4096 queries)."""
    import gf_orb_slam2_amd as G
    ext = G.ORBextractor(500, 1.2, 8, 20, 7)
    rng = np.random.default_rng(2026)
    sf = ext.GetScaleFactors()
    for case in range(40):
        n = int(rng.integers(1, 3500))
        m = int(rng.choice([0, 1, 50, 700, 4095, 4096, 6000, 12000]))
        x0, y0 = float(rng.choice([0.0, -37.5, 12.25])), float(rng.choice([0.0, -20.0, 8.5]))
        w, h = float(rng.integers(200, 2000)), float(rng.integers(150, 1200))
        b = (x0, y0, x0 + w, y0 + h)
        kp = np.zeros(n, oracle.KEYPOINT_DTYPE)
        kp["x"] = rng.uniform(x0 - 5, x0 + w + 5, n).astype(np.float32)      # a few keypoints outside the grid
        kp["y"] = rng.uniform(y0 - 5, y0 + h + 5, n).astype(np.float32)
        kp["octave"] = rng.integers(0, 8, n)
        kp["angle"] = rng.uniform(0, 360, n).astype(np.float32)
        desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        u = None if rng.random() < 0.4 else np.where(rng.random(n) < 0.6, kp["x"] - rng.uniform(0.5, 60, n), -1).astype(np.float32)
        taken = None if rng.random() < 0.4 else (rng.random(n) < 0.15).astype(np.uint8)
        q = np.zeros(m, oracle.PROJ_QUERY_DTYPE)
        qd = np.zeros((m, 32), np.uint8)
        if m:
            src = rng.integers(0, n, m)
            spread = float(rng.choice([0.5, 3.0, 15.0]))
            q["u"] = kp["x"][src] + rng.normal(0, spread, m); q["v"] = kp["y"][src] + rng.normal(0, spread, m)
            q["ur"] = q["u"] - rng.uniform(0, 60, m).astype(np.float32)
            q["radius"] = (np.float32(rng.choice([1.0, 3.0, 7.0, 15.0])) * sf[kp["octave"][src]]).astype(np.float32)
            mode_l = int(rng.integers(0, 4))
            octv = kp["octave"][src]
            if mode_l == 0: q["min_level"] = octv - 1; q["max_level"] = octv + 1
            elif mode_l == 1: q["min_level"] = octv; q["max_level"] = -1
            elif mode_l == 2: q["min_level"] = 0; q["max_level"] = octv
            else: q["min_level"] = -1; q["max_level"] = -1
            q["angle"] = ((kp["angle"][src] + rng.normal(0, 30, m)) % 360).astype(np.float32)
            fl = np.full(m, 5, np.int32); fl[rng.random(m) < 0.15] = 0; fl[rng.random(m) < 0.3] &= ~4
            q["flags"] = fl
            qd = desc[src].copy()
            for _ in range(int(rng.integers(0, 40))):
                sel = rng.random(m) < 0.5
                bits = rng.integers(0, 256, m)
                qd[sel, bits[sel] >> 3] ^= (1 << (bits[sel] & 7)).astype(np.uint8)
        use_ratio, ori = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        ratio, th = float(rng.choice([0.6, 0.8, 0.95])), int(rng.choice([40, 64, 100, 255]))
        ref = oracle.search_by_projection_queries(kp, desc, u, kp["angle"], b, q, qd, use_ratio, ratio, th, ori, taken)
        got = G.ORBmatcher(ratio, ori, extractor=ext).SearchByProjectionQueries(kp, desc, u, kp["angle"], b, q, qd, use_ratio=use_ratio,
                                                                               th_dist=th, kp_taken=taken)
        assert got[0] == ref[0], f"case {case}"
        np.testing.assert_array_equal(got[1], ref[1], err_msg=f"case {case}")
        np.testing.assert_array_equal(got[2][got[1] >= 0], ref[2][ref[1] >= 0], err_msg=f"case {case}")
    with pytest.raises(G.GfoError) as e:          # 256 would accept "no candidate" (bestDist = 256 in the reference)
        G.ORBmatcher(0.8, False, extractor=ext).SearchByProjectionQueries(kp, desc, None, kp["angle"], b, q, qd, th_dist=256)
    assert e.value.code == -1
    ext.close()
