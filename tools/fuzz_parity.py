"""Randomised parity sweep (not part of the test suite: minutes of GPU + oracle time): random image sizes,
extractor parameters and image statistics, GPU extraction vs the oracle, bit for bit; both pyramid paths.
usage: python tools/fuzz_parity.py [cases] [seed] [max_w max_h]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_frame
from oracle import orb_oracle as O

O.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAX_W, MAX_H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1300, 900)   # exclusive upper bounds of the image size
bad = 0
t0 = time.time()
for it in range(cases):
    w = int(rng.integers(64, MAX_W))
    h = int(rng.integers(48, MAX_H))
    nf = int(rng.choice([50, 300, 1000, 2000, 4000]))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.2, 1.3, 1.5, 2.0]))
    nl = int(rng.integers(1, 11 if sf < 1.4 else 5))
    ini = int(rng.integers(5, 60))
    mn = int(rng.integers(1, ini + 1))
    kind = int(rng.integers(0, 5))
    if kind == 0:
        img = synth_frame(w, h, int(rng.integers(0, 1 << 20)))
    elif kind == 1:      # white noise: FAST over-fires, both polarities everywhere
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 2:      # low-contrast noise around a level: second-round cells
        img = (rng.integers(0, 24, (h, w)) + int(rng.integers(0, 230))).astype(np.uint8)
    elif kind == 3:      # saturated blocks and stripes: ties, flat plateaus
        img = np.zeros((h, w), np.uint8)
        for _ in range(60):
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            img[y0:y0 + int(rng.integers(2, 80)), x0:x0 + int(rng.integers(2, 80))] = int(rng.choice([0, 255, 128, 64]))
        img[:: int(rng.integers(3, 17))] ^= 255
    else:                # smooth gradient + sparse impulses
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((xx * 255 // max(w - 1, 1) + yy * 255 // max(h - 1, 1)) // 2).astype(np.uint8)
        idx = rng.integers(0, h * w, max(h * w // 200, 1))
        img.reshape(-1)[idx] = rng.integers(0, 256, len(idx), dtype=np.uint8)
    os.environ["GFO_PYR_BAND_MIN_WG"] = "1" if it % 2 == 0 else "100000000"   # banded (forced) / per-level
    os.environ["GFO_PYR_LDS_KB"] = str(int(rng.choice([8, 16, 32, 64])))
    os.environ["GFO_PYR_MAX_W"] = "100000"
    os.environ["GFO_PYR_MAX_OVERHEAD"] = "100"
    os.environ["GFO_PYR_GROUP"] = str(2 + it % 4)
    ext = G.ORBextractor(nf, sf, nl, ini, mn)
    try:
        gk, gd = ext(img)
    except G.GfoError as e:
        print(f"case {it}: {w}x{h} nf={nf} sf={sf} nl={nl} th={ini}/{mn} kind={kind}: refused: {e}")
        if e.code != -1:     # only a documented refusal (invalid configuration) is acceptable
            bad += 1
        continue
    finally:
        ext.close()
    if it % 3 == 2:
        # every third case also as a BATCH (beyond the per-frame path: separate blur launch) with the blur forced onto the matrix cores
        # whatever the width (k_blur_mfma; levels narrower than 64 px send the launch back to the streaming form): every image against the
        # oracle, every blurred pixel of one of them against the oracle's blur of its own level
        os.environ["GFO_BLUR_MFMA"] = "1"
        imgs = [img] + [np.ascontiguousarray(np.roll(img, (7 * k, 13 * k), (0, 1))) for k in range(1, 10)]
        ext = G.ORBextractor(nf, sf, nl, ini, mn)
        try:
            gks, gds = ext.extract_batch(imgs)
            j = int(rng.integers(0, len(imgs)))
            oe = O.OracleExtractor(nf, sf, nl, ini, mn)
            okj, odj = oe(imgs[j])
            okb = len(gks[j]) == len(okj) and gks[j].tobytes() == okj.tobytes() and (gds[j] == odj).all()
            for l in range(nl):
                okb = okb and (ext.debug_blurred_level(l, image=j) == O.gaussian_blur7(oe.level(l))).all()
            ok0, od0 = O.OracleExtractor(nf, sf, nl, ini, mn)(imgs[0])
            okb = okb and gks[0].tobytes() == ok0.tobytes() and (gds[0] == od0).all()
            if not okb:
                bad += 1
                print(f"MISMATCH (batch) case {it}: {w}x{h} nf={nf} sf={sf} nl={nl} th={ini}/{mn} kind={kind} image {j}", flush=True)
        except G.GfoError as e:
            if e.code != -1:
                bad += 1
                print(f"case {it} (batch): {e}", flush=True)
        finally:
            ext.close()
            del os.environ["GFO_BLUR_MFMA"]
    ok, od = O.OracleExtractor(nf, sf, nl, ini, mn)(img)
    same = len(gk) == len(ok) and gk.tobytes() == ok.tobytes() and (gd == od).all()
    if not same:
        bad += 1
        print(f"MISMATCH case {it}: {w}x{h} nf={nf} sf={sf} nl={nl} th={ini}/{mn} kind={kind}: gpu {len(gk)} vs oracle {len(ok)}", flush=True)
    if it % 25 == 24:
        print(f"{it + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
