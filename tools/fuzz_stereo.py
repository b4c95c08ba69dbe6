"""Randomised parity sweep of the stereo association (host-array entry point and the chained device path) against
the oracle.  usage: python tools/fuzz_stereo.py [cases] [seed]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_stereo_pair
from oracle import orb_oracle as O

O.build()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
tot_kp = tot_m = 0
t0 = time.time()
for it in range(cases):
    w = int(rng.integers(200, 1000))
    h = int(rng.integers(150, 700))
    nf = int(rng.choice([200, 1000, 2000]))
    l, r = synth_stereo_pair(w, h, int(rng.integers(0, 1 << 20)))
    if rng.random() < 0.2:      # unrelated right image: few / no matches, empty buckets
        r = rng.integers(0, 256, (h, w), dtype=np.uint8)
    bf = float(rng.uniform(5.0, 80.0))
    fx = float(rng.uniform(200.0, 600.0))
    min_x = float(rng.choice([0.0, 0.0, -20.0, 15.0]))
    # the association's form: the library's choice, the per-keypoint form, or the row form with a random band height
    import os
    form = int(rng.choice([-1, 0, 2, 3, 5, 8, 13, 16, 32]))
    if form < 0:
        os.environ.pop("GFO_STEREO_ROWS", None)
    else:
        os.environ["GFO_STEREO_ROWS"] = str(form)
    if rng.random() < 0.3:
        os.environ["GFO_STEREO_CR"] = str(int(rng.choice([64, 128, 1024])))     # staging too small (read in place) / ample
    else:
        os.environ.pop("GFO_STEREO_CR", None)
    if rng.random() < 0.25 and w * h < 400 * 300:
        # a batch of P pairs (P >= 8: all row bands of a pair on one XCD, the last group of eight partly empty when P is not a
        # multiple of 8): every pair against the oracle
        P = int(rng.choice([8, 9, 13, 16, 19]))
        pairs = [synth_stereo_pair(w, h, int(rng.integers(0, 1 << 20))) for _ in range(P)]
        extb = G.ORBextractor(nf, 1.2, 8, 20, 7, max_batch=2 * P)
        mb = G.ORBmatcher(0.8, True, extractor=extb)
        kps, descs = extb.extract_batch([im for pr in pairs for im in pr])
        prm = G.StereoParams(h, bf, bf / fx, min_x)
        mb.stereo_match_batch(prm)
        sfb = extb.GetScaleFactors()
        for q in range(P):
            kl, dl, kr, dr = kps[2 * q], descs[2 * q], kps[2 * q + 1], descs[2 * q + 1]
            ref = O.stereo_match(kl, dl, kr, dr, sfb, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
            got = mb.stereo_fetch(q, max(len(kl), 1))
            tot_kp += len(kl)
            tot_m += int(ref[0])
            if not (got[0] == ref[0] and all(a[:len(kl)].tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:]))):
                bad += 1
                print(f"MISMATCH case {it} (batch of {P}, pair {q}): {w}x{h} nf={nf} form={form}", flush=True)
        extb.close()
        continue
    ext = G.ORBextractor(nf, 1.2, 8, 20, 7, max_batch=2)
    m = G.ORBmatcher(0.8, True, extractor=ext)
    (kl, kr), (dl, dr) = ext.extract_batch([l, r])
    sf = ext.GetScaleFactors()
    prm = G.StereoParams(h, bf, bf / fx, min_x)
    ref = O.stereo_match(kl, dl, kr, dr, sf, prm.n_rows, prm.mbf, prm.mb, prm.min_x)
    tot_kp += len(kl)
    tot_m += int(ref[0])
    m.stereo_match_batch(prm)                       # device path on the batch just extracted
    dev = m.stereo_fetch(0, max(len(kl), 1))
    host = m.ComputeStereoMatches(kl, dl, kr, dr, sf, prm)
    for name, got in (("device", dev), ("host", host)):
        ok = got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:]))
        if not ok:
            bad += 1
            print(f"MISMATCH case {it} ({name}): {w}x{h} nf={nf} bf={bf:.2f} fx={fx:.1f} minx={min_x}: nmatched {got[0]} vs {ref[0]}", flush=True)
    ext.close()
    if it % 25 == 24:
        print(f"{it + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {cases} cases, {bad} mismatches; {tot_kp} left keypoints, {tot_m} accepted matches in total")
sys.exit(1 if bad else 0)
