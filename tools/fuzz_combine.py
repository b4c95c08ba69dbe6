"""Stress of the frame combiner (not part of the test suite: GPU + oracle minutes): rounds of K threads, each with its own
combining extractor drawn from a few parameter sets and image sizes (so several engines are alive at once), every thread
mixing single-image calls and stereo frames, contexts created and destroyed between rounds -- every result against the
oracle, bit for bit.  usage: python tools/fuzz_combine.py [rounds] [seed]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import gf_orb_slam2_amd as G
from gf_orb_slam2_amd.synth import synth_frame
from oracle import orb_oracle as O

O.build()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
FX, BF = 435.2046959714599, 47.90639384423901
PARAMS = [(2000, 1.2, 8, 20, 7), (600, 1.2, 6, 25, 9), (1200, 1.3, 5, 15, 5)]
SIZES = [(752, 480), (640, 400), (320, 240)]
# reference results for every (parameter set, size, image index), made once
refs = {}
imgs = {}
t0 = time.time()
for pi, prm in enumerate(PARAMS):
    for si, (w, h) in enumerate(SIZES):
        oe = O.OracleExtractor(*prm)
        for k in range(4):
            im = imgs.setdefault((si, k), synth_frame(w, h, 100 * si + k))
            refs[(pi, si, k)] = oe(im)
        for k in (0, 2):
            (kl, dl), (kr, dr) = refs[(pi, si, k)], refs[(pi, si, k + 1)]
            refs[(pi, si, k, "st")] = O.stereo_match(kl, dl, kr, dr, oe.scale_factors, h, BF, BF / FX, 0.0)
print(f"oracle references: {time.time() - t0:.0f} s", flush=True)
bad = []
frames = 0
for rd in range(rounds):
    K = int(rng.integers(2, 13))
    plan = [(int(rng.integers(0, len(PARAMS))), int(rng.integers(0, len(SIZES))), int(rng.integers(3, 12)), int(rng.integers(0, 1 << 30))) for _ in range(K)]
    exts = [G.ORBextractor(*PARAMS[pi], max_batch=2, combining=True) for pi, _, _, _ in plan]
    rights = [G.ORBextractor(*PARAMS[pi], max_batch=2, combining=True) for pi, _, _, _ in plan]     # the right extractor of thread t's rig
    matchers = [G.ORBmatcher(0.8, True, extractor=e) for e in exts]

    def work(t):
        pi, si, n, seed = plan[t]
        r = np.random.default_rng(seed)
        w, h = SIZES[si]
        sp = G.StereoParams(h, BF, BF / FX, 0.0)
        for _ in range(n):
            u = r.random()
            if u < 0.25:     # the adapter's third call: the host-array association on this extractor's context, with or without windows
                k = int(r.choice([0, 2]))
                (kl, dl), (kr, dr) = refs[(pi, si, k)], refs[(pi, si, k + 1)]
                sfac = O.OracleExtractor(*PARAMS[pi]).scale_factors
                win = (None, None)
                if r.random() < 0.5:
                    d0 = r.uniform(0, 50, len(kl)).astype(np.float32)
                    win = (np.maximum(d0 - 6, 0).astype(np.float32), (d0 + 6).astype(np.float32))
                ref = refs[(pi, si, k, "st")] if win[0] is None else O.stereo_match(kl, dl, kr, dr, sfac, h, BF, BF / FX, 0.0, *win)
                got = matchers[t].ComputeStereoMatches(kl, dl, kr, dr, sfac, sp, *win)
                if not (got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:]))):
                    bad.append((rd, t, "assoc", pi, si, k))
            elif u < 0.45:
                # the adapter's whole pattern on a declared rig (gfo_ctx_pair): right image on a thread of its own, left here, then the
                # association -- answered from the frame's one submission, or computed when windows are passed
                k = int(r.choice([0, 2]))
                exts[t].pair_with(rights[t], sp)
                out = {}
                th = threading.Thread(target=lambda: out.__setitem__("r", rights[t](imgs[(si, k + 1)])))
                th.start()
                kl, dl = exts[t](imgs[(si, k)])
                th.join()
                kr, dr = out["r"]
                (okl, odl), (okr, odr) = refs[(pi, si, k)], refs[(pi, si, k + 1)]
                sfac = O.OracleExtractor(*PARAMS[pi]).scale_factors
                win = (None, None)
                if r.random() < 0.3:
                    d0 = r.uniform(0, 50, len(okl)).astype(np.float32)
                    win = (np.maximum(d0 - 6, 0).astype(np.float32), (d0 + 6).astype(np.float32))
                ref = refs[(pi, si, k, "st")] if win[0] is None else O.stereo_match(okl, odl, okr, odr, sfac, h, BF, BF / FX, 0.0, *win)
                got = matchers[t].ComputeStereoMatches(kl, dl, kr, dr, sfac, sp, *win)
                if not (kl.tobytes() == okl.tobytes() and kr.tobytes() == okr.tobytes() and (dl == odl).all() and (dr == odr).all()
                        and got[0] == ref[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[1:], ref[1:]))):
                    bad.append((rd, t, "rig", pi, si, k))
            elif u < 0.7:
                k = int(r.integers(0, 4))
                kp, d = exts[t](imgs[(si, k)])
                ok, od = refs[(pi, si, k)]
                if kp.tobytes() != ok.tobytes() or not (d == od).all():
                    bad.append((rd, t, "single", pi, si, k))
            else:
                k = int(r.choice([0, 2]))
                got = exts[t].extract_stereo(imgs[(si, k)], imgs[(si, k + 1)], sp)
                (kl, dl), (kr, dr), st = refs[(pi, si, k)], refs[(pi, si, k + 1)], refs[(pi, si, k, "st")]
                if not (got[0].tobytes() == kl.tobytes() and got[2].tobytes() == kr.tobytes() and (got[1] == dl).all() and (got[3] == dr).all()
                        and got[4] == st[0] and all(a.tobytes() == b.tobytes() for a, b in zip(got[5:], st[1:]))):
                    bad.append((rd, t, "stereo", pi, si, k))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(K)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    frames += sum(p[2] for p in plan)
    rig_frames = sum(e.combiner_counters()["rig_frames"] for e in exts)
    rig_served = sum(e.combiner_counters()["rig_served"] for e in exts)
    for e in exts + rights:
        e.close()
    print(f"[rigs: {rig_frames} frames as one submission, {rig_served} associations answered from them] ", end="")
    print(f"round {rd + 1}: {K} threads, {frames} calls so far, {len(bad)} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {rounds} rounds, {frames} calls, {len(bad)} mismatches", bad[:5])
sys.exit(1 if bad else 0)
