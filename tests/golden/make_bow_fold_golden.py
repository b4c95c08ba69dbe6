#!/usr/bin/env python3
"""Writes tests/golden/bow_fold.npz: outputs of the REFERENCE's own DBoW2 map classes (oracle/_ref/libdbow2_fold.so =
Thirdparty/DBoW2/DBoW2/BowVector.cpp + FeatureVector.cpp compiled unmodified, `make -C oracle ref`) on seeded
(word, weight, node) streams, for the 4 weightings x 3 norms of TemplatedVocabulary::transform (TemplatedVocabulary.h:1140-1212).

Run in the build container (the reference tree must exist):   python tests/golden/make_bow_fold_golden.py

Two kinds of stream:
  * "syn<k>": drawn directly -- word ids from a small alphabet (many repeats: addWeight sums in feature order), weights that are
    not exactly representable sums (logs), ~10 % stop words (weight 0), a handful of negative and denormal weights, node ids
    from a smaller alphabet; lengths 0, 1, 77, 2000, 8192;
  * "voc<k>": the stream the oracle's descent (orc_bow_stream) gives for seeded descriptors on a seeded synthetic vocabulary --
    the descriptors and the vocabulary's construction parameters are stored too, so that the GPU test can run gfo_compute_bow on
    exactly this input and compare its fold with the reference's.
These are reference outputs for the fold only: the descent that produces a "voc" stream is the oracle's (unpinned), which is why
the stream itself is part of the fixture."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orb_oracle as O   # noqa: E402

WEIGHTINGS = {"TF_IDF": 0, "TF": 1, "IDF": 2, "BINARY": 3}
NORMS = {"none": 0, "L1": 1, "L2": 2}
VOC_CASES = ((0, 10, 3, 2000, 1), (1, 4, 5, 3500, 2), (2, 6, 2, 8192, 1), (3, 10, 4, 1, 5), (4, 10, 4, 77, 3))   # seed, k, depth, n, levelsup


def syn_stream(seed, n):
    rng = np.random.default_rng(7000 + seed)
    nwords = max(1, n // 6 + 1)
    word = rng.integers(0, nwords, n).astype(np.uint32) * np.uint32(3) + np.uint32(seed)
    table = np.log(rng.uniform(1.1, 400.0, nwords))
    table[rng.random(nwords) < 0.1] = 0.0
    weight = table[(word.astype(np.int64) - seed) // 3].astype(np.float64)
    if n >= 77:
        weight[5] = -1.25          # `w > 0` drops it
        weight[9] = 5e-324         # the smallest denormal is a weight
        weight[11] = np.nan        # NaN > 0 is false: stopped
    node = rng.integers(0, max(1, nwords // 4 + 1), n).astype(np.uint32)
    return word, weight, node


def voc_descriptors(voc, seed, n):
    rng = np.random.default_rng(100 + seed)
    leaves = voc["descriptors"][voc["n_children"] == 0]
    desc = leaves[rng.integers(0, len(leaves), n)].copy()
    flips = rng.integers(0, 256, (n, 5))
    for j in range(5):
        desc[np.arange(n), flips[:, j] >> 3] ^= (1 << (flips[:, j] & 7)).astype(np.uint8)
    return desc


def main():
    O.build()
    if O.build_ref() is None or not O.ref_available():
        raise SystemExit("oracle/_ref/libdbow2_fold.so cannot be built here (no reference tree): fixtures not regenerated")
    out = {}
    streams = {}
    for s, n in enumerate((0, 1, 77, 2000, 8192)):
        streams[f"syn{s}"] = syn_stream(s, n)
    for seed, k, depth, n, levelsup in VOC_CASES:
        voc = O.make_vocabulary(k, depth, seed=seed, p_stop=0.1)
        desc = voc_descriptors(voc, seed, n)
        streams[f"voc{seed}"] = O.bow_stream(voc, desc, levelsup)
        out[f"voc{seed}.desc"] = desc
        out[f"voc{seed}.params"] = np.array([seed, k, depth, n, levelsup], np.int32)
    for name, (word, weight, node) in streams.items():
        out[f"{name}.word"], out[f"{name}.weight"], out[f"{name}.node"] = word, weight, node
        for wn, w in WEIGHTINGS.items():
            for nn, nm in NORMS.items():
                bw, bv, fn, fs, fi = O.ref_bow_fold(word, weight, node, w, nm)
                pre = f"{name}.{wn}.{nn}."
                out[pre + "bow_words"], out[pre + "bow_values"] = bw, bv
                if w == 0 and nm == 0:       # the FeatureVector does not depend on weighting / norm: stored once per stream
                    out[f"{name}.fv_nodes"], out[f"{name}.fv_start"], out[f"{name}.fv_items"] = fn, fs, fi
                else:
                    assert (fn == out[f"{name}.fv_nodes"]).all() and (fs == out[f"{name}.fv_start"]).all() and (fi == out[f"{name}.fv_items"]).all()
    path = os.path.join(ROOT, "tests", "golden", "bow_fold.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {len(streams)} streams x {len(WEIGHTINGS) * len(NORMS)} (weighting, norm) cases, {os.path.getsize(path)} bytes, made by {O._REF_FOLD}")


if __name__ == "__main__":
    main()
