"""Golden vectors of the good-feature matchers and the keyframe-pair SearchByBoW, made by the oracle (oracle/orb_oracle.c; variant table
tests/golden/variants.json) on the committed EuRoC extraction and the map of EuRoC_projection.npz:
  SearchByProjection_Budget (ORBmatcher.cc:45-153) at Tracking.cc:2166's th = 0.5 and at th = 1 -- what every point did at its turn, the
  IncreaseFound() calls, the frame after a clock that trips at its fifth reading;
  GetCandidates (ORBmatcher.h:152-172) for every point at th = 1 as a CSR of keypoint indices;
  SearchByBoW(KF, KF) (ORBmatcher.cc:635-768) between the left and the right extraction with 64 pseudo-nodes.
usage: python tests/golden/make_gf_golden.py        -> tests/golden/EuRoC_gf_matchers.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import orb_oracle as O  # noqa: E402


def main():
    O.build()
    kd = O.KEYPOINT_DTYPE
    kl = np.fromfile(os.path.join(HERE, "EuRoC_l_kp.bin"), kd)
    kr = np.fromfile(os.path.join(HERE, "EuRoC_r_kp.bin"), kd)
    dl = np.fromfile(os.path.join(HERE, "EuRoC_l_desc.bin"), np.uint8).reshape(-1, 32)
    dr = np.fromfile(os.path.join(HERE, "EuRoC_r_desc.bin"), np.uint8).reshape(-1, 32)
    u = np.load(os.path.join(HERE, "EuRoC_stereo.npz"))["u_right"]
    p = np.load(os.path.join(HERE, "EuRoC_projection.npz"))
    mps, mpd, taken = p["mps"], p["mp_desc"], p["taken"]
    sf = O.OracleExtractor(2000, 1.2, 8, 20, 7).scale_factors
    b = (0.0, 0.0, 752.0, 480.0)
    out = {}
    for tag, th in (("th05", 0.5), ("th1", 1.0)):
        nm, out_mp, out_sc, out_pt, found = O.search_by_projection_budget(kl, dl, u, sf, b, mps, mpd, th, 0.8, taken, 0)
        out.update({f"{tag}_nmatches": nm, f"{tag}_out_mp": out_mp, f"{tag}_out_score": out_sc, f"{tag}_out_point": out_pt, f"{tag}_found": found})
        print(tag, "matches", nm, "codes", {c: int((out_pt == c).sum()) for c in (-1, -2, -3)})
    nm5, mp5, sc5, pt5, f5 = O.search_by_projection_budget(kl, dl, u, sf, b, mps, mpd, 1.0, 0.8, taken, 5)
    out.update({"th1_trip5_nmatches": nm5, "th1_trip5_out_mp": mp5, "th1_trip5_out_score": sc5})
    pf = O.ProjectionFrame(kl, dl, u, sf, b, None)
    lists = [pf.candidates(mps[i], 1.0) for i in range(len(mps))]
    out["th1_cand_start"] = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    out["th1_cand_idx"] = np.concatenate(lists).astype(np.int32) if lists else np.zeros(0, np.int32)
    print("candidate table entries", len(out["th1_cand_idx"]))
    rng = np.random.default_rng(17)
    v1 = (rng.random(len(kl)) > 0.15).astype(np.uint8)
    v2 = (rng.random(len(kr)) > 0.15).astype(np.uint8)
    n1 = (dl[:, 0] >> 2).astype(np.int64)
    n2 = (dr[:, 0] >> 2).astype(np.int64)
    for ori in (0, 1):
        nmk, o12 = O.search_by_bow_keyframes(dl, kl["angle"], v1, O.make_feature_vector(n1), dr, kr["angle"], v2, O.make_feature_vector(n2), 0.75, bool(ori))
        out[f"bowkf_ori{ori}_nmatches"] = nmk
        out[f"bowkf_ori{ori}_out12"] = o12
        print("SearchByBoW(KF, KF) ori", ori, "matches", nmk)
    out["bowkf_valid1"], out["bowkf_valid2"] = v1, v2
    np.savez_compressed(os.path.join(HERE, "EuRoC_gf_matchers.npz"), **out)


if __name__ == "__main__":
    main()
