#!/usr/bin/env python3
"""One-command pin of the oracle's [OCV] blocks against a real OpenCV.

    python tests/golden/check_against_cv2.py            # anywhere `import cv2` gives 3.4.x (CMakeLists.txt:116-125)

The reference's CPU extractor calls OpenCV 3.4.1 for five pieces of arithmetic (src/ORBextractor.cc:80,102,811-817,1155,
1189); oracle/orb_oracle.c restates them from the published sources and tags them [OCV].  This image has no
OpenCV, so every GPU parity test is green against that restatement only -- "parity unpinned".  This script is the pin for
whoever has cv2: it runs each [OCV] stage of the oracle and the corresponding cv2 call on the same inputs (the reference's
own test/EuRoC_l.png / EuRoC_r.png as committed raw dumps, plus two synthetic frames) and compares them bit for bit,
stage by stage, so that a mismatch names the block AND the variant switch of oracle/ocv_variants.json that removes it
(resize: fixed-point / float; Gaussian taps and rounding; fastAtan2 with or without fused Horner steps).

  cv2 absent                -> prints "cv2 absent -- parity unpinned" and exits 0 (nothing to compare; this container)
  every stage equal         -> prints PINNED and exits 0
  any stage differs         -> prints the per-stage table, the variant that does match (if one does), exits 1;
                               the fix is one line of oracle/ocv_variants.json, then `python tests/golden/make_golden.py`,
                               then the kernel that implements the stage follows (k_pyramid.hip / k_blur.hip /
                               include/gfo_sincos.h), with every GPU parity test as its check.

CPU only; never runs near the GPU box.  Not comparable and therefore not attempted: cv2.ORB (a different selection --
no cell grid, no quadtree) and cvRound (no Python binding; the oracle uses lrintf = round-half-even, the documented
SSE behaviour).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def load_inputs():
    imgs = {}
    for name in ("EuRoC_l", "EuRoC_r"):
        imgs[name] = np.fromfile(os.path.join(HERE, f"{name}_752x480.u8"), np.uint8).reshape(480, 752)
    try:
        from gf_orb_slam2_amd.synth import synth_frame
        imgs["synth752_0"] = synth_frame(752, 480, 0)
        imgs["synth1080_1"] = synth_frame(1920, 1080, 1)
    except Exception as ex:  # noqa: BLE001 -- the synthetic generator is a convenience here, the EuRoC pair is the pin
        print(f"(synthetic frames skipped: {ex!r})")
    return imgs


def diff_stats(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    if a.shape != b.shape:
        return {"equal": False, "shape": [list(a.shape), list(b.shape)]}
    ne = a != b
    return {"equal": not ne.any(), "differing": int(ne.sum()), "of": int(ne.size),
            "max_abs": int(np.abs(a.astype(np.int64) - b.astype(np.int64)).max()) if ne.any() else 0}


def KERNEL_LINE(cv):
    """the rebuild that makes the kernels follow a blur candidate (csrc/gfo_internal.h: GFO_OCV_BLUR_ROUND, GFO_GAUSS_TAPS = the outer
    three taps and the centre)"""
    fl = []
    if "blur_round" in cv:
        fl.append(f"-DGFO_OCV_BLUR_ROUND={cv['blur_round']}")
    if "gauss_taps" in cv:
        t = cv["gauss_taps"]
        fl.append(f"-DGFO_GAUSS_TAPS={t[0]},{t[1]},{t[2]},{t[3]}")
    return 'make -C gf-orb-slam2_amd/csrc -B EXTRA="' + " ".join(fl) + '"  (then tools/check_variant.py compares that build with the like-switched oracle)'


def main():
    try:
        import cv2
    except ImportError:
        print("cv2 absent -- parity unpinned (nothing compared; run this where OpenCV 3.4.x is importable)")
        return 0
    from oracle import orb_oracle as O
    O.build()
    committed = O.load_ocv_variants()
    ver = cv2.__version__
    exact_version = ver.startswith("3.4")
    print(f"cv2 {ver}" + ("" if exact_version else "  -- NOT 3.4.x: the comparison is informative, it does not pin the reference's build"))
    imgs = load_inputs()
    report = {"cv2": ver, "committed_variants": committed, "stages": {}}
    ok = True

    def stage(name, equal, detail, hint=None):
        nonlocal ok
        report["stages"][name] = {"equal": bool(equal), **detail, **({"hint": hint} if hint and not equal else {})}
        ok = ok and bool(equal)
        print(f"  [{'ok' if equal else 'DIFFERS'}] {name}" + ("" if equal else f"  {json.dumps(detail)}" + (f"\n        -> {hint}" if hint else "")))

    # ---- cv::resize(INTER_LINEAR), ORBextractor.cc:1189: level l from the ORACLE's level l-1 (errors must not compound) ----
    print("resize (INTER_LINEAR, u8):")
    for iname, img in imgs.items():
        oe = O.OracleExtractor(2000, 1.2, 8, 20, 7)
        O.set_ocv_variants(**committed)
        oe.compute_pyramid(img)
        per_variant = {}
        for variant in (0, 1):
            O.set_ocv_variants(resize=variant)
            bad = 0
            worst = 0
            for l in range(1, 8):
                src = oe.level(l - 1)
                w, h = oe.level_size(l)
                d = diff_stats(O.resize_linear(src, w, h), cv2.resize(src, (w, h), interpolation=cv2.INTER_LINEAR))
                bad += d.get("differing", 1)
                worst = max(worst, d.get("max_abs", 255))
            per_variant[variant] = (bad, worst)
        O.set_ocv_variants(**committed)
        cur = committed["resize"]
        match = [v for v, (b, _) in per_variant.items() if b == 0]
        stage(f"resize/{iname}", per_variant[cur][0] == 0,
              {"differing_px_levels_1_7": per_variant[cur][0], "max_abs": per_variant[cur][1], "per_variant": {str(k): v[0] for k, v in per_variant.items()}},
              (f'set "resize": {match[0]} in oracle/ocv_variants.json; kernels: make -C gf-orb-slam2_amd/csrc -B EXTRA="-DGFO_OCV_RESIZE={match[0]}"') if match else "no variant of the table matches: restate cv::resize from this build's sources")

    # ---- copyMakeBorder(BORDER_REFLECT_101), ORBextractor.cc:1191-1197 ----
    print("copyMakeBorder (REFLECT_101, 19 px):")
    for iname, img in list(imgs.items())[:2]:
        oe = O.OracleExtractor(2000, 1.2, 8, 20, 7)
        oe.compute_pyramid(img)
        for l in (0, 7):
            lv = oe.level(l)
            d = diff_stats(oe.level(l, padded=True), cv2.copyMakeBorder(lv, 19, 19, 19, 19, cv2.BORDER_REFLECT_101))
            stage(f"border/{iname}/L{l}", d["equal"], d)

    # ---- GaussianBlur(7x7, sigma 2, REFLECT_101) on a clone of the level, ORBextractor.cc:1154-1155 ----
    print("GaussianBlur (7x7, sigma 2, u8):")
    k = cv2.getGaussianKernel(7, 2.0)
    print(f"  cv2.getGaussianKernel(7, 2) * 256 = {[round(float(x) * 256, 3) for x in k.ravel()]}")
    candidates = [("committed", {})] + [(f"round={r} centre={c}", {"blur_round": r, "gauss_taps": [18, 34, 49, c, 49, 34, 18]})
                                         for r in (0, 1) for c in (55, 54, 56)]
    for iname, img in imgs.items():
        oe = O.OracleExtractor(2000, 1.2, 8, 20, 7)
        oe.compute_pyramid(img)
        results = {}
        for cname, cv in candidates:
            O.set_ocv_variants(**committed)
            O.set_ocv_variants(**cv)
            bad = 0
            for l in (0, 3, 7):
                lv = oe.level(l)
                bad += diff_stats(O.gaussian_blur7(lv), cv2.GaussianBlur(lv.copy(), (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)).get("differing", 1)
            results[cname] = bad
        O.set_ocv_variants(**committed)
        match = [c for c, b in results.items() if b == 0]
        stage(f"blur/{iname}", results["committed"] == 0, {"differing_px": results["committed"], "per_candidate": results},
              (f"matching candidate: {match[0]} -> set blur_round / gauss_taps accordingly in oracle/ocv_variants.json; kernels: " + KERNEL_LINE(dict(candidates)[match[0]])) if match else
              "no candidate matches: print cv2.getGaussianKernel above against the taps, and the fixed-point path of this build")

    # ---- cv::FAST(threshold, nonmaxSuppression = true), ORBextractor.cc:811-817, on a whole level and on cell-sized crops ----
    print("FAST 9/16 + NMS:")
    ftype = getattr(cv2, "FAST_FEATURE_DETECTOR_TYPE_9_16", None)
    if ftype is None:
        ftype = cv2.FastFeatureDetector_TYPE_9_16
    for iname, img in list(imgs.items())[:3]:
        for t in (20, 7):
            det = cv2.FastFeatureDetector_create(threshold=t, nonmaxSuppression=True, type=ftype)
            views = [("full", img)] + [(f"cell{j}", np.ascontiguousarray(img[16 + 32 * j:16 + 32 * j + 38, 100 + 30 * j:100 + 30 * j + 36])) for j in range(6)]
            for vname, v in views:
                got = sorted((int(round(kp.pt[0])), int(round(kp.pt[1])), int(round(kp.response))) for kp in det.detect(v, None))
                ref = sorted(map(tuple, O.fast9_nms(v, t).tolist()))
                if got != ref:
                    both = len(set(got) & set(ref))
                    stage(f"fast/{iname}/t{t}/{vname}", False, {"cv2": len(got), "oracle": len(ref), "common": both},
                          "positions equal but responses differ -> the score definition; positions differ -> the arc test or the NMS")
                    break
            else:
                stage(f"fast/{iname}/t{t}", True, {"views": len(views)})

    # ---- cv::fastAtan2, ORBextractor.cc:102: the moments of real keypoints and a random sweep, bit patterns ----
    print("fastAtan2:")
    rng = np.random.default_rng(20260403)
    y = np.concatenate([rng.integers(-2 ** 21, 2 ** 21, 200000), rng.integers(-50, 50, 20000), [0, 0, 1, -1, 0]]).astype(np.float32)
    x = np.concatenate([rng.integers(-2 ** 21, 2 ** 21, 200000), rng.integers(-50, 50, 20000), [0, 1, 0, 0, -1]]).astype(np.float32)
    got_vec = cv2.phase(x, y, angleInDegrees=True).ravel().astype(np.float32)          # hal::fastAtan32f, the routine fastAtan2 calls
    got_scalar = np.array([cv2.fastAtan2(float(a), float(b)) for a, b in zip(y[:5000], x[:5000])], np.float32)
    res = {}
    for variant in (0, 1):
        O.set_ocv_variants(atan_fma=variant)
        mine = O.fast_atan2_n(y, x)
        res[variant] = (int((mine.view(np.uint32) != got_vec.view(np.uint32)).sum()), int((mine[:5000].view(np.uint32) != got_scalar.view(np.uint32)).sum()),
                        float(np.abs(mine - got_vec).max()))
    O.set_ocv_variants(**committed)
    cur = committed["atan_fma"]
    match = [v for v, r in res.items() if r[1] == 0]
    stage("fastAtan2/scalar", res[cur][1] == 0, {"differing_of_5000": res[cur][1], "per_variant": {str(k): v[1] for k, v in res.items()}},
          (f'set "atan_fma": {match[0]} in oracle/ocv_variants.json; kernels: make -C gf-orb-slam2_amd/csrc -B EXTRA="-DGFO_OCV_ATAN_FMA={match[0]}"') if match else "neither Horner form matches: check the coefficients and the 90/180/360 folding")
    stage("fastAtan2/vector(cv2.phase)", res[cur][0] == 0, {"differing_of": [res[cur][0], len(y)], "max_abs_deg": res[cur][2],
                                                            "per_variant": {str(k): v[0] for k, v in res.items()}},
          "the array routine (SIMD) and the scalar call may legitimately differ in this build: the scalar line above is the call site's")

    report["pinned"] = bool(ok and exact_version)
    out = os.path.join(HERE, "cv2_pin_report.json")
    with open(out, "w") as f:
        json.dump(report, f, indent=1)
    print(f"report: {out}")
    if ok:
        print(("PINNED: " if exact_version else "EQUAL (but cv2 is not 3.4.x): ") + f"every [OCV] block of oracle/orb_oracle.c equals cv2 {ver} on these inputs")
        return 0
    print("NOT PINNED: see the stages marked DIFFERS above; each names its switch in oracle/ocv_variants.json")
    return 1


if __name__ == "__main__":
    sys.exit(main())
