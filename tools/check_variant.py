#!/usr/bin/env python3
"""Checks ONE build of libgfo.so against the oracle set to the SAME [OCV] switches (oracle/ocv_variants.json's keys; include/gfo.h
gfo_build_variant): pyramid levels, FAST candidate sets, blurred planes, keypoints (angle by bit pattern) and descriptors of the
reference's EuRoC image, a synthetic frame and the per-frame (batch <= 8) and batched (32 images) code paths.

    GFO_LIB=gf-orb-slam2_amd/variants/libgfo_resize1.so python tools/check_variant.py        # one variant build
    python tools/check_variant.py                                                             # the product build (all zeros)

Prints one JSON line: the switches the library reports, whether every stage equals the like-switched oracle, and how many level
pixels / blurred pixels / angles / descriptor bytes differ from the DEFAULT oracle (a variant that changes nothing was not built).
Test infrastructure (it imports the oracle); tests/test_gpu_ocv_variants.py runs it once per library of `make -C csrc variants`."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd._lib import load_library
    from gf_orb_slam2_amd.synth import synth_frame
    from oracle import orb_oracle as O
    L = load_library()
    L.gfo_build_variant.argtypes = [__import__("ctypes").c_int]
    sw = {"resize": L.gfo_build_variant(0), "atan_fma": L.gfo_build_variant(1), "blur_round": L.gfo_build_variant(2),
          "gauss_taps": [L.gfo_build_variant(3 + k) for k in range(7)]}
    gold = os.path.join(ROOT, "tests", "golden")
    euroc = np.fromfile(os.path.join(gold, "EuRoC_l_752x480.u8"), np.uint8).reshape(480, 752)
    images = [euroc, synth_frame(752, 480, 3), np.ascontiguousarray(euroc[40:341, 100:521])]
    out = {"lib": os.environ.get("GFO_LIB") or "libgfo.so", "switches": sw, "equal": True, "first_mismatch": None}

    def default_products(img):
        O.set_ocv_variants(resize=0, atan_fma=0, blur_round=0, gauss_taps=[18, 34, 49, 55, 49, 34, 18])
        oe = O.OracleExtractor(2000, 1.2, 8, 20, 7)
        kp, d = oe(img)
        return [oe.level(l) for l in range(8)], [oe.level(l, blurred=True) for l in range(8)], kp, d

    diff = {"level_px": 0, "blur_px": 0, "angles": 0, "desc_bytes": 0, "keypoints": 0}
    for ii, img in enumerate(images):
        dl, db, dkp, dd = default_products(img)
        O.set_ocv_variants(**sw)
        oe = O.OracleExtractor(2000, 1.2, 8, 20, 7)
        okp, od = oe(img)
        h, w = img.shape
        for batch in (1, 32):
            ext = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=batch)
            if batch == 1:
                gkp, gd = ext(img)
            else:
                kps, ds = ext.extract_batch([img] * batch)
                gkp, gd = kps[batch - 1], ds[batch - 1]

            def bad(what):
                if out["equal"]:
                    out["equal"] = False
                    out["first_mismatch"] = f"image {ii} batch {batch}: {what}"
            for l in range(8):
                if not np.array_equal(ext.pyramid_level(l, image=batch - 1) if batch > 1 else ext.pyramid_level(l), oe.level(l)):
                    bad(f"pyramid level {l}")
            if batch == 1:
                for l in range(8):
                    if sorted(map(tuple, ext.debug_level_candidates(l).tolist())) != sorted(map(tuple, oe.level_candidates(l).tolist())):
                        bad(f"FAST candidates level {l}")
                    if oe.level_keypoint_count(l) > 0 and not np.array_equal(ext.debug_blurred_level(l), oe.level(l, blurred=True)):
                        bad(f"blurred level {l}")
            if len(gkp) != len(okp) or gkp.tobytes() != okp.tobytes():
                bad("keypoints")
            elif not np.array_equal(gd, od):
                bad("descriptors")
            ext.close()
        for l in range(8):
            diff["level_px"] += int((oe.level(l) != dl[l]).sum())
            diff["blur_px"] += int((oe.level(l, blurred=True) != db[l]).sum())
        if len(okp) == len(dkp) and np.array_equal(okp["x"], dkp["x"]) and np.array_equal(okp["y"], dkp["y"]):
            diff["angles"] += int((okp["angle"].view(np.uint32) != dkp["angle"].view(np.uint32)).sum())
            diff["desc_bytes"] += int((od != dd).sum())
        else:
            diff["keypoints"] += 1
    O.set_ocv_variants(resize=0, atan_fma=0, blur_round=0, gauss_taps=[18, 34, 49, 55, 49, 34, 18])
    out["differs_from_default_oracle"] = diff
    print(json.dumps(out))
    return 0 if out["equal"] else 1


if __name__ == "__main__":
    sys.exit(main())
