"""Every pixel of every blurred level against the oracle (cv::GaussianBlur 7x7, sigma 2, BORDER_REFLECT_101 on a clone of the level,
ORBextractor.cc:1154-1155), for BATCHES -- the path that runs the blur on the matrix cores (k_blur_mfma: band-matrix products in i8
MFMA, border reflection folded into the band) -- and for the streaming form of the same launch (GFO_BLUR_MFMA=0).  The extraction
tests compare descriptors (512 samples of the blurred plane per keypoint); this one compares the planes."""
import numpy as np
import pytest

from conftest import synth_frame

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("form", ["matrix_cores", "streaming"])
@pytest.mark.parametrize("w,h,nl,sf", [(752, 480, 8, 1.2), (753, 481, 8, 1.2), (640, 479, 6, 1.3), (1241, 376, 8, 1.2), (1920, 1080, 4, 1.5),
                                       (131, 97, 3, 1.2), (64, 40, 2, 1.1), (200, 150, 8, 1.2)])
def test_blurred_planes_of_a_batch(oracle, monkeypatch, form, w, h, nl, sf):
    """Widths that are and are not multiples of 64 / 16 / 4, a level-0 pitch equal to the width, levels of a few dozen pixels (the last case
    has levels narrower than 64 px: the whole launch takes the streaming form), twelve images = beyond the per-frame (fused) path."""
    import gf_orb_slam2_amd as G
    monkeypatch.setenv("GFO_BLUR_MFMA", "1" if form == "matrix_cores" else "0")
    imgs = [synth_frame(w, h, i) for i in range(12)]
    imgs[3] = np.full((h, w), 255, np.uint8)            # a saturated plane: the sum 257 x 257 x 255 clamps to 255
    imgs[5] = np.zeros((h, w), np.uint8)
    rng = np.random.default_rng(w * 31 + h)
    imgs[7] = rng.integers(0, 256, (h, w), dtype=np.uint8)   # noise: every border pixel differs from its neighbours
    ext = G.ORBextractor(500, sf, nl, 20, 7)
    try:
        ext.extract_batch(imgs)
        for i in (0, 3, 5, 7, 11):
            oe = oracle.OracleExtractor(500, sf, nl, 20, 7)
            oe(imgs[i])
            for l in range(nl):
                got = ext.debug_blurred_level(l, image=i)
                # (the reference -- and the oracle's extractor -- blur only the levels that hold keypoints, ORBextractor.cc:1150-1155; the
                #  device blurs every level of a batch: each is compared with the oracle's blur of the oracle's own level)
                want = oracle.gaussian_blur7(oe.level(l))
                assert got.shape == want.shape
                np.testing.assert_array_equal(got, want, err_msg=f"image {i} level {l} ({form})")
                if oe.level_keypoint_count(l) > 0:
                    np.testing.assert_array_equal(oe.level(l, blurred=True), want)
    finally:
        ext.close()
