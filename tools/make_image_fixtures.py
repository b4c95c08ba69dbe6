#!/usr/bin/env python3
"""Dump the reference's 8-bit test images (test/EuRoC_l.png, test/EuRoC_r.png,
752x480 grey) as raw row-major u8 planes under tests/golden/.

The PNGs are test *data* of the reference, not source; the raw dumps are what the
C oracle and the GPU tests read (no PNG decoder needed on the GPU box).
Run once in the build container:  python tools/make_image_fixtures.py
"""
import os
import sys
import numpy as np
from PIL import Image

REF = "/root/reference/test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    for name in ("EuRoC_l", "EuRoC_r"):
        im = Image.open(os.path.join(REF, name + ".png"))
        assert im.mode == "L", im.mode
        a = np.asarray(im, dtype=np.uint8)
        assert a.shape == (480, 752), a.shape
        dst = os.path.join(OUT, f"{name}_752x480.u8")
        a.tofile(dst)
        print(dst, a.shape, int(a.min()), int(a.max()))


if __name__ == "__main__":
    sys.exit(main())
