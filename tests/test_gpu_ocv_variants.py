"""The [OCV] switch table on BOTH sides.  The oracle restates four pieces of OpenCV 3.4.1 arithmetic from its
published sources behind one switch each (oracle/ocv_variants.json; ORBextractor.cc:102 fastAtan2, :1155 GaussianBlur, :1189 resize).
The kernels follow the same switches at compile time (csrc/gfo_internal.h, `make -C gf-orb-slam2_amd/csrc variants`): whoever runs
tests/golden/check_against_cv2.py against a real cv2 3.4.x and is told "turn resize to 1" rebuilds -- no kernel is rewritten by hand.

Each variant library (ONE switch turned) must equal the oracle set to the same switches, stage by stage, through the per-frame and
the batched code paths; and the switch must actually change bytes (a variant that is a no-op was not built).  One subprocess per
library: a process loads one libgfo.so."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

VARDIR = os.path.join(ROOT, "gf-orb-slam2_amd", "variants")
DEFAULT_TAPS = [18, 34, 49, 55, 49, 34, 18]
VARIANTS = {
    "resize1": ({"resize": 1}, "level_px"),
    "atanfma1": ({"atan_fma": 1}, None),               # a fused Horner step moves an angle by an ulp on a few keypoints, or on none
    "blurround1": ({"blur_round": 1}, "blur_px"),
    "taps256": ({"gauss_taps": [18, 34, 49, 54, 49, 34, 18]}, "blur_px"),
}


def _run(lib):
    env = dict(os.environ)
    if lib:
        env["GFO_LIB"] = lib
    else:
        env.pop("GFO_LIB", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_variant.py")], capture_output=True, text=True, timeout=600, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines, p.stderr[-3000:]
    return p.returncode, json.loads(lines[-1]), p.stderr


def test_product_build_is_the_default_variant():
    rc, j, err = _run(None)
    assert j["switches"] == {"resize": 0, "atan_fma": 0, "blur_round": 0, "gauss_taps": DEFAULT_TAPS}
    assert rc == 0 and j["equal"], (j, err[-2000:])
    assert all(v == 0 for v in j["differs_from_default_oracle"].values())


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variant_build_equals_the_like_switched_oracle(name):
    lib = os.path.join(VARDIR, f"libgfo_{name}.so")
    if not os.path.exists(lib):
        pytest.skip(f"{lib} is not built (make -C gf-orb-slam2_amd/csrc variants; __graft_entry__.build() does)")
    want, must_change = VARIANTS[name]
    rc, j, err = _run(lib)
    expect = {"resize": 0, "atan_fma": 0, "blur_round": 0, "gauss_taps": DEFAULT_TAPS}
    expect.update(want)
    assert j["switches"] == expect, j
    assert rc == 0 and j["equal"], (j["first_mismatch"], err[-2000:])
    d = j["differs_from_default_oracle"]
    if must_change:
        assert d[must_change] > 1000, d                  # the switch is live: thousands of bytes move
    keep = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(keep):
        with open(os.path.join(keep, f"ocv_variant_{name}.json"), "w") as fh:
            json.dump(j, fh)
