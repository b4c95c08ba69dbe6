"""The C++ adapters (gf-orb-slam2_amd/adapter/*.cc) must compile against the reference's
UNCHANGED include/ORBextractor.h -- that is what "Frame.cc and Tracking.cc link unchanged" rests on.
OpenCV is not in this image, so the check is syntax-only against a tiny type stand-in
(tests/cv_standin/); it runs only where the reference tree is mounted (the build container)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

REF_INC = "/root/reference/include"


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_INC, "ORBextractor.h")), reason="reference headers not mounted")
@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")
def test_extractor_adapter_matches_reference_header():
    src = os.path.join(ROOT, "gf-orb-slam2_amd", "adapter", "ORBextractor_gfo.cc")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-D__SSE2__", "-I", os.path.join(ROOT, "tests", "cv_standin"),
           "-I", REF_INC, "-I", os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_INC, "ORBmatcher.h")), reason="reference headers not mounted")
@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ missing")
@pytest.mark.parametrize("guard", ["GFO_ADAPTER_ALL", "GFO_ADAPTER_STEREO", "GFO_ADAPTER_PROJECTION", "GFO_ADAPTER_PROJ_LAST",
                                   "GFO_ADAPTER_PROJ_KF", "GFO_ADAPTER_BOW", "GFO_ADAPTER_COMPUTE_BOW"])
def test_matcher_adapters_match_reference_headers(guard):
    """adapter/matchers_gfo.cc defines Frame:: / ORBmatcher:: members with the reference's own signatures: it must
    parse against the UNCHANGED include/Frame.h, ORBmatcher.h, KeyFrame.h, MapPoint.h and the vendored DBoW2 headers
    (every member it reads or writes exists there with a compatible type), each swap guard on its own and all together.
    Syntax only: OpenCV / Armadillo are replaced by declaration stand-ins (tests/cv_standin/), nothing is linked."""
    src = os.path.join(ROOT, "gf-orb-slam2_amd", "adapter", "matchers_gfo.cc")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-D__SSE2__", "-D" + guard, "-I", os.path.join(ROOT, "tests", "cv_standin"),
           "-I", REF_INC, "-I", os.path.dirname(REF_INC), "-I", os.path.join(ROOT, "include"), src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
