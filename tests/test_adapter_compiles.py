"""The C++ adapter (gf-orb-slam2_amd/adapter/ORBextractor_gfo.cc) must compile against the reference's
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
