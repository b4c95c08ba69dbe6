"""Builds libgfo.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_library(force=False, jobs=8):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(jobs)]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    out = os.path.join(_HERE, "libgfo.so")
    if not os.path.exists(out):
        raise RuntimeError("libgfo.so was not produced")
    return out


def build_variants(jobs=8):
    """The [OCV] variant builds (csrc/Makefile `variants`, include/gfo.h gfo_build_variant): libgfo.so with one switch of
    the checker's table (ocv_variants.json) turned each, into gf-orb-slam2_amd/variants/ -- what tests/test_gpu_ocv_variants.py compares with the
    like-switched oracle.  Never loaded by the product (GFO_LIB selects a library explicitly)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "variants"])
