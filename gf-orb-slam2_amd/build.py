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
