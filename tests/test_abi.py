"""CPU tests of the boundary: libgfo.so loads, exports every symbol include/gfo.h declares, and
fails loudly (no CPU fallback) when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "gfo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gfo_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def G():
    import gf_orb_slam2_amd as G
    if not os.path.exists(G.lib_path()):
        G.build_library()
    return G


def test_library_exports_every_declared_symbol(G):
    lib = ctypes.CDLL(G.lib_path())
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gfo.h but not exported"
    from gf_orb_slam2_amd._lib import SYMBOLS
    assert sorted(SYMBOLS) == syms


def test_variant_libraries_export_the_same_symbols(G):
    """The [OCV]-switch builds (gf-orb-slam2_amd/variants, `make variants`) are the product library with other compile-time switches: a
    variant left over from before an entry point was added would fail to load on the GPU box, so it is caught here."""
    import glob
    libs = sorted(glob.glob(os.path.join(ROOT, "gf-orb-slam2_amd", "variants", "libgfo_*.so")))
    if not libs:
        pytest.skip("variants not built (G.build_variants())")
    for path in libs:
        lib = ctypes.CDLL(path)
        for s in header_symbols():
            assert hasattr(lib, s), f"{os.path.basename(path)} lacks {s}: rebuild with `make -C gf-orb-slam2_amd/csrc variants`"


def test_keypoint_layout_matches_cv_keypoint(G):
    assert G.KEYPOINT_DTYPE.itemsize == 28
    assert [G.KEYPOINT_DTYPE.fields[n][1] for n in ("x", "y", "size", "angle", "response", "octave", "class_id")] == [0, 4, 8, 12, 16, 20, 24]


def test_hamming_host_helper(G):
    rng = np.random.default_rng(0)
    for _ in range(50):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert G.ORBmatcher.DescriptorDistance(a, b) == int(np.unpackbits(a ^ b).sum())


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "gf-orb-slam2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "orb_oracle" not in txt and "oracle/" not in txt.replace("oracle/ ", ""), f"{f} references the oracle"


def test_no_gpu_means_loud_failure(G):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(G.GfoError) as ei:
        G.ORBextractor(2000, 1.2, 8, 20, 7)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_bad_parameters_rejected_before_touching_the_device(G):
    from gf_orb_slam2_amd._lib import Params, load_library
    L = load_library()
    ctx = ctypes.c_void_p()
    for bad in (Params(2000, 1.2, 0, 20, 7, 1), Params(2000, 1.2, 17, 20, 7, 1), Params(0, 1.2, 8, 20, 7, 1),
                Params(2000, 1.0, 8, 20, 7, 1), Params(2000, 1.2, 8, 0, 7, 1), Params(2000, 1.2, 8, 20, 300, 1)):
        assert L.gfo_ctx_create(ctypes.byref(bad), 0, ctypes.byref(ctx)) == -1
        assert not ctx.value
    assert L.gfo_ctx_create(None, 0, ctypes.byref(ctx)) == -1


def test_one_hip_runtime_is_mapped():
    """load_library() pre-loads PyTorch's bundled HIP/HSA runtime and then verifies that libgfo.so did not bring in a
    second copy (two runtimes in one process stall or lose the GPU); any import order must end with one of each."""
    import gf_orb_slam2_amd as G
    from gf_orb_slam2_amd._lib import mapped_hip_runtimes
    G.load_library()
    import torch  # noqa: F401  (after libgfo, the order a lazy importer gets)
    m = mapped_hip_runtimes()
    assert len(m["libamdhip64"]) == 1, m
    assert len(m["libhsa-runtime64"]) <= 1, m
