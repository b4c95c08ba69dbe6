"""GPU parity tests of the extractor: HIP path (through the C ABI) vs the CPU oracle,
bit-exact (integer / byte / index work; float fields compared by bit pattern)."""
import numpy as np
import pytest

from conftest import synth_frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext():
    import gf_orb_slam2_amd as G
    e = G.ORBextractor(2000, 1.2, 8, 20, 7, max_batch=4)
    yield e
    e.close()


def _cmp_extract(ext, oracle, img, nfeatures=2000):
    oe = oracle.OracleExtractor(nfeatures, 1.2, 8, 20, 7)
    okp, odesc = oe(img)
    gkp, gdesc = ext(img)
    # stage-by-stage first, so a failure names the stage
    for l in range(8):
        np.testing.assert_array_equal(ext.pyramid_level(l), oe.level(l), err_msg=f"pyramid level {l}")
    for l in range(8):
        oc = oe.level_candidates(l)
        gc = ext.debug_level_candidates(l)
        oset = sorted(map(tuple, oc.tolist()))
        gset = sorted(map(tuple, gc.tolist()))
        assert gset == oset, f"FAST candidates differ at level {l}: {len(gset)} vs {len(oset)}"
    for l in range(8):
        if oe.level_keypoint_count(l) > 0:
            np.testing.assert_array_equal(ext.debug_blurred_level(l), oe.level(l, blurred=True), err_msg=f"blur level {l}")
    assert len(gkp) == len(okp)
    for f in ("octave", "class_id"):
        np.testing.assert_array_equal(gkp[f], okp[f], err_msg=f)
    for f in ("x", "y", "size", "response", "angle"):
        np.testing.assert_array_equal(gkp[f].view(np.uint32), okp[f].view(np.uint32), err_msg=f)
    np.testing.assert_array_equal(gdesc, odesc)
    return len(gkp)


def test_euroc_left_bit_exact(ext, oracle, euroc_l):
    n = _cmp_extract(ext, oracle, euroc_l)
    assert n >= 1900


def test_euroc_right_bit_exact(ext, oracle, euroc_r):
    _cmp_extract(ext, oracle, euroc_r)


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_synthetic_752x480_bit_exact(ext, oracle, idx):
    _cmp_extract(ext, oracle, synth_frame(752, 480, idx))


def test_batch_matches_single(ext, oracle, euroc_l, euroc_r):
    imgs = [euroc_l, euroc_r, synth_frame(752, 480, 5), euroc_l]
    kps, descs = ext.extract_batch(imgs)
    for im, k, d in zip(imgs, kps, descs):
        oe = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)
        okp, odesc = oe(im)
        assert k.tobytes() == okp.tobytes()
        np.testing.assert_array_equal(d, odesc)


def test_two_contexts_from_two_threads(oracle, euroc_l, euroc_r):
    """The reference runs its left and right extractor objects from two threads at once (Frame.cc:84-87): two
    contexts driven concurrently (ctypes releases the GIL during the calls) stay bit-exact, 30 frames each."""
    import threading
    import gf_orb_slam2_amd as G
    refs = [oracle.OracleExtractor(2000, 1.2, 8, 20, 7)(im) for im in (euroc_l, euroc_r)]
    exts = [G.ORBextractor(2000, 1.2, 8, 20, 7) for _ in range(2)]
    errors = []

    def work(i, img):
        try:
            for _ in range(30):
                k, d = exts[i](img)
                if k.tobytes() != refs[i][0].tobytes() or not (d == refs[i][1]).all():
                    errors.append(f"thread {i}: result differs")
                    return
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e!r}")

    ts = [threading.Thread(target=work, args=(i, im)) for i, im in enumerate((euroc_l, euroc_r))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for e in exts:
        e.close()
    assert not errors, errors


def test_kernels_are_resolved_before_any_thread_launches_and_eight_first_frames_at_once():
    """gfo_ctx_create resolves every kernel of the library once per device under a mutex (gfo_kernels_preloaded), so that threads
    issuing their FIRST frames at the same moment never race through the runtime's lazy code-object loading (round 3: eight threads,
    first k_pack_results launch, an abort under rocprofv3 -- profiles/boundary_trace_r04.txt).  In a FRESH process (this one has
    launched everything long ago): the count is complete after the first context and does not move with use; then the C harness'
    own scenario, eight threads x gfo_extract_stereo from a cold start, combiner off, checksums equal."""
    import os
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    if os.environ.get("GFO_PRELOAD", "1") == "0":
        pytest.skip("GFO_PRELOAD=0: the lazy behaviour was asked for")
    code = ("import sys; sys.path.insert(0, %r)\nimport numpy as np\nimport gf_orb_slam2_amd as G\n"
            "L = G.load_library()\nassert L.gfo_kernels_preloaded() == 0\n"
            "e = G.ORBextractor(500, 1.2, 8, 20, 7)\nn = L.gfo_kernels_preloaded()\nassert n >= 30, n\n"
            "e2 = G.ORBextractor(700, 1.2, 8, 20, 7)\nassert L.gfo_kernels_preloaded() == n\n"
            "im = np.random.default_rng(0).integers(0, 256, (240, 320), dtype=np.uint8)\nk, d = e(im); k2, d2 = e2(im)\n"
            "assert L.gfo_kernels_preloaded() == n and len(k) > 0\nprint('PRELOADED', n)\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=170)
    assert r.returncode == 0 and "PRELOADED" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler for the harness")
    exe = "/tmp/gfo_bt_first_frames"
    subprocess.run([cc, "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "c", "boundary_throughput.c"), "-o", exe,
                    "-ldl", "-lpthread", "-lm"], check=True, capture_output=True, timeout=120)
    import json
    from gf_orb_slam2_amd._lib import lib_path
    r = subprocess.run([exe, lib_path(), os.path.join(ROOT, "tests", "golden"), "0.3", "stereo", "8", "0"], capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    p = json.loads(r.stdout)["points"][0]
    assert p["streams"] == 8 and p["errors"] == 0 and p["result_mismatches"] == 0 and p["stereo_frames"] > 100, p


def test_two_threads_replanning_against_each_other(oracle):
    """The reference runs its left and right extractor on two host threads (Frame.cc:84-87): one thread may be planning
    an arena (hipMalloc / hipMemcpy) while the other is mid-pipeline.  Both alternate between two image sizes here, so
    that every call re-plans: results stay bit-exact and nothing fails.  (This is the test that showed hipGraph capture
    cannot be on by default: with GFO_GRAPH=1 the planning thread's hipMemcpy fails while the other thread captures.)"""
    import threading
    import gf_orb_slam2_amd as G
    imgs = [synth_frame(400, 300, 5), synth_frame(336, 256, 6)]
    oe = oracle.OracleExtractor(600, 1.2, 8, 20, 7)
    refs = [oe(im) for im in imgs]
    exts = [G.ORBextractor(600, 1.2, 8, 20, 7) for _ in range(2)]
    errors = []

    def work(t):
        try:
            for it in range(40):
                k = (it + t) & 1
                kp, d = exts[t](imgs[k])
                if kp.tobytes() != refs[k][0].tobytes() or not (d == refs[k][1]).all():
                    errors.append(f"thread {t} iteration {it}: result differs")
                    return
        except Exception as ex:  # noqa: BLE001
            errors.append(f"thread {t}: {ex!r}")

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    for e in exts:
        e.close()
    assert not errors, errors


def test_graph_replay_opt_in(oracle, euroc_l):
    """GFO_GRAPH=1: the per-frame launch sequence is captured once per geometry and replayed (single HIP thread only,
    see gfo_api.hip); same bits as the plain launches"""
    import os
    import gf_orb_slam2_amd as G
    os.environ["GFO_GRAPH"] = "1"
    try:
        ext = G.ORBextractor(2000, 1.2, 8, 20, 7)
    finally:
        del os.environ["GFO_GRAPH"]
    ok, od = oracle.OracleExtractor(2000, 1.2, 8, 20, 7)(euroc_l)
    for _ in range(3):
        kp, d = ext(euroc_l)
        assert kp.tobytes() == ok.tobytes() and (d == od).all()
    ext.close()


def test_chained_contexts_give_the_same_results_and_survive_destruction(oracle):
    """gfo_ctx_chain: an ordering edge between two contexts' submissions changes no result, refuses nonsense, and an edge
    to a destroyed context is gone rather than dangling."""
    import gf_orb_slam2_amd as G
    imgs = [synth_frame(320, 240, 900 + i) for i in range(4)]
    a = G.ORBextractor(500, 1.2, 8, 20, 7, max_batch=2)
    b = G.ORBextractor(500, 1.2, 8, 20, 7, max_batch=2)
    ref = [oracle.OracleExtractor(500, 1.2, 8, 20, 7)(im) for im in imgs]
    for stage in (1, 2, 3, 4):
        a.chain_after(b, stage)
        b.chain_after(a, stage)
        for rep in range(3):                       # alternate, as a pipelined application does
            ka, da = a.extract_batch(imgs[:2])
            kb, db = b.extract_batch(imgs[2:])
            for i in range(2):
                assert ka[i].tobytes() == ref[i][0].tobytes() and (da[i] == ref[i][1]).all()
                assert kb[i].tobytes() == ref[2 + i][0].tobytes() and (db[i] == ref[2 + i][1]).all()
    with pytest.raises(G.GfoError):
        a.chain_after(a, 1)
    with pytest.raises(G.GfoError):
        a.chain_after(b, 9)
    b.close()                                      # a's edge to b must not dangle
    ka, da = a.extract_batch(imgs[:2])
    assert ka[0].tobytes() == ref[0][0].tobytes()
    a.chain_after(None)
    a.close()
