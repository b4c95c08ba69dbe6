"""AddressSanitizer + UndefinedBehaviorSanitizer over the host code, on the CPU (sanitizers cannot run on the GPU
pool).  Four jobs, each a build with -fsanitize=address,undefined -fno-sanitize-recover=undefined and a run whose stderr must hold no
report:
  1. the ORACLE (oracle/orb_oracle.c) -- the checker every parity claim rests on: tests/test_oracle.py and a 2000-case slice of the fuzz
     generators (tests/san/oracle_fuzz_slice.py) in a Python with the sanitizer runtime preloaded (oracle/Makefile `san`);
  2. the adapter's context table (adapter/gfo_context_table.h) with its counting stand-in (tests/host/context_table_check.cc);
  3. the frame combiner and the stereo rigs (csrc/gfo_combine.hip, unmodified) on the fake backend of the ThreadSanitizer job;
  4. the ABI's own host code (csrc/gfo_api.hip + gfo_combine.hip, unmodified) on a fake HIP runtime whose "device" memory is exact-size
     host memory (tests/host/fakehip, tests/host/gfo_api_san.cc): argument checks, plan(), staging blocks, capacities, delivery.
Round 6's first runs found: the oracle copying a blurred plane that was never made (a level without keypoints) and reflecting into a
level of zero pixels; gfo_extract_batch taking a negative capacity (image i's arrays are kp + i * cap).  All three are fixed, not
suppressed."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]


def _need_sanitizers(tmp_path):
    if shutil.which("g++") is None:
        pytest.skip("g++ missing")
    probe = subprocess.run(["g++"] + SAN + ["-x", "c++", "-", "-o", str(tmp_path / "probe")], input="int main(){return 0;}", capture_output=True, text=True)
    if probe.returncode != 0:
        pytest.skip("-fsanitize=address,undefined is not available with this g++")


def _clean(r, what):
    bad = [l for l in r.stderr.splitlines() if "ERROR: AddressSanitizer" in l or "runtime error:" in l or "ERROR: LeakSanitizer" in l]
    assert not bad, f"{what}: {bad[:3]}\n" + r.stderr[:6000]
    assert r.returncode == 0, f"{what}: exit code {r.returncode}\n" + r.stderr[-3000:] + r.stdout[-1000:]


def _san_python_env():
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    return dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1",     # (leaks: the interpreter's own)
                ORB_ORACLE_LIB=os.path.join(ROOT, "oracle", "_san", "liborb_oracle_san.so"))


def test_oracle_tests_under_asan_ubsan(tmp_path):
    _need_sanitizers(tmp_path)
    env = _san_python_env()
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert " passed" in r.stdout and " failed" not in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    _clean(r, "tests/test_oracle.py on the sanitized oracle")


def test_oracle_fuzz_slice_under_asan_ubsan(tmp_path):
    _need_sanitizers(tmp_path)
    env = _san_python_env()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san", "oracle_fuzz_slice.py"), "2000", "7"], capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=900)
    _clean(r, "fuzz slice on the sanitized oracle")
    assert r.stdout.startswith("ok ") and "extract=200" in r.stdout and "bow=400" in r.stdout, r.stdout


def test_context_table_under_asan_ubsan(tmp_path):
    _need_sanitizers(tmp_path)
    exe = str(tmp_path / "context_table_check")
    r = subprocess.run(["g++", "-std=c++11", "-O1", "-g", "-Wall"] + SAN + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "gf-orb-slam2_amd", "adapter"),
                        os.path.join(ROOT, "tests", "host", "context_table_check.cc"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    _clean(r, "context_table_check")
    assert r.stdout.startswith("OK")


def test_combiner_under_asan_ubsan(tmp_path):
    _need_sanitizers(tmp_path)
    exe = str(tmp_path / "combine_san")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g"] + SAN + ["-x", "c++", os.path.join(ROOT, "tests", "host", "combine_tsan.cc"), "-o", exe, "-lpthread"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=600)
    _clean(r, "the combiner harness")
    assert "combine_tsan ok" in r.stdout, r.stdout[-2000:]


def test_abi_host_code_under_asan_ubsan(tmp_path):
    """csrc/gfo_api.hip and gfo_combine.hip, the product files, compiled with g++ against the fake HIP runtime"""
    _need_sanitizers(tmp_path)
    exe = str(tmp_path / "gfo_api_san")
    host = os.path.join(ROOT, "tests", "host")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g"] + SAN + ["-I", os.path.join(host, "fakehip"), "-x", "c++", os.path.join(host, "gfo_api_san.cc"),
                        os.path.join(host, "san_api_tu.cc"), os.path.join(host, "san_combine_tu.cc"), "-o", exe, "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    env = {k: v for k, v in os.environ.items() if not k.startswith("GFO_")}
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    _clean(r, "the ABI's host code")
    assert "gfo_api_san ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
