import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_u8(name, w=752, h=480):
    return np.fromfile(os.path.join(GOLDEN, name), np.uint8).reshape(h, w)


@pytest.fixture(scope="session")
def euroc_l():
    return load_u8("EuRoC_l_752x480.u8")


@pytest.fixture(scope="session")
def euroc_r():
    return load_u8("EuRoC_r_752x480.u8")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orb_oracle
    orb_oracle.build()
    return orb_oracle


def synth_frame(w, h, idx=0):
    """Synthetic S2 frame of SURVEY.md 8d: three octaves of value noise + 400 random dark/bright
    rectangles, seeded numpy.random.default_rng(20260403 + idx)."""
    rng = np.random.default_rng(20260403 + idx)
    img = np.zeros((h, w), np.float32)
    for o, amp in ((64, 60.0), (32, 30.0), (16, 15.0)):
        gh, gw = h // o + 2, w // o + 2
        g = rng.random((gh, gw), dtype=np.float32)
        ys = np.arange(h, dtype=np.float32) / o
        xs = np.arange(w, dtype=np.float32) / o
        y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
        img += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
    img += 70.0
    for _ in range(400):
        rw, rh = rng.integers(6, 41, 2)
        x = rng.integers(0, w - rw); y = rng.integers(0, h - rh)
        img[y:y + rh, x:x + rw] += rng.choice([-1.0, 1.0]) * rng.uniform(25, 90)
    return np.clip(img, 0, 255).astype(np.uint8)
