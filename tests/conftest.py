"""pytest configuration: `gpu` marker + shared fixtures.

`-m "not gpu"`: oracle vs goldens, host logic, C-ABI symbol export, gloo sharding.
`-m gpu`: parity of the HIP path (through libmcorr's C ABI) against the oracle.
"""

import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))

    return load


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")


def blob_stack(moving: bool):
    """the reference's test fixtures (tests/test_estimate_motion.py:13-33,
    tests/test_correct_motion.py:15-32): 5x64x64 Gaussian blob, moving +2/+1 px per frame"""
    t, h, w = 5, 64, 64
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing="ij")
    img = torch.zeros(t, h, w)
    for f in range(t):
        cy = (h // 2 + (2 * f if moving else 0)) % h
        cx = (w // 2 + (f if moving else 0)) % w
        img[f] = torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 10**2))
    return img


def drift_stack(t, h, w, seed=1234, noise=1.0, pad=64):
    """white-noise texture cropped at integer drift + noise"""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(h + 2 * pad, w + 2 * pad, generator=g)
    dy = torch.round(torch.linspace(-6, 8, t)).long()
    dx = torch.round(torch.linspace(5, -4, t)).long()
    frames = [
        base[pad - dy[f] : pad - dy[f] + h, pad - dx[f] : pad - dx[f] + w]
        + noise * torch.randn(h, w, generator=g)
        for f in range(t)
    ]
    return torch.stack(frames), dy, dx


def ramp_field(t=5, g=2):
    f = torch.zeros(2, t, g, g)
    for i in range(t):
        f[0, i], f[1, i] = 0.1 * i, 0.05 * i
    return f
