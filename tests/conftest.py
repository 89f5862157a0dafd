"""Shared pytest configuration.

Markers
-------
gpu : needs a real MI355X (driver runs ``-m gpu`` on the GPU box, ``-m "not gpu"`` here).

Fixtures mirror the reference's tests/conftest.py:12-51 (same seeds and shapes).
"""

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
_TEST_SEED = 42


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture
def random_signal():
    return np.random.default_rng(_TEST_SEED).standard_normal(22050).astype(np.float32)


@pytest.fixture
def chirp_signal():
    sr = 22050
    t = np.linspace(0, 1.0, sr, dtype=np.float32)
    return np.sin(2 * np.pi * (100 + 900 * t / 2) * t).astype(np.float32)


@pytest.fixture
def short_signal():
    return np.random.default_rng(_TEST_SEED).standard_normal(1024).astype(np.float32)


@pytest.fixture
def batch_signals():
    return np.random.default_rng(_TEST_SEED).standard_normal((4, 22050)).astype(np.float32)


@pytest.fixture
def sine_signal():
    sr = 22050
    t = np.linspace(0, 1.0, sr, dtype=np.float32)
    return np.sin(2 * np.pi * 440 * t).astype(np.float32)
