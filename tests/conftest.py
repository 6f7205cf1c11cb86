"""Test configuration: the `gpu` marker, import paths, shared helpers."""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import pytest

import os

# the library looks at its TVBF_* tuning / test switches only in a process started with this set
os.environ.setdefault("TVBF_TUNING", "1")

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu`)")


@pytest.fixture(scope="session")
def pkg():
    import tvbf_amd

    return tvbf_amd.package


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


def load_golden(name: str):
    return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
