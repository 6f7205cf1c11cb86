"""Random catalogues -- block structures, stored forms (dense fp32 / fp64 / int64 / bool, CSR with 32-
or 64-bit indices), tie-heavy and one-hot data, weights, K up to beyond the sweeps' limit, thresholds,
short band lists that force the scan -- through the float64 re-rank: the lists are the float64
oracle's, i.e. the reference's expression (scripts/populate_database.py:180-218) evaluated per
block with the oracle's cosine_similarity (tests/fuzz_exact.py holds the generator)."""

from __future__ import annotations

import importlib.util
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def test_random_catalogues_exact_path_equals_float64_oracle(pkg, device):
    spec = importlib.util.spec_from_file_location("fuzz_exact", Path(__file__).resolve().parent / "fuzz_exact.py")
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    assert fuzz.run_seeds(0, 64) == 0
