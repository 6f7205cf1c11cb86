#!/usr/bin/env python3
"""The drop-in `compute_and_store_similarities` on a synthetic catalogue of production size, from the
feature files on disk to the sink (the step of the reference's scripts/populate_database.py:85-247
that this build replaces), timed with the wall clock:

    python tools/bench_populate.py [shows]

Writes the five feature files and shows_metadata.csv in the reference's on-disk form
(scripts/compute_features.py:115-129: genre multi-hot int64, TF-IDF CSR float64 with 500 columns
and unit rows, platform one-hot of 21 float64, type dummies bool, language one-hot of 6 float64)
into a temporary directory, then runs the drop-in twice -- with the array-native sink and with the
reference's dict-of-records sink.  Prints one JSON line.  (The reference's own loop takes 12.6 ms
per show at 10,000 shows on 8 host cores; bench.py times its restatement on the GPU box.)"""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import scipy.sparse as sp


def write_catalogue(path: Path, n: int, seed: int = 7) -> dict:
    rng = np.random.default_rng(seed)
    genre = (rng.random((n, 24)) < 0.1).astype(np.int64)
    nnz_per_row = rng.integers(8, 48, size=n)
    nnz_per_row[rng.random(n) < 0.02] = 0  # shows without a summary: all-zero TF-IDF rows
    indptr = np.concatenate([[0], np.cumsum(nnz_per_row)])
    cols = np.concatenate([np.sort(rng.choice(500, size=c, replace=False)) for c in nnz_per_row]) if n else np.empty(0, int)
    vals = np.abs(rng.standard_normal(indptr[-1])) + 0.05
    text = sp.csr_matrix((vals, cols, indptr), shape=(n, 500), dtype=np.float64)
    norms = np.sqrt(np.asarray(text.multiply(text).sum(axis=1)).ravel())
    norms[norms == 0] = 1.0
    text = sp.diags(1.0 / norms) @ text
    text = sp.csr_matrix(text)

    def one_hot(width, dtype, p=None):
        out = np.zeros((n, width), dtype=dtype)
        out[np.arange(n), rng.choice(width, size=n, p=p)] = 1
        return out

    p_platform = np.r_[np.full(20, 0.7 / 20), 0.3]
    feats = {"genre_features": genre, "text_features": text,
             "platform_features": one_hot(21, np.float64, p_platform),
             "type_features": one_hot(8, bool), "language_features": one_hot(6, np.float64, [0.6, 0.1, 0.1, 0.08, 0.07, 0.05])}
    for key, value in feats.items():
        if sp.issparse(value):
            sp.save_npz(path / f"{key}.npz", value)
        else:
            np.save(path / f"{key}.npy", value)
    ids = np.arange(1, n + 1) * 3 + 11
    (path / "shows_metadata.csv").write_text("id,name\n" + "\n".join(f"{i},show {i}" for i in ids) + "\n")
    return feats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("shows", nargs="?", type=int, default=80_000)
    args = ap.parse_args()
    import torch

    import tvbf_amd  # noqa: F401
    from tvbingefriend_recommendation_service_amd.scripts.populate_database import (
        ColumnarSimilaritySink, MemorySimilaritySink, compute_and_store_similarities)

    out = {"shows": args.shows}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        feats = write_catalogue(tmp, args.shows)
        out["feature_columns"] = int(sum(v.shape[1] for v in feats.values()))
        torch.zeros(1, device="cuda").sum().item()  # context up before the clock starts
        for name, sink in (("warm-up", ColumnarSimilaritySink()), ("columnar_sink_s", ColumnarSimilaritySink()),
                           ("record_dict_sink_s", MemorySimilaritySink())):
            t0 = time.perf_counter()
            stats = compute_and_store_similarities(tmp, repo=sink)
            torch.cuda.synchronize()
            if name != "warm-up":
                out[name] = round(time.perf_counter() - t0, 3)
        from tvbingefriend_recommendation_service_amd.scripts.compute_similarities import load_features

        t0 = time.perf_counter()
        load_features(tmp)
        out["of_which_reading_the_feature_files_s"] = round(time.perf_counter() - t0, 3)
        out["records"] = int(stats["total_records"])
        out["shows_with_records"] = int(stats["unique_shows"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
