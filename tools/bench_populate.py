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


sys.path.insert(0, str(ROOT / "tests"))
from synth import write_feature_files as write_catalogue  # noqa: E402


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
