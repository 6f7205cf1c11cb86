#!/usr/bin/env python3
"""scripts/compute_similarities.py's counterpart driven from feature files on disk at a size whose
n x n matrices cannot exist (reference: scripts/compute_similarities.py:84-133, :180-262), wall clock:

    python tools/bench_compute_similarities.py [shows]

Writes the five feature files in the reference's on-disk form into a temporary directory, runs
main() twice (the second run is reported) and prints one JSON line: the time from the files to the
four statistics blocks and the top-20 lists, and the statistics themselves."""
import json
import logging
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    import torch

    import tvbf_amd  # noqa: F401
    from synth import write_feature_files
    from tvbingefriend_recommendation_service_amd.scripts import compute_similarities as cs

    logging.disable(logging.CRITICAL)
    with tempfile.TemporaryDirectory() as tmp:
        feats = write_feature_files(Path(tmp), n)
        torch.zeros(1, device="cuda").sum().item()
        out = None
        for _ in range(2):
            t0 = time.perf_counter()
            out = cs.main(["--input-dir", tmp])
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        t0 = time.perf_counter()
        cs.load_features(Path(tmp))
        read = time.perf_counter() - t0
    print(json.dumps({"shows": n, "feature_columns": int(sum(v.shape[1] for v in feats.values())),
                      "matrix_bytes_the_reference_would_hold": cs.matrix_bytes(feats),
                      "files_to_statistics_and_top20_s": round(wall, 3), "of_which_reading_the_feature_files_s": round(read, 3),
                      "pairs_kept": int(out["top_neighbours"].count.sum()), "statistics": out["statistics"]}))


if __name__ == "__main__":
    main()
