"""Times the REAL reference beside the oracle port on the same inputs (build container only: the
reference tree does not travel to the GPU box).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo python /root/repo/tests/golden/time_reference.py

Matrix mode: SimilarityComputer.compute_all_similarities (ml/similarity_computer.py:132-169) followed
by the selection of scripts/populate_database.py:195-218 on every row (np.argsort(row)[::-1] + walk).
Row-loop mode: the literal per-show loop of scripts/populate_database.py:170-218, restated (that
script needs Python >= 3.11 and is not importable here), on a sample of rows.
The port is oracle/similarity_oracle.py, which bench.py times on the GPU box as `cpu_baseline`.
"""

from __future__ import annotations

import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

from tvbingefriend_recommendation_service.ml.similarity_computer import SimilarityComputer  # the reference

from oracle import similarity_oracle as so
from synth import hybrid_like_blocks


def ref_select(h, k, ms):
    kept_total = 0
    for i in range(h.shape[0]):
        kept = 0
        row = h[i]
        for j in np.argsort(row)[::-1]:
            if j == i:
                continue
            if row[j] < ms:
                break
            if kept >= k:
                break
            kept += 1
        kept_total += kept
    return kept_total


def main():
    import logging

    logging.disable(logging.CRITICAL)
    out = []
    for n in (3000, 10000):
        feats = hybrid_like_blocks(n, seed=30001, d_genre=28, d_text=500, as_features=True)
        t0 = time.perf_counter()
        sims = SimilarityComputer().compute_all_similarities(feats)
        t1 = time.perf_counter()
        kept = ref_select(sims["hybrid_similarity"], 20, 0.1)
        t2 = time.perf_counter()
        del sims
        # the port, as bench.py's cpu_baseline runs it: one Z, chunked float64 GEMM, partition + sort per row
        blocks, weights = so.hybrid_blocks(feats, 0.4, 0.5, 0.1, variant="computer")
        t3 = time.perf_counter()
        z = so.build_z(blocks, weights)
        idx, score, count = so.topk_from_z(z, 20, 0.1, rows=range(n), chunk=512)
        t4 = time.perf_counter()
        # the literal per-show loop on a sample
        rows = list(range(0, n, max(1, n // 16)))[:16]
        t5 = time.perf_counter()
        so.populate_rows(feats, list(range(n)), 0.4, 0.5, 0.1, 20, 0.1, rows=rows)
        t6 = time.perf_counter()
        rec = {"n": n, "d": "28 + 500 (CSR 3 %) + 32",
               "reference_matrices_s": t1 - t0, "reference_selection_s": t2 - t1,
               "reference_pairs_per_s": n * n / (t2 - t0),
               "port_s": t4 - t3, "port_pairs_per_s": n * n / (t4 - t3),
               "pairs_kept_reference": int(kept), "pairs_kept_port": int(count.sum()),
               "row_loop_ms_per_show": (t6 - t5) / len(rows) * 1e3,
               "row_loop_pairs_per_s": len(rows) * n / (t6 - t5)}
        out.append(rec)
        print(json.dumps(rec), flush=True)
    return out


if __name__ == "__main__":
    main()
