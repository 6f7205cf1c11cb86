#!/usr/bin/env python3
"""Times the memory-bound kernels around the fused similarity kernel on one GPU and prints
achieved GB/s against their algorithmic bytes (one JSON line per kernel).

    python tools/bench_aux.py [--n 100000] [--d 1024] [--k 20] [--reps 20]
"""

from __future__ import annotations

import argparse
import json
import os
os.environ.setdefault("TVBF_TUNING", "1")  # the library looks at its TVBF_* switches only then
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0


def timed(fn, reps):
    import torch

    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    ev[1].synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000)
    ap.add_argument("--d", type=int, default=1024)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    import numpy as np
    import scipy.sparse as sp
    import torch

    import tvbf_amd  # noqa: F401
    from tvbingefriend_recommendation_service_amd import engine

    dev = torch.device("cuda:0")
    n, d, k = args.n, args.d, args.k
    dt = d - 64
    rng = np.random.default_rng(1)
    genre = (rng.random((n, 32)) < 0.08).astype(np.int64)
    text = sp.random(n, dt, density=0.03, format="csr", random_state=2, dtype=np.float64)
    meta = np.zeros((n, 32))
    meta[np.arange(n), rng.integers(0, 21, n)] = 1
    meta[np.arange(n), 21 + rng.integers(0, 5, n)] = 1
    meta[np.arange(n), 26 + rng.integers(0, 6, n)] = 1
    offsets, scales = [0, 32, 32 + dt, d], [0.4 ** 0.5, 0.5 ** 0.5, 0.1 ** 0.5]
    src = [engine.DenseSource(torch.from_numpy(genre).to(dev), 0),
           engine.CsrSource(torch.from_numpy(text.indptr).to(dev), torch.from_numpy(text.indices).to(dev),
                            torch.from_numpy(text.data).to(dev), dt, 32),
           engine.DenseSource(torch.from_numpy(meta).to(dev), 32 + dt)]
    ld = engine.padded_ld(d)
    z = torch.empty((n, ld), dtype=torch.float32, device=dev)
    out = []

    ms = timed(lambda: engine.assemble_normalize(src, n, offsets, scales, out=z), args.reps)
    nnz = text.nnz
    bytes_ = n * ld * 4 + nnz * 12 + (n + 1) * 4 + n * 32 * 16
    out.append(("assemble_normalize_kernel", ms, bytes_, "Z written once + CSR (8 B value, 4 B index) and dense sources read once"))

    x = z.clone()
    ms = timed(lambda: engine.normalize_blocks(x, offsets, scales, out=z), args.reps)
    out.append(("normalize_blocks_kernel", ms, 2 * n * ld * 4, "X read + Z written"))

    for nq in (1, 8):
        q = z[:nq].clone()
        ws = torch.empty(engine.native.load().tvbf_query_workspace_bytes(nq, n, k) // 8 + 2, dtype=torch.int64, device=dev)
        ms = timed(lambda: engine.query_topk(q, z, k, 0.1, self_index=list(range(nq)), workspace=ws), args.reps)
        out.append((f"query_topk nq={nq} (scores + select + merge)", ms, n * ld * 4 + 2 * nq * n * 4,
                    "Z read once + scores written and read"))

    top = engine.cosine_topk(z[:4096], k, 0.1, z_q=None) if False else None
    idx = torch.randint(0, n, (n, k), dtype=torch.int32, device=dev)
    ms = timed(lambda: engine.component_scores(z, offsets, scales, idx), max(2, args.reps // 4))
    out.append(("component_scores_kernel", ms, n * k * 2 * ld * 4, "both rows of every kept pair (cache hits included)"))

    ms = timed(lambda: engine.gram_stats(z), 2)
    print(json.dumps({"kernel": "gram_stats (sample sweeps + one band sweep + selection in the buffer; TVBF_STATS_ROUTE=radix: three sweeps)", "ms": round(ms, 3),
                      "TFLOP/s_executed": round((1.0 if os.environ.get("TVBF_STATS_ROUTE") != "radix" and n >= 24576 else 3.0)
                                                * n * n * ld / (ms * 1e-3) / 1e12, 1),
                      "note": "n^2 D flops per sweep over the pairs above the diagonal (2 D per pair); band route: one such "
                              "sweep (the sample's three small sweeps not counted), radix route: three",
                      "workload": f"n={n} d={d}"}), flush=True)

    for name, ms, b, what in out:
        gbs = b / (ms * 1e-3) / 1e9
        print(json.dumps({"kernel": name, "ms": round(ms, 4), "algorithmic_bytes": int(b), "GB/s": round(gbs, 1),
                          "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 3), "bytes_counted": what,
                          "workload": f"n={n} d={d} k={k}"}), flush=True)


if __name__ == "__main__":
    main()
