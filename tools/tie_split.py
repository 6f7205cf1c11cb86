"""How the ranks of a top-K list split into clear ranks, exact float64 ties and near-ties.

For every show of a hybrid-like catalogue the float64 scores against all shows are computed on
the GPU by torch (rows normalised in float64), the K + 40 best kept, and counted:
  * ranks whose float64 score lies within 2e-6 / 1e-12 of a neighbouring rank;
  * for a float64 re-rank of K + s candidates picked by fp32 score: rows whose K-th float64 score
    is NOT clear of the (K+s)-th by a margin (1e-6), i.e. rows the re-rank could not certify, and
    how many of those are exact float64 tie groups over the whole stretch.
Usage: python tools/tie_split.py [n] [d] [k] [kind]
"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]


def main():
    import torch

    from test_gpu_configs import _hybrid_chunk, _z_float64

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    kind = sys.argv[4] if len(sys.argv) > 4 else "hybrid"
    ms = 0.1 if kind == "hybrid" else -1.0
    dev = torch.device("cuda:0")
    if kind == "hybrid":
        offsets, weights = [0, 32, d - 32, d], [0.4, 0.5, 0.1]
        x = _hybrid_chunk(torch, dev, n, d, 30003)
    else:
        offsets, weights = [0, d], [1.0]
        gen = torch.Generator(device=dev)
        gen.manual_seed(30004)
        x = torch.randn((n, d), generator=gen, device=dev)
    z64 = _z_float64(torch, x, offsets, weights)
    del x
    ext = 40
    slacks = (2, 4, 8, 12, 16, 24, 32, 40)
    tot = dict(ranks=0, near2e6=0, near1e12=0, exact_eq=0)
    uncert = {s: 0 for s in slacks}
    uncert_exact = {s: 0 for s in slacks}
    uncert_thr = {s: 0 for s in slacks}
    chunk = 2500
    ar = torch.arange(k, device=dev)
    for c0 in range(0, n, chunk):
        r = min(chunk, n - c0)
        h = z64[c0:c0 + r] @ z64.T
        h[torch.arange(r, device=dev), c0 + torch.arange(r, device=dev)] = -float("inf")
        v, _ = torch.topk(h, k + ext + 1, dim=1)
        cnt = (v[:, :k] >= ms).sum(1)
        valid = ar[None, :] < cnt[:, None]
        prev = torch.cat([torch.full((r, 1), float("inf"), dtype=torch.float64, device=dev), v[:, :k - 1]], 1)
        gp, gn = prev - v[:, :k], v[:, :k] - v[:, 1:k + 1]
        tot["ranks"] += int(valid.sum())
        tot["near2e6"] += int((valid & ((gp <= 2e-6) | (gn <= 2e-6))).sum())
        tot["near1e12"] += int((valid & ((gp <= 1e-12) | (gn <= 1e-12))).sum())
        tot["exact_eq"] += int((valid & ((gp == 0) | (gn == 0))).sum())
        full = cnt == k
        for s in slacks:
            # candidates K+s picked in fp32: certified when the K-th float64 score clears the (K+s)-th
            last = v[:, k + s - 1]
            listfull = last >= ms - 2e-6  # the candidate list is full only if K+s scores pass the (banded) threshold
            bad = full & listfull & (v[:, k - 1] - last <= 1e-6)
            uncert[s] += int(bad.sum())
            uncert_exact[s] += int((bad & (v[:, k - 1] - last <= 1e-12)).sum())
            uncert_thr[s] += int((bad & (v[:, k - 1] <= ms + 1e-6)).sum())
        del h, v
    print(f"n={n} d={d} k={k} {kind}: {tot}")
    print(f"  not clear at 2e-6: {tot['near2e6'] / tot['ranks']:.4f}; of them within 1e-12 of a neighbour: "
          f"{tot['near1e12'] / max(tot['near2e6'], 1):.4f}; exactly equal: {tot['exact_eq'] / max(tot['near2e6'], 1):.4f}")
    for s in slacks:
        print(f"  slack {s:2d}: rows not certified {uncert[s]} ({uncert[s] / n:.4%}); whole stretch within 1e-12: "
              f"{uncert_exact[s]}; K-th score at the threshold: {uncert_thr[s]}")


if __name__ == "__main__":
    main()
