#!/usr/bin/env python3
"""Random shapes through every route of the top-K sweep (run on the GPU box):

    python tests/fuzz_routes.py [first_seed] [count] [big]

For every seed: a random catalogue (dense gaussian, hybrid-like, few distinct rows, one-hot rows),
random n, d, K and threshold.  The general sweep on the LDS-DMA loop is the reference point; the
register-staged loop, the lists in global memory (where K allows either), the fused symmetric
sweeps and the stored-matrix route must return the same idx / score / count bit for bit, and the
lists are checked against float64 scores computed on the host.  Prints one line per seed and exits
non-zero at the first difference (the case is named before it runs)."""
from __future__ import annotations

import os
os.environ.setdefault("TVBF_TUNING", "1")  # the library looks at its TVBF_* switches only then
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch  # noqa: E402

import tvbf_amd  # noqa: F401,E402
from oracle import similarity_oracle as so  # noqa: E402  (checker only)
from parity import check_topk_against_scores  # noqa: E402
from tvbingefriend_recommendation_service_amd import engine  # noqa: E402


def make_rows(rng, kind, n, d):
    if kind == "dense":
        return rng.standard_normal((n, d), dtype=np.float32)
    if kind == "hybrid":
        x = np.zeros((n, d), dtype=np.float32)
        g = max(4, d // 8)
        x[:, :g] = rng.random((n, g)) < 0.15
        t = np.abs(rng.standard_normal((n, d - 2 * g), dtype=np.float32))
        t *= rng.random((n, d - 2 * g)) < 0.1
        x[:, g:d - g] = t
        x[np.arange(n), d - g + rng.integers(0, g, n)] = 1.0
        return x
    if kind == "few":  # a handful of distinct rows: long runs of equal scores
        base = rng.standard_normal((int(rng.integers(2, 12)), d), dtype=np.float32)
        return base[rng.integers(0, len(base), n)]
    x = np.zeros((n, d), dtype=np.float32)  # one-hot rows: scores are 0 or 1
    x[np.arange(n), rng.integers(0, min(d, 24), n)] = 1.0
    return x


def with_env(pairs, fn):
    old = {k: os.environ.get(k) for k in pairs}
    os.environ.update(pairs)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def same(a, b):
    return all(torch.equal(getattr(a, f), getattr(b, f)) for f in ("idx", "score", "count"))


def run_seeds(first: int, count: int, big: bool = False) -> int:
    """0 when every route agreed for every seed (an oracle mismatch raises AssertionError).
    ``big``: catalogues of up to 20,000 shows and rows of up to 2,048 floats."""
    dev = torch.device("cuda:0")
    for seed in range(first, first + count):
        rng = np.random.default_rng(1000 + seed)
        kind = ("dense", "hybrid", "few", "onehot")[int(rng.integers(0, 4))]
        n = int(rng.choice([5000, 8200, 12000, 20000] if big else [130, 257, 640, 1000, 1921, 2500, 4097, 6000]))
        d = int(rng.choice([64, 576, 1024, 2048] if big else [32, 64, 96, 128, 192, 256, 320, 512]))
        k = int(rng.choice([1, 2, 5, 10, 20, 30, 31, 50, 100, 116]))
        ms = float(rng.choice([-1.0, 0.0, 0.1, 0.3]))
        print(f"seed {seed}: {kind} n={n} d={d} k={k} min={ms}", end=" ", flush=True)
        x = make_rows(rng, kind, n, d)
        z = engine.normalize_blocks(torch.from_numpy(x).to(dev), [0, d], [1.0])  # rows padded to 32 or a multiple of 64 floats
        ld = z.shape[1]
        ref = engine.cosine_topk(z, k, ms)
        torch.cuda.synchronize()
        routes = []
        routes.append(("staged", with_env({"TVBF_TILE_LOOP": "staged"}, lambda: engine.cosine_topk(z, k, ms))))
        if k <= 30:
            routes.append(("global lists", with_env({"TVBF_LIST": "global"}, lambda: engine.cosine_topk(z, k, ms))))
        if ld % 64 == 0:
            if with_env({"TVBF_SYM_ROUTE": "matrix"}, lambda: engine.symmetric_workspace_bytes(n, ld, k)) > 0:
                routes.append(("matrix", with_env({"TVBF_SYM_ROUTE": "matrix"}, lambda: engine.cosine_topk_symmetric(z, k, ms))))
            if n > 1920 and with_env({"TVBF_SYM_ROUTE": "fused"}, lambda: engine.symmetric_workspace_bytes(n, ld, k)) > 0:
                routes.append(("fused symmetric", with_env({"TVBF_SYM_ROUTE": "fused"}, lambda: engine.cosine_topk_symmetric(z, k, ms))))
        if ld % 64 == 0:  # every rank of a symmetric sharded job, driven in lockstep on this GPU
            from test_gpu_symmetric import _lockstep_ranks
            from tvbingefriend_recommendation_service_amd import sharding

            world = int(rng.choice([2, 3, 4, 5, 8]))
            stored = bool(rng.integers(0, 2))
            if sharding.symmetric_route_fits(world, k, ld):
                got, _flag = _lockstep_ranks(dev, z, n, world, k, ms, own_matrix=stored)
                routes.append((f"{world} ranks{' (stored own blocks)' if stored else ''}", got))
        torch.cuda.synchronize()
        for name, got in routes:
            if not same(ref, got):
                print(f"\nDIFFERENT: {name}")
                return 1
        zo = so.l2_normalize_rows(x.astype(np.float64))
        rows = np.sort(rng.choice(n, size=min(n, 256), replace=False))
        h = zo[rows] @ zo.T
        idx, score, cnt = (t.cpu().numpy() for t in (ref.idx, ref.score, ref.count))
        stats = {}
        for o, r in enumerate(rows):
            check_topk_against_scores(h[o:o + 1], int(r), k, ms, idx[r:r + 1], score[r:r + 1], cnt[r:r + 1],
                                      what=f"seed {seed} row {r}", stats=stats)
        print(f"ok ({', '.join(nm for nm, _ in routes)}; max err {stats.get('max_err', 0):.1e})", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(run_seeds(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40,
                       big=len(sys.argv) > 3 and sys.argv[3] == "big"))
