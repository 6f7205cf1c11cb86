#!/usr/bin/env python3
"""Random catalogues through the float64 re-rank (run on the GPU box):

    python tests/fuzz_exact.py [first_seed] [count]

For every seed: random n, block structure (1-6 blocks in the groups genre / text / metadata),
stored forms (dense fp32 / fp64 / int64 / bool, CSR fp32 / fp64 with 32- or 64-bit indices), data
kinds (dense, sparse, few distinct rows = long runs of exactly tied scores, one-hot), weights (also
negative ones and sums far from 1), K
(also beyond the sweeps' limit) and threshold; now and then short candidate lists so that the
later stages (band, scan) are taken.  engine.exact_topk must return the lists of the float64
oracle: the reference's expression (scripts/populate_database.py:180-218) evaluated with the
oracle's cosine_similarity per block -- indices equal at every rank that is not a float64 tie,
scores within 1e-12, equal scores by index.  Prints one line per seed (the case is named before it
runs) and exits non-zero at the first difference."""
from __future__ import annotations

import os
os.environ.setdefault("TVBF_TUNING", "1")
import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import torch  # noqa: E402

import tvbf_amd  # noqa: F401,E402
from oracle import similarity_oracle as so  # noqa: E402  (checker only)
from parity import REFINED_GAP, REFINED_TOL, check_topk_against_scores  # noqa: E402
from tvbingefriend_recommendation_service_amd import engine  # noqa: E402


def make_block(rng, kind, n, w, pick=None):
    if pick is not None:  # show i is a copy of pattern pick[i] in every block: whole shows repeat
        base = np.abs(rng.standard_normal((int(pick.max()) + 1, w))) * (rng.random((int(pick.max()) + 1, w)) < 0.6)
        return base[pick]
    if kind == "dense":
        return rng.standard_normal((n, w))
    if kind == "sparse":
        x = np.abs(rng.standard_normal((n, w))) * (rng.random((n, w)) < min(0.5, 4.0 / w + 0.05))
        return x
    if kind == "few":  # a handful of distinct rows: long runs of equal scores
        base = np.abs(rng.standard_normal((int(rng.integers(2, 9)), w))) * (rng.random((1, w)) < 0.6)
        return base[rng.integers(0, len(base), n)]
    x = np.zeros((n, w))  # one-hot rows: cosines are 0 or 1
    x[np.arange(n), rng.integers(0, min(w, 12), n)] = 1.0
    return x


def store_block(rng, x, z_col, dev):
    """One block in a random stored form; returns (source, the float64 values the form holds)."""
    binary = np.all((x == 0) | (x == 1))
    integral = np.all(x == np.round(x))
    forms = ["f64", "f32", "csr64", "csr32"] + (["i64"] if integral else []) + (["bool"] if binary else [])
    form = forms[int(rng.integers(0, len(forms)))]
    if form == "f64":
        return engine.DenseSource(torch.from_numpy(np.ascontiguousarray(x)).to(dev), z_col), x, form
    if form == "f32":
        x32 = x.astype(np.float32)
        return engine.DenseSource(torch.from_numpy(x32).to(dev), z_col), x32.astype(np.float64), form
    if form == "i64":
        return engine.DenseSource(torch.from_numpy(x.astype(np.int64)).to(dev), z_col), x, form
    if form == "bool":
        return engine.DenseSource(torch.from_numpy(x.astype(bool)).to(dev), z_col), x, form
    data_t = np.float64 if form == "csr64" else np.float32
    idx_t = np.int64 if rng.integers(0, 2) else np.int32
    m = sp.csr_matrix(x.astype(data_t))
    src = engine.CsrSource(torch.from_numpy(m.indptr.astype(idx_t)).to(dev), torch.from_numpy(m.indices.astype(idx_t)).to(dev),
                           torch.from_numpy(m.data).to(dev), x.shape[1], z_col)
    return src, np.asarray(m.todense(), dtype=np.float64), form + ("/i64" if idx_t is np.int64 else "/i32")


def run_seeds(first: int, count: int) -> int:
    dev = torch.device("cuda:0")
    for seed in range(first, first + count):
        rng = np.random.default_rng(5000 + seed)
        n = int(rng.choice([1, 2, 7, 60, 129, 300, 777, 1500, 2600]))
        nb = int(rng.integers(1, 7))
        groups = sorted(int(g) for g in rng.integers(0, 3, nb))
        widths = [int(rng.choice([3, 8, 12, 32, 40, 100, 300])) for _ in range(nb)]
        kinds = [("dense", "sparse", "few", "onehot")[int(rng.integers(0, 4))] for _ in range(nb)]
        if rng.integers(0, 3) == 0:
            kinds = ["few"] * nb  # whole shows repeat: exact ties in the hybrid score
        gw = [float(v) for v in rng.choice([0.0, 0.1, 0.25, 0.4, 0.5, 1.0, 1.0, 3.5, -0.75, -2.0], 3)]
        if sum(abs(v) for v in gw) == 0:
            gw[0] = 1.0
        k = int(rng.choice([1, 3, 10, 20, 22, 30, 50, 92, 100, 108, 116, 130]))
        ms = float(rng.choice([-1.0, 0.0, 0.1, 0.25, 0.6]))
        short_lists = int(rng.integers(0, 4)) == 0
        print(f"seed {seed}: n={n} blocks={list(zip(groups, widths, kinds))} w={gw} k={k} min={ms}"
              f"{' short band lists' if short_lists else ''}", end=" ", flush=True)
        offsets = [0]
        sources, blocks, forms = [], [], []
        pick = rng.integers(0, int(rng.integers(2, 40)), n) if kinds == ["few"] * nb else None
        for b in range(nb):
            x = make_block(rng, kinds[b], n, widths[b], pick)
            s, held, form = store_block(rng, x, offsets[-1], dev)
            sources.append(s), blocks.append(held), forms.append(form)
            offsets.append(offsets[-1] + widths[b])
        gcount = [groups.count(g) for g in range(3)]
        # the reference's expression: per group the mean of its blocks' cosines, then (wg g + wt t) + wm m
        cos = [so.cosine_similarity(x) for x in blocks]
        gm = []
        for g in range(3):
            members = [cos[b] for b in range(nb) if groups[b] == g]
            if not members:
                gm.append(np.zeros((n, n)))
                continue
            acc = members[0].copy()
            for m in members[1:]:
                acc = acc + m
            gm.append(acc / len(members) if len(members) > 1 else acc)
        h = (gw[0] * gm[0] + gw[1] * gm[1]) + gw[2] * gm[2]
        per_block = [gw[groups[b]] / gcount[groups[b]] for b in range(nb)]
        if min(per_block) >= 0:
            z = engine.assemble_normalize(sources, n, offsets, [float(np.sqrt(w)) for w in per_block])
            z_q = None
        else:  # a negative weight has no square root: unit blocks on the column side, the weights on the query side
            z = engine.assemble_normalize(sources, n, offsets, [1.0] * nb)
            z_q = engine.assemble_normalize(sources, n, offsets, [float(w) for w in per_block])
        old = engine.BAND_LIST_MAX
        if short_lists:
            engine.BAND_LIST_MAX = int(rng.choice([k + 2, 40, 200]))
        try:
            top = engine.exact_topk(z, sources, offsets, groups, gw, k, ms, n=n, z_q=z_q)
        finally:
            engine.BAND_LIST_MAX = old
        torch.cuda.synchronize()
        idx, score, cnt = (t.cpu().numpy() for t in (top.idx, top.score, top.count))
        stats = {}
        mag = max(1.0, sum(abs(v) for v in gw))  # float64 rounding grows with the scores' magnitude
        check_topk_against_scores(h, 0, k, ms, idx, score, cnt, tol=mag * REFINED_TOL, gap=mag * REFINED_GAP,
                                  what=f"seed {seed}", stats=stats, f64=True, check_tie_order=False)
        for r in range(n):
            c = int(cnt[r])
            if np.any(idx[r, :c] == r) or np.any(idx[r, c:] != -1):
                print(f"\nrow {r}: self or stale entries in the list")
                return 1
            eq = np.diff(score[r, :c]) == 0
            if np.any(np.diff(idx[r, :c])[eq] <= 0):
                print(f"\nrow {r}: equal float64 scores not in index order")
                return 1
        print(f"ok ({', '.join(forms)}; stages {top.stats}; max err {stats.get('max_err', 0):.1e}, "
              f"{stats.get('exact', 0)}/{stats.get('ranks', 0)} ranks by index)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(run_seeds(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 40))
