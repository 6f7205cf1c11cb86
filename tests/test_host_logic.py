"""CPU tests of the host side: C-ABI surface, feature packing, row sharding (gloo, 2 ranks)."""

from __future__ import annotations

import ctypes
import os
import re
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import similarity_oracle as so
from synth import hybrid_like_blocks

ROOT = Path(__file__).resolve().parent.parent


# ---------------------------------------------------------------- C ABI ----

def test_library_exports_every_declared_symbol(pkg):
    from tvbingefriend_recommendation_service_amd import native

    header = (ROOT / "include" / "tvbf.h").read_text()
    declared = sorted(set(re.findall(r"\b(tvbf_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations found in include/tvbf.h"
    assert sorted(native.SYMBOLS) == declared
    lib = native.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in tvbf.h but not exported"
    assert lib.tvbf_version() >= 100


def test_argument_validation_needs_no_gpu(pkg):
    """Bad arguments are rejected on the host before any device work."""
    from tvbingefriend_recommendation_service_amd import native

    lib = native.load()
    p = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    rc = lib.tvbf_cosine_topk_f32(p, 4, 0, p, 4, 32, 32, 0, 0.1, 1, p, p, p, p, 1 << 20, None)
    assert rc == -1 and b"k must be" in lib.tvbf_last_error()
    rc = lib.tvbf_cosine_topk_f32(p, 4, 0, p, 4, 36, 36, 5, 0.1, 1, p, p, p, p, 1 << 20, None)
    assert rc == -2 and b"multiple of 32" in lib.tvbf_last_error()
    rc = lib.tvbf_cosine_topk_f32(None, 4, 0, p, 4, 32, 32, 5, 0.1, 1, p, p, p, p, 1 << 20, None)
    assert rc == -1
    rc = lib.tvbf_cosine_topk_f32(p, 4, 0, p, 4, 32, 32, 5, float("nan"), 1, p, p, p, p, 1 << 20, None)
    assert rc == -1
    off = (ctypes.c_int32 * 3)(0, 8, 4)
    sc = (ctypes.c_float * 2)(1.0, 1.0)
    rc = lib.tvbf_normalize_blocks_f32(p, 4, 4, 4, off, sc, 2, p, 4, None)
    assert rc == -2
    rc = lib.tvbf_upper_tri_stats_f64(p, 1, 1, p, p, 1 << 20, None)
    assert rc == -1 and b"empty" in lib.tvbf_last_error()
    assert lib.tvbf_topk_workspace_bytes(0, 10, 32, 5) == 0
    assert lib.tvbf_topk_workspace_bytes(10, 10, 32, 117) == 0
    assert lib.tvbf_topk_workspace_bytes(1000, 1000, 32, 20) >= 1000 * 20 * 8


def test_product_fails_loudly_without_gpu(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from tvbingefriend_recommendation_service_amd import engine, native
    from tvbingefriend_recommendation_service_amd.ml.similarity_computer import SimilarityComputer

    with pytest.raises(native.NativeLibraryError):
        SimilarityComputer().compute_genre_similarity(np.eye(3))
    with pytest.raises(native.NativeLibraryError):
        engine.normalize_blocks(torch.zeros(4, 4), [0, 4], [1.0])


# ------------------------------------------------------ feature packing ----

def test_pack_features_layouts(pkg):
    from tvbingefriend_recommendation_service_amd import features as feat

    f = hybrid_like_blocks(50, seed=1, d_genre=7, d_text=30, as_features=True)
    x, lay = feat.pack_features(f, 2.0, 3.0, 1.0, "computer")
    assert x.dtype == np.float32 and x.flags.c_contiguous
    assert lay.offsets == [0, 8, 40, 72] and lay.widths == [7, 30, 32] and lay.groups == [0, 1, 2]
    assert lay.weights == pytest.approx([2 / 6, 3 / 6, 1 / 6])
    assert np.all(x[:, 7] == 0) and np.all(x[:, 38:40] == 0)
    assert np.array_equal(x[:, :7], np.asarray(f["genre_features"], dtype=np.float32))
    assert np.array_equal(x[:, 8:38], f["text_features"].toarray().astype(np.float32))
    x2, lay2 = feat.pack_features(f, 0.4, 0.5, 0.1, "populate")
    assert lay2.offsets == [0, 8, 40, 64, 72, 80] and lay2.groups == [0, 1, 2, 2, 2]
    assert lay2.weights == pytest.approx([0.4, 0.5, 0.1 / 3, 0.1 / 3, 0.1 / 3])
    # the packed, normalised, scaled matrix reproduces the oracle's hybrid matrix
    for xx, ll, variant in ((x, lay, "computer"), (x2, lay2, "populate")):
        w = (2.0, 3.0, 1.0) if variant == "computer" else (0.4, 0.5, 0.1)
        blocks = [xx[:, ll.offsets[b] : ll.offsets[b + 1]].astype(np.float64) for b in range(ll.nb)]
        z = so.build_z(blocks, ll.weights)
        zo = so.build_z(*so.hybrid_blocks(f, *w, variant=variant))
        assert np.abs(z @ z.T - zo @ zo.T).max() <= 1e-7  # fp32 storage of the TF-IDF values
    with pytest.raises(ValueError):
        feat.pack_features(f, -0.1, 0.5, 0.6, "populate")[1].scales()
    with pytest.raises(KeyError):
        feat.pack_features({"genre_features": np.eye(2)}, 1, 1, 1)
    assert feat.BlockLayout([0, 4, 8], [4, 4], [0.0, 1.0], [0, 1]).component_scales() == [1.0, 1.0]


def test_fold_components(pkg):
    from tvbingefriend_recommendation_service_amd import features as feat

    lay = feat.BlockLayout([0, 4, 8, 12, 16, 20], [4] * 5, [0.4, 0.5, 0.1, 0.1, 0.1], [0, 1, 2, 2, 2])
    comp = np.arange(2 * 3 * 5, dtype=np.float32).reshape(2, 3, 5)
    out = feat.fold_components(comp, lay)
    assert out.shape == (2, 3, 3)
    assert np.allclose(out[..., 0], comp[..., 0]) and np.allclose(out[..., 2], comp[..., 2:].mean(axis=-1))


# -------------------------------------------------------- row sharding ----

def test_shard_bounds(pkg):
    from tvbingefriend_recommendation_service_amd.sharding import shard_bounds

    s = [shard_bounds(10, 4, r) for r in range(4)]
    assert [(x.row0, x.rows) for x in s] == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert all(x.per_rank == 3 and x.padded_total == 12 for x in s)
    s = [shard_bounds(5, 8, r) for r in range(8)]
    assert sum(x.rows for x in s) == 5 and s[7].rows == 0 and s[7].row0 == 5
    assert shard_bounds(100_000, 8, 7).row0 == 87_500
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _shard_worker(rank, world, port, n, k, min_sim, ret):
    """One gloo rank: CPU stand-ins for the two kernels, the real sharding code around them."""
    import torch
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import tvbf_amd  # noqa: F401
    from tvbingefriend_recommendation_service_amd import engine, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, t, p, ty, la = hybrid_like_blocks(n, seed=11, d_genre=12, d_text=40)
        x = np.hstack([g, t, p, ty, la]).astype(np.float32)
        offsets, weights = [0, 12, 52, 84], [0.4, 0.5, 0.1]
        scales = [float(np.sqrt(w)) for w in weights]
        sh = sharding.shard_bounds(n, world, rank)
        xs = np.zeros((sh.per_rank, x.shape[1]), np.float32)
        xs[: sh.rows] = x[sh.row0 : sh.row0 + sh.rows]

        def normalize_fn(xt, off, sc):
            xx = xt.numpy().astype(np.float64)
            z = np.hstack([s * so.l2_normalize_rows(xx[:, off[b] : off[b + 1]]) for b, s in enumerate(sc)])
            return torch.from_numpy(np.ascontiguousarray(z))

        class OracleSession:
            """CPU stand-in for engine.TopkSession: scores each piece with the oracle, keeps
            (score, global column) candidates, merges them in finish()."""

            def __init__(self, z_q, nq, q_row0, kk, ms):
                self.zq, self.nq, self.q_row0, self.k, self.ms = z_q.numpy()[:nq], nq, q_row0, kk, ms
                self.scores, self.cols = [], []

            def accumulate(self, *ranges):
                for t, n_rows, col0 in ranges:
                    if n_rows == 0:
                        continue
                    zc = t.numpy()[:n_rows]
                    self.scores.append(self.zq @ zc.T)
                    self.cols.append(np.arange(col0, col0 + n_rows))

            def finish(self):
                h = np.hstack(self.scores)
                cols = np.concatenate(self.cols)
                assert len(set(cols.tolist())) == len(cols) == n, "every column exactly once"
                order = np.argsort(cols)
                i, s, c = so.topk_fast(h[:, order], self.q_row0, self.k, self.ms)
                return engine.TopK(torch.from_numpy(i), torch.from_numpy(s), torch.from_numpy(c))

        top, z_all = sharding.sharded_topk(torch.from_numpy(xs), offsets, scales, sh, k, min_sim,
                                           normalize_fn=normalize_fn, session_fn=OracleSession)
        assert z_all.shape[0] == sh.padded_total and top.idx.shape[0] == sh.rows
        own, after, before = sharding.column_pieces(sh)
        assert own[1] + after[1] + before[1] == n
        idx = sharding.gather_lists(top.idx, sh)
        score = sharding.gather_lists(top.score, sh)
        count = sharding.gather_lists(top.count, sh)
        if rank == 0:
            ret["idx"], ret["score"], ret["count"] = idx.numpy(), score.numpy(), count.numpy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [101, 64])
def test_sharded_topk_two_ranks_gloo(pkg, n):
    """world_size 2 over gloo: shard -> normalise -> all-gather -> own rows -> concatenated
    lists equal the single-process result."""
    import torch.multiprocessing as mp

    k, min_sim = 7, 0.1
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, n, k, min_sim, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    g, t, p_, ty, la = hybrid_like_blocks(n, seed=11, d_genre=12, d_text=40)
    x = np.hstack([g, t, p_, ty, la]).astype(np.float32).astype(np.float64)
    z = np.hstack([np.sqrt(w) * so.l2_normalize_rows(x[:, a:b])
                   for w, a, b in ((0.4, 0, 12), (0.5, 12, 52), (0.1, 52, 84))])
    # the pieces and the whole are summed by BLAS in different orders: compare tie-aware
    from parity import check_topk_against_scores

    check_topk_against_scores(z @ z.T, 0, k, min_sim, ret["idx"], ret["score"], ret["count"], tol=1e-9,
                              what=f"gloo shards n={n}", check_tie_order=False)


def test_block_sources_match_packed_layout(pkg):
    """The stored-form sources (CSR text, int64 genre, bool type) describe exactly the matrix
    that pack_features builds on the host, for both variants."""
    from synth import hybrid_like_blocks
    from tvbingefriend_recommendation_service_amd import features as feat

    f = hybrid_like_blocks(64, 3, d_text=61, as_features=True)
    for variant in ("computer", "populate"):
        x, layout = feat.pack_features(f, 0.4, 0.5, 0.1, variant)
        sources, layout2 = feat.block_sources(f, 0.4, 0.5, 0.1, variant)
        assert (layout2.offsets, layout2.widths, layout2.weights, layout2.groups) == \
               (layout.offsets, layout.widths, layout.weights, layout.groups)
        y = np.zeros_like(x)
        kinds = []
        for s in sources:
            if s[0] == "dense":
                y[:, s[2]:s[2] + s[1].shape[1]] = s[1]
                kinds.append(s[1].dtype)
            else:
                _, indptr, indices, data, width, zc = s
                for i in range(64):
                    y[i, zc + indices[indptr[i]:indptr[i + 1]]] = data[indptr[i]:indptr[i + 1]]
                kinds.append("csr")
        assert np.array_equal(x, y)
        assert kinds == [np.int64, "csr", np.float64, np.bool_, np.float64]  # nothing converted on the host
    with pytest.raises(KeyError):
        feat.block_sources({"genre_features": f["genre_features"]}, 0.4, 0.5, 0.1)


def test_columnar_table_equals_record_dicts(pkg, tmp_path):
    """neighbour_table / iter_table_rows / write_table_csv carry exactly the pairs and values of
    the record dicts (scripts/populate_database.py:210-221 of the reference), in the same order."""
    import csv

    from tvbingefriend_recommendation_service_amd.ml.similarity_computer import TopKResult
    from tvbingefriend_recommendation_service_amd.scripts import populate_database as pd_

    rng = np.random.default_rng(4)
    n, k = 40, 6
    count = rng.integers(0, k + 1, size=n).astype(np.int32)
    count[3] = 0
    idx = np.full((n, k), -1, dtype=np.int32)
    score = np.zeros((n, k), dtype=np.float32)
    comp = np.zeros((n, k, 3), dtype=np.float64)
    for r in range(n):
        c = count[r]
        idx[r, :c] = rng.choice(n, size=c, replace=False)
        score[r, :c] = np.sort(rng.random(c).astype(np.float32))[::-1]
        comp[r, :c] = rng.random((c, 3))
    ids = [1000 + 7 * i for i in range(n)]
    res = TopKResult(idx, score, count, comp)
    recs = pd_.neighbour_records(res, ids, 0, n)
    table = pd_.neighbour_table(res, ids)
    flat = [(sid, r["similar_show_id"], r["similarity_score"], r["genre_score"], r["text_score"], r["metadata_score"])
            for sid, items in recs.items() for r in items]
    rows = [row for batch in pd_.iter_table_rows(table, batch_size=7) for row in batch]
    assert rows == flat and all(type(v) in (int, float) for v in rows[0])
    assert all(len(b) <= 7 for b in pd_.iter_table_rows(table, batch_size=7))
    path = tmp_path / "pairs.csv"
    assert pd_.write_table_csv(table, path) == len(flat)
    with open(path, newline="") as f:
        got = list(csv.reader(f))
    assert tuple(got[0]) == pd_.TABLE_COLUMNS and len(got) == len(flat) + 1
    for line, ref in zip(got[1:], flat):
        assert int(line[0]) == ref[0] and int(line[1]) == ref[1]
        assert np.float32(float(line[2])) == np.float32(ref[2])
        assert all(abs(float(a) - b) <= 1e-9 for a, b in zip(line[3:], ref[3:]))
    sink = pd_.ColumnarSimilaritySink()
    assert sink.store_similarity_table(table) == len(flat)
    st = sink.get_similarity_stats()
    assert st["total_records"] == len(flat) and st["unique_shows"] == len(recs)


# ------------------------------------------ symmetric sharded sweep (gloo) ----

def _f32_ordered(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return np.where(u & 0x80000000, u ^ 0xFFFFFFFF, u | 0x80000000)


def _ordered_f32(m):
    m = np.asarray(m, dtype=np.uint64)
    u = np.where(m & 0x80000000, m ^ 0x80000000, m ^ 0xFFFFFFFF).astype(np.uint32)
    return u.view(np.float32)


def _sym_worker(rank, world, port, n, k, min_sim, ret, resident=False):
    """One gloo rank of sharded_topk_symmetric: the real partition logic and collectives around a
    CPU stand-in for the device job (float64 products, keys in the device format)."""
    import torch
    import torch.distributed as dist

    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import tvbf_amd  # noqa: F401
    from tvbingefriend_recommendation_service_amd import engine, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, t, p, ty, la = hybrid_like_blocks(n, seed=11, d_genre=12, d_text=40)
        x = np.hstack([g, t, p, ty, la]).astype(np.float32)
        offsets, weights = [0, 12, 52, 84], [0.4, 0.5, 0.1]
        scales = [float(np.sqrt(w)) for w in weights]
        sh = sharding.shard_bounds(n, world, rank)
        xs = np.zeros((sh.per_rank, x.shape[1]), np.float32)
        xs[: sh.rows] = x[sh.row0 : sh.row0 + sh.rows]

        def normalize_fn(xt, off, sc):
            xx = xt.numpy().astype(np.float64)
            z = np.hstack([s * so.l2_normalize_rows(xx[:, off[b] : off[b + 1]]) for b, s in enumerate(sc)])
            return torch.from_numpy(np.ascontiguousarray(z))

        class OracleJob:
            """CPU stand-in with the phases of sharding.SymmetricShardJob."""

            def __init__(self, z_shard, shard, kk, ms):
                self.sh, self.k, self.ms = shard, kk, ms
                self.zq = z_shard.numpy()[: shard.rows]
                self.row_scores = [[] for _ in range(shard.rows)]   # per own row: (score f32, global col)
                self.col_scores = {}                                # global row -> [(score f32, global col)]
                self.entries = 0                                    # score-matrix entries computed here

            def _sweep(self, r_lo, zc, col0, both):
                h = (self.zq[r_lo:] @ zc.T).astype(np.float32)
                self.entries += h.size
                for i in range(h.shape[0]):
                    gi = self.sh.row0 + r_lo + i
                    for j in range(h.shape[1]):
                        self.row_scores[r_lo + i].append((h[i, j], col0 + j))
                        if both:
                            self.col_scores.setdefault(col0 + j, []).append((h[i, j], gi))

            def phase_own(self):
                self._sweep(0, self.zq, self.sh.row0, both=False)

            def own_bounds(self):
                return torch.zeros((self.sh.per_rank,), dtype=torch.int64)

            def phase_half(self, z_all, bounds_all):
                assert bounds_all.shape == (self.sh.padded_total,)
                za = z_all.numpy()
                for first, rows in sharding.half_pieces(self.sh):
                    if rows:
                        self._sweep(0, za[first : first + rows], first, both=True)
                pt = sharding.partner(self.sh.world, self.sh.rank)
                if pt is not None and self.sh.rank >= self.sh.world // 2:
                    h0 = sharding.partner_split(self.sh, self.sh.rank)
                    c0 = min(pt * self.sh.per_rank, self.sh.n)
                    cols = min((pt + 1) * self.sh.per_rank, self.sh.n) - c0
                    if self.sh.rows - h0 > 0 and cols > 0:
                        self._sweep(h0, za[c0 : c0 + cols], c0, both=True)

            @staticmethod
            def _keys(pairs):
                s = np.array([p_[0] for p_ in pairs], dtype=np.float32)
                c = np.array([p_[1] for p_ in pairs], dtype=np.uint64)
                return (_f32_ordered(s) << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - c)

            def lists_for_peers(self, bounds_all):
                send = np.zeros((self.sh.world, self.sh.per_rank, self.k), dtype=np.uint64)
                for grow, pairs in self.col_scores.items():
                    keys = np.sort(self._keys(pairs))[::-1][: self.k]
                    send[grow // self.sh.per_rank, grow % self.sh.per_rank, : len(keys)] = keys
                return torch.from_numpy(send.view(np.int64))

            @property
            def flag(self):
                return torch.zeros((4 + self.sh.padded_total,), dtype=torch.int32)

            def finish(self, z_all, recv, flag):
                assert int(flag.max().item()) == 0
                rk = recv.numpy().view(np.uint64)
                idx = -np.ones((self.sh.rows, self.k), np.int32)
                score = np.zeros((self.sh.rows, self.k), np.float32)
                count = np.zeros((self.sh.rows,), np.int32)
                for i in range(self.sh.rows):
                    keys = self._keys(self.row_scores[i])
                    theirs = rk[:, i, :].ravel()
                    keys = np.concatenate([keys, theirs[(theirs & np.uint64(0xFFFFFFFF)) != 0]])
                    cols = (np.uint64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF))).astype(np.int64)
                    assert len(set(cols.tolist())) == len(cols), "a pair reached this row twice"
                    keys = keys[cols != self.sh.row0 + i]
                    keys = np.sort(keys)[::-1]
                    sc = _ordered_f32(keys >> np.uint64(32))
                    keep = sc >= self.ms
                    keys, sc = keys[keep][: self.k], sc[keep][: self.k]
                    c = len(keys)
                    idx[i, :c] = (np.uint64(0xFFFFFFFF) - (keys & np.uint64(0xFFFFFFFF))).astype(np.int32)
                    score[i, :c] = sc
                    count[i] = c
                return engine.TopK(torch.from_numpy(idx), torch.from_numpy(score), torch.from_numpy(count))

        holder = {}

        def job_fn(*a):
            holder["job"] = OracleJob(*a)
            return holder["job"]

        if resident:  # every rank holds all features: it builds all normalised rows itself, no all-gather of them
            xa = np.zeros((sh.padded_total, x.shape[1]), np.float32)
            xa[:n] = x
            z_res = normalize_fn(torch.from_numpy(xa), offsets, scales)
            top, z_all = sharding.sharded_topk_symmetric_z(z_res[rank * sh.per_rank:(rank + 1) * sh.per_rank], sh, k,
                                                           min_sim, job_fn=job_fn, z_all=z_res)
            assert z_all is z_res
        else:
            top, z_all = sharding.sharded_topk_symmetric(torch.from_numpy(xs), offsets, scales, sh, k, min_sim,
                                                         normalize_fn=normalize_fn, job_fn=job_fn)
        entries = torch.tensor([holder["job"].entries], dtype=torch.int64)
        dist.all_reduce(entries)
        idx = sharding.gather_lists(top.idx, sh)
        score = sharding.gather_lists(top.score, sh)
        count = sharding.gather_lists(top.count, sh)
        if rank == 0:
            ret["idx"], ret["score"], ret["count"] = idx.numpy(), score.numpy(), count.numpy()
            ret["entries"] = int(entries.item())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,world,resident", [(101, 2, False), (64, 2, False), (130, 3, False), (150, 4, False),
                                              (101, 2, True), (150, 4, True)])
def test_sharded_topk_symmetric_gloo(pkg, n, world, resident):
    """Real ranks over gloo: the partition (half peers, split partner block), the all-gathers, the
    all-to-all of lists and the flag reduction hand every pair of shows to exactly one rank, and
    the merged lists equal the single-process result -- with the normalised shards all-gathered,
    and with all normalised rows built on every rank (``resident``)."""
    import torch.multiprocessing as mp

    from tvbingefriend_recommendation_service_amd import sharding

    k, min_sim = 7, 0.1
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sym_worker, args=(r, world, port, n, k, min_sim, ret, resident)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    rows = [sharding.shard_bounds(n, world, r).rows for r in range(world)]
    # own blocks in full, every block between two shards once
    assert ret["entries"] == sum(a * a for a in rows) + sum(rows[a] * rows[b] for a in range(world) for b in range(a))
    g, t, p_, ty, la = hybrid_like_blocks(n, seed=11, d_genre=12, d_text=40)
    x = np.hstack([g, t, p_, ty, la]).astype(np.float32).astype(np.float64)
    z = np.hstack([np.sqrt(w) * so.l2_normalize_rows(x[:, a:b])
                   for w, a, b in ((0.4, 0, 12), (0.5, 12, 52), (0.1, 52, 84))])
    from parity import check_topk_against_scores

    check_topk_against_scores(z @ z.T, 0, k, min_sim, ret["idx"], ret["score"], ret["count"], tol=1e-6,
                              what=f"gloo symmetric shards n={n} world={world}", check_tie_order=False)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8, 9, 16])
@pytest.mark.parametrize("n", [10, 1000, 1001, 100_000, 99_999])
def test_symmetric_partition_covers_every_pair_once(pkg, world, n):
    """Pure arithmetic of the symmetric sharded route: own blocks, the half pieces and the second
    sweep of the upper partners together hold every unordered pair of different shards' shows
    exactly once (counted with interval sizes, no GPU and no process group)."""
    from tvbingefriend_recommendation_service_amd import sharding

    shards = [sharding.shard_bounds(n, world, r) for r in range(world)]
    owner = np.repeat(np.arange(world), [s.rows for s in shards])  # shard of every show
    assert len(owner) == n
    covered = np.zeros((world, world), dtype=np.int64)  # [row shard, column shard] -> show pairs computed there
    for s in shards:
        for first, rows in sharding.half_pieces(s):
            if rows:
                cols = owner[first : first + rows]
                assert (cols != s.rank).all(), "a rank's half pieces never hold its own columns"
                for c in np.unique(cols):
                    covered[s.rank, c] += s.rows * int((cols == c).sum())
        pt = sharding.partner(world, s.rank)
        if pt is not None and s.rank >= world // 2:
            h0 = sharding.partner_split(s, s.rank)
            assert 0 <= h0 <= s.rows and (h0 % 128 == 0 or h0 == s.rows)
            covered[s.rank, pt] += (s.rows - h0) * shards[pt].rows
    for a in range(world):
        for b in range(a + 1, world):
            assert covered[a, b] + covered[b, a] == shards[a].rows * shards[b].rows, (world, n, a, b)
    # balance: no rank computes much more than the mean
    work = covered.sum(axis=1)
    if n >= 1000 and world > 1:
        assert work.max() <= 1.1 * work.mean() + 2 * 128 * max(s.rows for s in shards)


def test_record_dicts_built_in_c_equal_the_python_loop(pkg):
    """csrc/_tvbf_records.so builds the reference's dict[int, list[dict]] form of a result
    (populate_database.py:210-221); it must return exactly what the Python loop returns -- ids as
    a list or as an array, whole result or a slice of rows, shows without neighbours left out --
    and step aside for ids that are not integers."""
    from tvbingefriend_recommendation_service_amd import native
    from tvbingefriend_recommendation_service_amd.ml.similarity_computer import TopKResult
    from tvbingefriend_recommendation_service_amd.scripts import populate_database as pd_

    if native.build_records_ext() is None:
        pytest.skip("no C compiler or no Python headers here")
    assert pd_._load_records_ext(), "the extension was built but does not load"
    rng = np.random.default_rng(3)
    n, k = 300, 7
    count = rng.integers(0, k + 1, n).astype(np.int32)
    idx = np.full((n, k), -1, dtype=np.int32)
    score = np.zeros((n, k), dtype=np.float32)
    comp = np.zeros((n, k, 3), dtype=np.float64)
    for r in range(n):
        c = count[r]
        idx[r, :c] = rng.choice(n, size=c, replace=False)
        score[r, :c] = np.sort(rng.random(c).astype(np.float32))[::-1]
        comp[r, :c] = rng.random((c, 3))
    res = TopKResult(idx, score, count, comp)
    ids = [10_000_000_000 + 3 * i for i in range(n)]  # beyond 32 bits
    for show_ids in (ids, np.asarray(ids)):
        for start, stop in ((0, n), (17, 203), (5, 5)):
            want = pd_.neighbour_records(res, show_ids, start, stop, use_ext=False)
            got = pd_.neighbour_records(res, show_ids, start, stop)
            assert got == want and list(got) == list(want)
            for key, items in got.items():
                assert type(key) is int and all(list(d) == list(w) for d, w in zip(items, want[key]))
                assert all(type(v) is (int if name == "similar_show_id" else float) for d in items for name, v in d.items())
    names = [f"show-{i}" for i in range(n)]  # not integers: the Python loop
    assert pd_.neighbour_records(res, names, 0, n) == pd_.neighbour_records(res, names, 0, n, use_ext=False)
    bad = TopKResult(idx.copy(), score, count, comp)
    bad.idx[np.nonzero(count)[0][0], 0] = n + 5
    with pytest.raises(ValueError):
        pd_.neighbour_records(bad, ids, 0, n)


def test_catalogue_route_is_bounded_by_available_memory():
    """engine.choose_catalogue_route: the symmetric call's workspace (score matrices, column-side
    buffers) is taken only while it fits into the available memory; the plain sweep gives the same
    lists with a small workspace."""
    import tvbf_amd  # noqa: F401
    from tvbingefriend_recommendation_service_amd import engine

    gib = 1 << 30
    assert engine.choose_catalogue_route(2 * gib, 50 << 20, 100 * gib) == "symmetric"
    assert engine.choose_catalogue_route(9 * gib, 50 << 20, 8 * gib) == "plain"          # does not fit
    assert engine.choose_catalogue_route(int(7.5 * gib), 50 << 20, 8 * gib) == "plain"   # inside the headroom
    assert engine.choose_catalogue_route(0, 50 << 20, 100 * gib) == "plain"              # shape not taken
    assert engine.choose_catalogue_route(2 * gib, 50 << 20, 100 * gib, "plain") == "plain"
    assert engine.choose_catalogue_route(9 * gib, 50 << 20, 8 * gib, "symmetric") == "symmetric"  # forced
    with pytest.raises(MemoryError):
        engine.choose_catalogue_route(9 * gib, 2 * gib, 1 * gib)
    assert engine.candidate_count(20) == 28 and engine.candidate_count(22) == 30
    assert engine.candidate_count(23) == 39 and engine.candidate_count(90) == 106 and engine.candidate_count(100) == 116
    assert engine.candidate_count(110) == 116 and engine.candidate_count(116) == 116


@pytest.mark.parametrize("world,per,k", [(2, 5, 3), (3, 4, 2), (8, 7, 20)])
def test_packed_exchange_carries_lists_and_overflow_words(pkg, world, per, k):
    """sharding.pack_lists_and_flags / unpack_lists_and_flags: what every rank unpacks after the
    all-to-all (simulated by indexing) are the lists its peers found for its shows and the
    element-wise maximum of their overflow words for those shows."""
    import torch

    from tvbingefriend_recommendation_service_amd import sharding

    gen = torch.Generator().manual_seed(world * 100 + per)
    shards = [sharding.Shard(r, world, world * per - 1, per, r * per, per if r < world - 1 else per - 1) for r in range(world)]
    sends = [torch.randint(-2**62, 2**62, (world, per, k), generator=gen, dtype=torch.int64) for _ in range(world)]
    flags = [torch.randint(0, 2, (4 + world * per,), generator=gen, dtype=torch.int32) * (torch.rand(4 + world * per, generator=gen) < 0.3)
             for _ in range(world)]
    packed = [sharding.pack_lists_and_flags(sends[r], flags[r].to(torch.int32), shards[r]) for r in range(world)]
    for r in range(world):
        recv = torch.stack([packed[p][r] for p in range(world)])  # the all-to-all
        lists, flag = sharding.unpack_lists_and_flags(recv, shards[r], k)
        assert lists.shape == (world, per, k) and lists.stride(1) == k and lists.stride(2) == 1
        for p in range(world):
            assert torch.equal(lists[p], sends[p][r])
        want = torch.stack([flags[p][4 + r * per : 4 + (r + 1) * per] for p in range(world)]).max(dim=0).values.to(torch.int32)
        assert torch.equal(flag[4 + r * per : 4 + (r + 1) * per], want)
        assert int(flag[0]) == int(want.max())
        rest = torch.cat([flag[1 : 4 + r * per], flag[4 + (r + 1) * per :]])
        assert int(rest.abs().max()) == 0
