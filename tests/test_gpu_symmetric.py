"""GPU tests of the symmetric whole-catalogue sweep: every pair of shows computed once, lists
equal to the plain sweep's bit for bit (and the plain sweep is checked against the oracle in
test_gpu_topk.py), here once more against the float64 oracle directly."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import similarity_oracle as so
from parity import check_topk_against_scores
from synth import dense_gaussian, hybrid_like_blocks

pytestmark = pytest.mark.gpu


def _to_np(t):
    return t.detach().cpu().numpy()


def _z_hybrid(device, n, seed, d_genre=16, d_text=208):
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    g, t, p, ty, la = hybrid_like_blocks(n, seed=seed, d_genre=d_genre, d_text=d_text)
    x = np.ascontiguousarray(np.hstack([g, t, p, ty, la]), dtype=np.float32)
    offsets = [0, d_genre, d_genre + d_text, x.shape[1]]
    scales = [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)]
    return engine.normalize_blocks(torch.from_numpy(x).to(device), offsets, scales), x, offsets


def _sym(z, k, ms, expect_overflow=False, **kw):
    """Symmetric sweep on its own workspace; checks whether the lists came from the symmetric
    route itself or from the plain re-run after an overflow."""
    import os

    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, ld = z.shape
    before = os.environ.get("TVBF_SYM_ROUTE")
    os.environ["TVBF_SYM_ROUTE"] = "fused"  # small catalogues would otherwise take the stored-matrix route
    try:
        nbytes = engine.symmetric_workspace_bytes(n, ld, k)
        assert nbytes > 0
        ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=z.device)
        top = engine.cosine_topk_symmetric(z, k, ms, workspace=ws, **kw)
        info = engine.symmetric_inspect(ws, n, ld, k)
    finally:
        if before is None:
            del os.environ["TVBF_SYM_ROUTE"]
        else:
            os.environ["TVBF_SYM_ROUTE"] = before
    if expect_overflow is not None:
        assert (info["overflowed"] > 0) == expect_overflow, info
    assert info["appended"] > 0
    return top


def _assert_same(a, b, what):
    assert np.array_equal(_to_np(a.count), _to_np(b.count)), what
    assert np.array_equal(_to_np(a.idx), _to_np(b.idx)), what
    assert np.array_equal(_to_np(a.score), _to_np(b.score)), what


@pytest.mark.parametrize("own", ["matrix", "fused"])
@pytest.mark.parametrize("n,shards", [(1500, 16), (1500, 5), (1500, 3), (1500, 2), (2049, 4), (2049, 7), (4000, 16)])
def test_symmetric_equals_plain_hybrid(pkg, device, monkeypatch, n, shards, own):
    """Odd and even shard counts, ragged last shard and last tile: identical lists -- with the
    shards' own blocks stored as a batch of matrices (the default) and swept by the fused kernel."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    monkeypatch.setenv("TVBF_SYM_SHARDS", str(shards))
    monkeypatch.setenv("TVBF_SYM_OWN", own)
    z, _, _ = _z_hybrid(device, n, seed=200 + n)
    assert z.shape[1] % 64 == 0
    for k, ms in ((20, 0.1), (7, 0.3)):
        plain = engine.cosine_topk(z, k, ms)
        sym = _sym(z, k, ms)
        torch.cuda.synchronize()
        _assert_same(sym, plain, f"n={n} shards={shards} k={k}")


@pytest.mark.parametrize("n,shards,ring", [(4000, 4, 2), (4000, 4, 3), (4000, 4, 4), (2600, 2, 4), (2600, 2, 7),
                                           (5000, 3, 13), (3000, 5, 2)])
def test_symmetric_two_levels_equals_plain(pkg, device, monkeypatch, n, shards, ring):
    """Every shard swept symmetrically itself (a ring of sub-shards: own columns, half sweep inside
    the ring, bounds merged from row lists and column buffers) before the half sweep over the
    shards: identical lists for even and odd rings, ragged sub-shards, rings of one row block."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    monkeypatch.setenv("TVBF_SYM_SHARDS", str(shards))
    monkeypatch.setenv("TVBF_SYM_RING", str(ring))
    monkeypatch.setenv("TVBF_SYM_OWN", "fused")  # the rings belong to the fused own-block sweeps
    z, _, _ = _z_hybrid(device, n, seed=900 + n + ring)
    # (31, 0.05): lists in global memory; hundreds of shows tie at two shared metadata values, so a
    # few column buffers may run full on these small matrices -- the lists must agree either way
    for k, ms, overflow in ((20, 0.1, False), (31, 0.05, None)):
        plain = engine.cosine_topk(z, k, ms)
        sym = _sym(z, k, ms, expect_overflow=overflow)
        torch.cuda.synchronize()
        _assert_same(sym, plain, f"n={n} shards={shards} ring={ring} k={k}")


@pytest.mark.parametrize("knob,value", [("TVBF_ITEMS", "37"), ("TVBF_ITEMS", "1000"), ("TVBF_MIN_STRETCH", "1"),
                                        ("TVBF_MIN_STRETCH", "9"), ("TVBF_COL_DRAIN", "16"), ("TVBF_COL_WINDOW", "256"),
                                        ("TVBF_COL_WINDOW", "128")])
def test_work_partition_never_changes_the_lists(pkg, device, monkeypatch, knob, value):
    """How the tiles are cut into stretches, when the column queue is drained and how often a
    show's bound is refreshed moves or saves work, nothing else: both sweeps give the same lists."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    monkeypatch.setenv("TVBF_SYM_SHARDS", "6")
    z, _, _ = _z_hybrid(device, 2500, seed=77)
    want = engine.cosine_topk(z, 20, 0.1)
    torch.cuda.synchronize()
    monkeypatch.setenv(knob, value)
    plain = engine.cosine_topk(z, 20, 0.1)
    sym = _sym(z, 20, 0.1)
    torch.cuda.synchronize()
    _assert_same(plain, want, f"plain sweep with {knob}={value}")
    _assert_same(sym, want, f"symmetric sweep with {knob}={value}")


@pytest.mark.parametrize("n,d,k", [(1300, 64, 24), (3000, 128, 20), (1025, 192, 1)])
def test_symmetric_dense_against_oracle(pkg, device, monkeypatch, n, d, k):
    """No min_similarity cut, negative scores in the lists: against the float64 oracle and the plain sweep."""
    import torch

    monkeypatch.setenv("TVBF_SYM_SHARDS", "3")  # fewer than 2048 shows: not taken by default

    from tvbingefriend_recommendation_service_amd import engine

    x = dense_gaussian(n, d, seed=77 + n)
    z = engine.normalize_blocks(torch.from_numpy(x).to(device), [0, d], [1.0])
    sym = _sym(z, k, -1.0)
    plain = engine.cosine_topk(z, k, -1.0)
    torch.cuda.synchronize()
    _assert_same(sym, plain, f"dense {n}x{d}")
    zo = so.build_z([x.astype(np.float64)], [1.0])
    check_topk_against_scores(zo @ zo.T, 0, k, -1.0, _to_np(sym.idx), _to_np(sym.score), _to_np(sym.count),
                              what=f"symmetric dense {n}x{d} k={k}")


@pytest.mark.parametrize("n,shards,k,ms,self_too", [
    (2100, 16, 20, 0.1, False),    # 17 row blocks, 16 shards: blocks of one and two row blocks, ragged last one
    (1999, 15, 100, -1.0, False),  # lists in global memory, threshold off: every score of a block is a candidate
    (3333, 3, 7, 0.3, True),       # large blocks, the show itself kept
    (4097, 32, 31, 0.05, False),   # one row in the last row block
])
def test_own_blocks_as_a_batch_of_matrices(pkg, device, monkeypatch, n, shards, k, ms, self_too):
    """The shards' own blocks through the batched triangle kernel and the row selection (the default
    of the whole-catalogue call): lists equal to the plain sweep and to the fused own-block sweeps,
    and the workspace holds the blocks' matrices."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    monkeypatch.setenv("TVBF_SYM_SHARDS", str(shards))
    monkeypatch.setenv("TVBF_SYM_ROUTE", "fused")
    z, _, _ = _z_hybrid(device, n, seed=4000 + n)
    ld = z.shape[1]
    plain = engine.cosine_topk(z, k, ms, exclude_self=not self_too)
    results = {}
    for own in ("matrix", "fused"):
        monkeypatch.setenv("TVBF_SYM_OWN", own)
        results[own] = (_sym(z, k, ms, expect_overflow=None, exclude_self=not self_too),
                        engine.symmetric_workspace_bytes(n, ld, k))
    torch.cuda.synchronize()
    _assert_same(results["matrix"][0], plain, f"own blocks stored, n={n} shards={shards} k={k}")
    _assert_same(results["fused"][0], plain, f"own blocks fused, n={n} shards={shards} k={k}")
    widest = max((v + 1) * ((n + 127) // 128) // shards - v * ((n + 127) // 128) // shards for v in range(shards))
    assert results["matrix"][1] > n * widest * 128 * 4  # the workspace holds the blocks' matrices


def test_symmetric_keeps_self_when_asked(pkg, device, monkeypatch):
    import torch

    monkeypatch.setenv("TVBF_SYM_SHARDS", "4")

    from tvbingefriend_recommendation_service_amd import engine

    z, _, _ = _z_hybrid(device, 1800, seed=9)
    sym = _sym(z, 10, 0.1, exclude_self=False)
    plain = engine.cosine_topk(z, 10, 0.1, exclude_self=False)
    torch.cuda.synchronize()
    _assert_same(sym, plain, "exclude_self=False")
    # nearly every show leads its own list (identical twins with a lower index come first)
    assert (_to_np(sym.idx)[:, 0] == np.arange(1800)).mean() > 0.9


def test_symmetric_overflow_recomputes_plainly(pkg, device, monkeypatch):
    """Column buffers of two entries overflow at once: the gated plain sweep takes over on the
    device and the lists are still exact."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    z, _, _ = _z_hybrid(device, 1700, seed=31)
    plain = engine.cosine_topk(z, 20, 0.1)
    monkeypatch.setenv("TVBF_SYM_SHARDS", "2")
    monkeypatch.setenv("TVBF_SYM_COL_CAP", "2")
    sym = _sym(z, 20, 0.1, expect_overflow=True)
    torch.cuda.synchronize()
    _assert_same(sym, plain, "overflow")


def test_symmetric_with_cliques_of_identical_shows(pkg, device):
    """Groups of identical rows (exact ties at score 1 and at every other score): ties are
    resolved by column index on both sides exactly as in the plain sweep."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    rng = np.random.default_rng(5)
    base = dense_gaussian(60, 128, seed=3)
    x = base[rng.integers(0, 60, size=2500)]  # about 40 copies of each of 60 shows
    z = engine.normalize_blocks(torch.from_numpy(np.ascontiguousarray(x)).to(device), [0, 128], [1.0])
    for k in (5, 20):
        sym = _sym(z, k, -1.0)
        plain = engine.cosine_topk(z, k, -1.0)
        torch.cuda.synchronize()
        _assert_same(sym, plain, f"cliques k={k}")


def test_symmetric_shape_limits(pkg, device, monkeypatch):
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    assert engine.symmetric_workspace_bytes(100_000, 1024, 20) > 0
    assert engine.symmetric_workspace_bytes(100_000, 1024 + 32, 20) == 0  # row length must be a multiple of 64
    assert engine.symmetric_workspace_bytes(100_000, 1024, 50) > 0        # large K: lists in global memory
    assert engine.symmetric_workspace_bytes(100_000, 1024, 100) > 0
    # small catalogues: the scores are stored as an n x n matrix (rows padded to 32 floats)
    assert engine.symmetric_workspace_bytes(1900, 64, 5) == 1900 * 1920 * 4
    assert engine.symmetric_workspace_bytes(24576, 64, 5) == 24576 * 24576 * 4
    assert 0 < engine.symmetric_workspace_bytes(24577, 64, 5) < 24577 * 24577 * 4  # fused sweeps: lists, own blocks, column buffers
    # long lists (kept in global memory by the fused sweeps) take the matrix route further up
    assert engine.symmetric_workspace_bytes(46336, 64, 50) == 46336 * 46336 * 4
    assert 0 < engine.symmetric_workspace_bytes(46337, 64, 50) < 46337 * 46337 * 4
    assert engine.symmetric_workspace_bytes(1900, 96, 5) == 0
    monkeypatch.setenv("TVBF_SYM_ROUTE", "fused")
    assert engine.symmetric_workspace_bytes(1900, 64, 5) == 0             # fewer than 16 blocks of 128 shows
    assert engine.symmetric_workspace_bytes(2048, 64, 5) > 0
    z = torch.zeros((100, 64), device=device)
    with pytest.raises(ValueError):
        engine.cosine_topk_symmetric(z, 5, 0.0)


@pytest.mark.parametrize("n,d,k,ms,kind", [(1, 64, 5, -1.0, "dense"), (2, 64, 1, -1.0, "dense"), (100, 64, 20, -1.0, "dense"),
                                           (129, 128, 100, -1.0, "dense"), (1500, 192, 20, 0.02, "dense"),
                                           (4000, 256, 20, 0.1, "hybrid"), (2600, 320, 50, 0.05, "hybrid"),
                                           (5000, 64, 20, -1.0, "clique")])
def test_stored_matrix_route_equals_plain(pkg, device, n, d, k, ms, kind):
    """Small catalogues: tiles on and above the diagonal stored once with their transposes, one
    scan per row.  Same fp32 scores, same total order: lists equal the plain sweep's bit for bit --
    one-tile and ragged catalogues, n < K, thresholds, exact ties, K = 100, self kept or left out."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    if kind == "hybrid":
        z, _, _ = _z_hybrid(device, n, seed=n + d, d_genre=32, d_text=d - 64)
    elif kind == "clique":
        z = _with_clique(device, n, 900, d, seed=n)
    else:
        z = engine.normalize_blocks(torch.from_numpy(dense_gaussian(n, d, seed=n + d)).to(device), [0, d], [1.0])
    assert z.shape[1] == d and engine.symmetric_workspace_bytes(n, d, k) == n * ((n + 31) // 32 * 32) * 4
    for exclude_self in (True, False):
        got = engine.cosine_topk_symmetric(z, k, ms, exclude_self=exclude_self)
        plain = engine.cosine_topk(z, k, ms, exclude_self=exclude_self)
        torch.cuda.synchronize()
        _assert_same(got, plain, f"{kind} {n}x{d} k={k} exclude_self={exclude_self}")
    assert engine.symmetric_tiles(n, d, k) == ((n + 127) // 128) * ((n + 127) // 128 + 1) // 2
    # the stored scores themselves: the matrix tvbf_gram_f32 returns, symmetric bit for bit
    if n >= 100:
        nbytes = engine.symmetric_workspace_bytes(n, d, k)
        ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=device)
        engine.cosine_topk_symmetric(z, k, ms, workspace=ws)
        m = ws.view(torch.float32)[: n * ((n + 31) // 32 * 32)].view(n, -1)[:, :n]
        assert torch.equal(m, engine.gram(z))
        assert torch.equal(m, m.T)


def test_symmetric_benchmark_shape_sample(pkg, device):
    """N = 100k x D = 1024, K = 20: equal to the plain sweep on every row (bitwise)."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, d, k, ms = 100_000, 1024, 20, 0.1
    gen = torch.Generator(device=device)
    gen.manual_seed(6)
    x = torch.zeros((n, d), device=device)
    x[:, :32] = (torch.rand((n, 32), generator=gen, device=device) < 0.08).float()
    text = torch.randn((n, d - 64), generator=gen, device=device).abs_()
    text *= torch.rand((n, d - 64), generator=gen, device=device) < 0.03
    x[:, 32 : d - 32] = text
    del text
    x[torch.arange(n, device=device), d - 32 + torch.randint(0, 32, (n,), generator=gen, device=device)] = 1.0
    z = engine.normalize_blocks(x, [0, 32, d - 32, d], [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)])
    del x
    sym = _sym(z, k, ms)
    plain = engine.cosine_topk(z, k, ms)
    torch.cuda.synchronize()
    assert torch.equal(sym.count, plain.count)
    assert torch.equal(sym.idx, plain.idx)
    assert torch.equal(sym.score, plain.score)


def _lockstep_ranks(device, z, n, world, k, ms, col_cap=1024, sides=None, own_matrix=False):
    """All ranks of a symmetric sharded job on one GPU, phase by phase, with the collectives
    replaced by indexing; returns the concatenated lists and the reduced overflow flag."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine, sharding

    shards = [sharding.shard_bounds(n, world, r) for r in range(world)]
    per = shards[0].per_rank
    z_all = torch.zeros((per * world, z.shape[1]), device=device)
    z_all[:n] = z[:n]
    def matrix_for(sh):  # a rank's buffer for the scores of its own block
        nbytes = engine.own_matrix_bytes(sh.rows, z.shape[1]) if own_matrix else 0
        return torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=device) if nbytes else None

    jobs = [sharding.SymmetricShardJob(z_all[r * per : (r + 1) * per], sh, k, ms, col_cap=col_cap,
                                       colside=None if sides is None else sides[r], own_matrix=matrix_for(sh))
            for r, sh in enumerate(shards)]
    for j in jobs:
        j.phase_own()
    bounds_all = torch.cat([j.own_bounds() for j in jobs])  # the all-gather
    for j in jobs:
        j.phase_half(z_all, bounds_all)
    sends = [j.lists_for_peers(bounds_all) for j in jobs]
    flag = torch.stack([j.flag for j in jobs]).max(dim=0).values.contiguous()  # the max-reduction
    tops = []
    for r, j in enumerate(jobs):
        recv = torch.stack([sends[p][r] for p in range(world)]).contiguous()  # the all-to-all
        tops.append(j.finish(z_all, recv, flag))
    torch.cuda.synchronize()
    cat = lambda f: torch.cat([getattr(t, f)[: sh.rows] for t, sh in zip(tops, shards)])  # noqa: E731
    return engine.TopK(cat("idx"), cat("score"), cat("count")), int(flag[0].item())


@pytest.mark.parametrize("n,world", [(3000, 2), (3000, 3), (3000, 4), (5000, 8), (2049, 4), (700, 8), (1300, 5), (10, 8),
                                     (9000, 2), (9100, 4), (13000, 3)])  # the last three: own blocks large enough for the inner symmetric sweep
def test_symmetric_sharded_job_equals_plain(pkg, device, n, world):
    """Every block between two shards is computed by one of them only (the partner block of an
    even world half by each); the lists equal the single-GPU plain sweep bit for bit."""
    from tvbingefriend_recommendation_service_amd import engine

    z, _, _ = _z_hybrid(device, n, seed=400 + n + world)
    for k, ms in ((20, 0.1), (3, 0.25)):
        plain = engine.cosine_topk(z, k, ms)
        got, flag = _lockstep_ranks(device, z, n, world, k, ms)
        assert flag == 0
        _assert_same(got, plain, f"n={n} world={world} k={k}")


@pytest.mark.parametrize("n,world", [(3000, 2), (5000, 8), (2049, 4), (700, 8), (10, 8), (9100, 4), (13000, 3)])
def test_symmetric_sharded_job_with_stored_own_blocks(pkg, device, n, world):
    """Every rank's own block through the stored-matrix route (one list slot and the bound keys),
    the blocks between shards as before: the lists equal the single-GPU plain sweep bit for bit."""
    from tvbingefriend_recommendation_service_amd import engine

    z, _, _ = _z_hybrid(device, n, seed=700 + n + world)
    for k, ms in ((20, 0.1), (50, 0.05)):
        plain = engine.cosine_topk(z, k, ms)
        got, _ = _lockstep_ranks(device, z, n, world, k, ms, own_matrix=True)
        _assert_same(got, plain, f"stored own blocks n={n} world={world} k={k}")


def test_plain_sharded_route_with_stored_own_block(pkg, device):
    """The plain sharded route with the own block through the stored matrix and the other columns
    as two ranges of the gathered matrix: lists of the single sweep."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine, sharding

    n, k, ms, world = 7000, 20, 0.1, 3
    z, _, _ = _z_hybrid(device, n, seed=31)
    full = engine.cosine_topk(z, k, ms)
    for rank in range(world):
        sh = sharding.shard_bounds(n, world, rank)
        own, after, before = sharding.column_pieces(sh)
        zq = z[sh.row0:sh.row0 + sh.rows].contiguous()
        ses = engine.TopkSession(zq, sh.rows, sh.row0, k, ms)
        nbytes = engine.own_matrix_bytes(sh.rows, z.shape[1])
        assert nbytes == sh.rows * ((sh.rows + 31) // 32 * 32) * 4
        assert ses.accumulate_own_matrix(torch.empty(nbytes // 8 + 1, dtype=torch.int64, device=device)) == 1
        ses.accumulate((z[after[0]:], after[1], after[0]), (z, before[1], before[0]))
        top = ses.finish()
        torch.cuda.synchronize()
        sl = slice(sh.row0, sh.row0 + sh.rows)
        for f in ("idx", "score", "count"):
            assert torch.equal(getattr(top, f), getattr(full, f)[sl]), (rank, f)


@pytest.mark.parametrize("n,world,shards,ring", [(9000, 2, 4, 2), (9100, 4, 2, 3), (13000, 3, 3, 4)])
def test_symmetric_sharded_job_two_levels(pkg, device, monkeypatch, n, world, shards, ring):
    """Ranks whose own blocks are swept on two levels (as blocks of 16 or more row blocks per
    shard are by default): same lists as the single-GPU plain sweep."""
    from tvbingefriend_recommendation_service_amd import engine

    z, _, _ = _z_hybrid(device, n, seed=600 + n + world)
    plain = engine.cosine_topk(z, 20, 0.1)
    monkeypatch.setenv("TVBF_SYM_SHARDS", str(shards))
    monkeypatch.setenv("TVBF_SYM_RING", str(ring))
    got, flag = _lockstep_ranks(device, z, n, world, 20, 0.1)
    assert flag == 0
    _assert_same(got, plain, f"n={n} world={world} shards={shards} ring={ring}")


def test_symmetric_sharded_job_overflow(pkg, device):
    """Column buffers of one entry overflow on every rank: all ranks re-run plainly, same lists."""
    from tvbingefriend_recommendation_service_amd import engine

    z, _, _ = _z_hybrid(device, 2500, seed=12)
    plain = engine.cosine_topk(z, 20, 0.1)
    got, flag = _lockstep_ranks(device, z, 2500, 4, 20, 0.1, col_cap=1)
    assert flag == 1
    _assert_same(got, plain, "sharded overflow")


def test_facade_topk_takes_symmetric_sweep_and_agrees(pkg, device, monkeypatch):
    """SimilarityComputer.topk on a catalogue large enough for the symmetric sweep: the same
    TopKResult as with the plain sweep forced, bit for bit, component scores included."""
    from tvbingefriend_recommendation_service_amd.ml.similarity_computer import SimilarityComputer

    feats = hybrid_like_blocks(9000, seed=21, d_genre=24, d_text=264, as_features=True)  # 24 + 264 + 32 = 320 columns
    sc = SimilarityComputer(genre_weight=0.4, text_weight=0.5, metadata_weight=0.1)
    monkeypatch.setenv("TVBF_SWEEP", "plain")
    plain = sc.topk(feats, k=20, min_similarity=0.1)
    monkeypatch.setenv("TVBF_SWEEP", "auto")
    auto = sc.topk(feats, k=20, min_similarity=0.1)
    monkeypatch.setenv("TVBF_SWEEP", "symmetric")
    forced = sc.topk(feats, k=20, min_similarity=0.1)
    for other in (auto, forced):
        assert np.array_equal(plain.idx, other.idx) and np.array_equal(plain.score, other.score)
        assert np.array_equal(plain.count, other.count) and np.array_equal(plain.comp, other.comp)
    assert int(plain.count.sum()) > 0


def test_symmetric_k24_large(pkg, device):
    """The largest K of the symmetric route (lists capped at 34 entries beside the column queue)."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    z, _, _ = _z_hybrid(device, 6000, seed=77)
    # thousands of shows tie at the K-th score of some shows here (metadata-only matches): whether
    # one of their buffers overflows into the plain re-run or not, the lists must be the same
    sym = _sym(z, 24, 0.05, expect_overflow=None)
    plain = engine.cosine_topk(z, 24, 0.05)
    torch.cuda.synchronize()
    _assert_same(sym, plain, "k=24")


def _with_clique(device, n, members, d, seed):
    """Dense gaussian shows of which `members` are copies of one show (scattered over the catalogue)."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    rng = np.random.default_rng(seed)
    x = dense_gaussian(n, d, seed=seed)
    who = rng.permutation(n)[:members]
    x[who] = x[who[0]]
    return engine.normalize_blocks(torch.from_numpy(np.ascontiguousarray(x)).to(device), [0, d], [1.0])


def test_symmetric_large_group_of_identical_shows(pkg, device, monkeypatch):
    """1500 identical shows: each of them ties with all the others at score 1, so their column
    buffers take hundreds of candidates and their bounds are refreshed from the buffered windows
    (a bound can then be one of the buffered keys itself).  Lists must still equal the plain sweep's."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, k = 8000, 20
    monkeypatch.setenv("TVBF_SYM_ROUTE", "fused")  # the column side is what is tested here
    z = _with_clique(device, n, 1500, 128, seed=41)
    nbytes = engine.symmetric_workspace_bytes(n, z.shape[1], k)
    ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=device)
    sym = engine.cosine_topk_symmetric(z, k, -1.0, workspace=ws)
    info = engine.symmetric_inspect(ws, n, z.shape[1], k)
    assert info["max_per_show"] >= 256, info  # the refresh path was taken
    plain = engine.cosine_topk(z, k, -1.0)
    torch.cuda.synchronize()
    _assert_same(sym, plain, "group of identical shows")


def test_symmetric_sharded_job_with_identical_shows(pkg, device):
    """The same through the sharded job (bounds refreshed in each rank's gathered copy)."""
    from tvbingefriend_recommendation_service_amd import engine

    n, k = 9000, 20
    z = _with_clique(device, n, 2000, 128, seed=43)
    plain = engine.cosine_topk(z, k, -1.0)
    for world in (2, 3):
        got, _ = _lockstep_ranks(device, z, n, world, k, -1.0)
        _assert_same(got, plain, f"identical shows, world={world}")


@pytest.mark.parametrize("k", [27, 30, 31, 50, 100])
def test_symmetric_large_k(pkg, device, k):
    """K above 30: the row lists live in global memory, the column side works the same way."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    z, _, _ = _z_hybrid(device, 5000, seed=300 + k)
    sym = _sym(z, k, 0.05, expect_overflow=None)
    plain = engine.cosine_topk(z, k, 0.05)
    torch.cuda.synchronize()
    _assert_same(sym, plain, f"k={k}")
    x = dense_gaussian(3000, 128, seed=k)
    zd = engine.normalize_blocks(torch.from_numpy(x).to(device), [0, 128], [1.0])
    _assert_same(_sym(zd, k, -1.0), engine.cosine_topk(zd, k, -1.0), f"dense k={k}")


def test_symmetric_sharded_job_large_k(pkg, device):
    from tvbingefriend_recommendation_service_amd import engine, sharding

    z, _, _ = _z_hybrid(device, 9100, seed=5)
    k = 50
    assert sharding.symmetric_route_fits(4, k, z.shape[1]) and sharding.symmetric_route_fits(8, k, z.shape[1])
    assert not sharding.symmetric_route_fits(8, 100, z.shape[1])
    plain = engine.cosine_topk(z, k, 0.1)
    for world in (2, 4, 8):
        got, _ = _lockstep_ranks(device, z, 9100, world, k, 0.1)
        _assert_same(got, plain, f"k=50 world={world}")


def test_topk_calls_can_be_captured_in_a_graph(pkg, device, monkeypatch):
    """The library allocates nothing and clears memory with kernels, so a caller can capture a
    whole top-K call in a HIP graph: every replay (with unrelated eager work and synchronisations
    in between) must reproduce the eager lists, for the symmetric and for the plain call."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, d, k = 4096, 128, 20
    z = engine.normalize_blocks(torch.from_numpy(dense_gaussian(n, d, seed=8)).to(device), [0, d], [1.0])
    ref = engine.cosine_topk(z, k, -1.0)
    torch.cuda.synchronize()
    nb = engine.symmetric_workspace_bytes(n, d, k)  # the stored-matrix route at this size
    monkeypatch.setenv("TVBF_SYM_ROUTE", "fused")
    nb = max(nb, engine.symmetric_workspace_bytes(n, d, k))
    monkeypatch.delenv("TVBF_SYM_ROUTE")
    ws = torch.empty((nb + 15) // 16 * 2, dtype=torch.int64, device=device)  # also large enough for the plain call
    out = engine.TopK(torch.empty((n, k), dtype=torch.int32, device=device), torch.empty((n, k), device=device),
                      torch.empty((n,), dtype=torch.int32, device=device))
    calls = {"symmetric, stored matrix": lambda: engine.cosine_topk_symmetric(z, k, -1.0, workspace=ws, out=out),
             "symmetric, fused": lambda: engine.cosine_topk_symmetric(z, k, -1.0, workspace=ws, out=out),
             "plain": lambda: engine.cosine_topk(z, k, -1.0, workspace=ws, out=out)}
    side = torch.cuda.Stream()
    for name, call in calls.items():
        if name == "symmetric, fused":  # the route is chosen when the call is made, i.e. at capture
            monkeypatch.setenv("TVBF_SYM_ROUTE", "fused")
        else:
            monkeypatch.delenv("TVBF_SYM_ROUTE", raising=False)
        with torch.cuda.stream(side):
            call()  # first call outside the capture (sets kernel attributes)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            call()
        for rep in range(3):
            out.idx.fill_(-7)
            out.score.zero_()
            out.count.zero_()
            _ = torch.randn((128, 128), device=device) @ torch.randn((128, 128), device=device)
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            _assert_same(out, ref, f"{name} replay {rep}")


def test_column_side_state_is_reusable(pkg, device):
    """A rank keeps one column-side state for all its jobs: after the first job only the entries
    that job wrote are cleared.  A job on tie-heavy data (full buffers, refreshed bounds) followed
    by jobs on other catalogues of the same size must leave no trace."""
    from tvbingefriend_recommendation_service_amd import engine, sharding

    n, k, world = 9000, 20, 3
    per = sharding.shard_bounds(n, world, 0).per_rank
    sides = [engine.ColumnSide(per * world, device) for _ in range(world)]
    cases = [(_with_clique(device, n, 2500, 128, seed=51), -1.0), (_z_hybrid(device, n, seed=52)[0], 0.1),
             (_with_clique(device, n, 700, 128, seed=53), -1.0)]
    for z, ms in cases:
        plain = engine.cosine_topk(z, k, ms)
        got, _ = _lockstep_ranks(device, z, n, world, k, ms, sides=sides)
        _assert_same(got, plain, "reused column-side state")
