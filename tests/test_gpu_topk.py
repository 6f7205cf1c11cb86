"""GPU parity tests of the HIP kernels (through the C ABI) against the float64 oracle."""

from __future__ import annotations

import hashlib

import numpy as np
import pytest

from conftest import load_golden
from oracle import similarity_oracle as so
from parity import SCORE_TOL, check_topk_against_golden, check_topk_against_scores
from synth import dense_gaussian, hybrid_like_blocks

pytestmark = pytest.mark.gpu


def _to_np(t):
    return t.detach().cpu().numpy()


def _run_topk(pkg, device, blocks, weights, k, min_similarity, q_row0=0, nq=None, exclude_self=True):
    """blocks: list of float arrays (n, d_b); returns (TopK on host, z on device, offsets, scales)."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    x = np.ascontiguousarray(np.hstack([np.asarray(b, dtype=np.float32) for b in blocks]))
    offsets = np.concatenate([[0], np.cumsum([b.shape[1] for b in blocks])]).astype(int).tolist()
    scales = [float(np.sqrt(w)) for w in weights]
    xd = torch.from_numpy(x).to(device)
    z = engine.normalize_blocks(xd, offsets, scales)
    n = x.shape[0]
    top = engine.cosine_topk(z, k, min_similarity, q_row0=q_row0, nq=nq, exclude_self=exclude_self)
    torch.cuda.synchronize()
    return top, z, offsets, scales


def _oracle_h(blocks, weights):
    z = so.build_z([np.asarray(b, dtype=np.float32).astype(np.float64) for b in blocks], weights)
    return z @ z.T, z


@pytest.mark.parametrize(
    "n,d,k",
    [(1000, 96, 20), (130, 4, 1), (257, 36, 50), (129, 100, 100), (64, 32, 20), (3, 8, 2), (1, 8, 5),
     (2048, 512, 20)],
)
def test_dense_matches_oracle(pkg, device, n, d, k):
    x = dense_gaussian(n, d, seed=1000 + n + d)
    top, *_ = _run_topk(pkg, device, [x], [1.0], k, -1.0)
    h, _ = _oracle_h([x], [1.0])
    stats = {}
    check_topk_against_scores(h, 0, k, -1.0, _to_np(top.idx), _to_np(top.score), _to_np(top.count),
                              what=f"dense {n}x{d} k={k}", stats=stats)
    # ranks more than 2e-6 from both neighbours were compared by index: nearly all of them
    assert stats["exact"] >= 0.98 * stats["ranks"] or n < 4, stats
    assert stats["max_err"] <= 1e-6, stats


def test_normalize_matches_oracle(pkg, device):
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    rng = np.random.default_rng(5)
    g = (rng.random((333, 7)) < 0.3).astype(np.float32)
    t = np.abs(rng.standard_normal((333, 61))).astype(np.float32)
    m = rng.standard_normal((333, 5)).astype(np.float32)
    g[10] = 0
    t[11] = 0
    m[12] = 0
    t[13] *= 1e-9  # below sklearn's guard of 10*eps: divided by 1, stays tiny
    x = np.hstack([g, t, m])
    z = engine.normalize_blocks(torch.from_numpy(x).to(device), [0, 7, 68, 73],
                                [np.sqrt(0.4), np.sqrt(0.5), np.sqrt(0.1)])
    torch.cuda.synchronize()
    zz = _to_np(z)
    assert zz.shape == (333, 128)  # 73 columns padded to a multiple of 64
    assert np.all(zz[:, 73:] == 0)
    ref = np.hstack([np.sqrt(w) * so.l2_normalize_rows(b) for b, w in ((g, 0.4), (t, 0.5), (m, 0.1))])
    assert np.abs(zz[:, :73] - ref).max() <= 2e-7
    assert np.all(zz[10, :7] == 0) and np.all(zz[11, 7:68] == 0) and np.all(zz[12, 68:73] == 0)


def test_hybrid_ties_min_similarity_and_components(pkg, device):
    from tvbingefriend_recommendation_service_amd import engine

    n = 1500
    g, t, p, ty, la = hybrid_like_blocks(n, seed=77, d_genre=24, d_text=200)
    blocks, weights = [g, t, np.hstack([p, ty, la])], [0.4, 0.5, 0.1]
    top, z, offsets, scales = _run_topk(pkg, device, blocks, weights, 20, 0.1)
    h, zo = _oracle_h(blocks, weights)
    idx, score, count = _to_np(top.idx), _to_np(top.score), _to_np(top.count)
    check_topk_against_scores(h, 0, 20, 0.1, idx, score, count, what="hybrid ties")
    # rows without genres and text can only reach 0.1 (metadata): at or below the threshold
    import torch

    comp = engine.component_scores(z, offsets, scales, top.idx)
    torch.cuda.synchronize()
    comp = _to_np(comp)
    for b, blk in enumerate(blocks):
        bn = so.l2_normalize_rows(np.asarray(blk, dtype=np.float64))
        for r in range(0, n, 37):
            c = count[r]
            ref = bn[r] @ bn[idx[r, :c]].T
            assert np.abs(comp[r, :c, b] - ref).max(initial=0) <= SCORE_TOL
            assert np.all(comp[r, c:, b] == 0)


@pytest.mark.parametrize("n,d,k,ms", [(900, 96, 20, -1.0), (1300, 160, 40, 0.02)])
def test_rows_of_odd_k_tile_count_take_the_staged_loop(pkg, device, n, d, k, ms):
    """A caller's own Z whose row length is a multiple of 32 but not of 64 (engine.normalize_blocks
    pads to 64) runs the register-staged tile loop: fused top-K, dense matrix and statistics
    against the float64 oracle."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    x = dense_gaussian(n, d, seed=n + d)
    zo = so.l2_normalize_rows(x.astype(np.float64))
    z = torch.from_numpy(zo.astype(np.float32)).to(device).contiguous()
    assert z.shape[1] % 64 == 32 and engine.symmetric_workspace_bytes(n, d, k) == 0
    top = engine.catalogue_topk(z, k, ms)
    torch.cuda.synchronize()
    z64 = z.cpu().numpy().astype(np.float64)
    h = z64 @ z64.T
    check_topk_against_scores(h, 0, k, ms, _to_np(top.idx), _to_np(top.score), _to_np(top.count), what=f"staged {n}x{d}")
    s32 = engine.gram(z).cpu().numpy()
    assert np.abs(s32 - h).max() <= 1e-6
    got = engine.gram_stats(z).cpu().numpy()
    v = s32.astype(np.float64)[np.triu_indices(n, k=1)]
    assert got[2] == v.min() and got[3] == v.max() and got[4] == np.median(v) and abs(got[0] - v.mean()) <= 1e-12


def test_row_shard_matches_full(pkg, device):
    """Query rows [q_row0, q_row0+nq) of a shard give the same lists as the full run, bit for bit."""
    n, d, k = 700, 64, 20
    x = dense_gaussian(n, d, seed=9)
    full, *_ = _run_topk(pkg, device, [x], [1.0], k, -1.0)
    part, *_ = _run_topk(pkg, device, [x], [1.0], k, -1.0, q_row0=300, nq=250)
    assert np.array_equal(_to_np(full.idx)[300:550], _to_np(part.idx))
    assert np.array_equal(_to_np(full.score)[300:550], _to_np(part.score))
    assert np.array_equal(_to_np(full.count)[300:550], _to_np(part.count))


def test_edge_cases(pkg, device):
    # N < K: every other show is kept, rest padded
    x = dense_gaussian(5, 16, seed=3)
    top, *_ = _run_topk(pkg, device, [x], [1.0], 20, -1.0)
    assert _to_np(top.count).tolist() == [4] * 5
    # everything below the threshold: no neighbours at all
    top, *_ = _run_topk(pkg, device, [x], [1.0], 20, 1.5)
    assert _to_np(top.count).tolist() == [0] * 5 and np.all(_to_np(top.idx) == -1)
    # duplicates keep score 1.0 and are excluded by index only; zero rows score 0 with everything
    y = np.zeros((6, 8), np.float32)
    y[0] = y[1] = y[2] = [1, 2, 3, 4, 0, 0, 0, 0]
    y[3] = [0, 0, 0, 0, 1, 0, 0, 0]
    top, *_ = _run_topk(pkg, device, [y], [1.0], 3, 0.0)
    idx, score, count = _to_np(top.idx), _to_np(top.score), _to_np(top.count)
    assert idx[0].tolist() == [1, 2, 3] and np.allclose(score[0, :2], 1.0, atol=1e-6) and score[0, 2] == 0
    assert idx[1].tolist() == [0, 2, 3]
    assert idx[4].tolist() == [0, 1, 2] and np.all(score[4] == 0) and count[4] == 3
    # exclude_self off: the diagonal comes first
    top, *_ = _run_topk(pkg, device, [y], [1.0], 2, 0.5, exclude_self=False)
    assert _to_np(top.idx)[0].tolist() == [0, 1] and _to_np(top.idx)[3].tolist() == [3, -1]


@pytest.mark.parametrize("name", ["dense_512x64", "dense_2048x512"])
@pytest.mark.parametrize("k", [20, 50])
def test_golden_dense(pkg, device, name, k):
    g = load_golden(name)
    n, d, seed = int(g["n"]), int(g["d"]), int(g["seed"])
    x = dense_gaussian(n, d, seed)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["x_sha256"]), "input generator drifted"
    top, *_ = _run_topk(pkg, device, [x], [1.0], k, -1.0)
    xn = so.l2_normalize_rows(x.astype(np.float64))
    stats = {}
    check_topk_against_golden(
        g[f"idx_k{k}"], g[f"sorted_scores_k{k}"], k, _to_np(top.idx), _to_np(top.score), _to_np(top.count),
        pair_score=lambda r, cols: xn[r] @ xn[cols].T, what=f"{name} k={k}", clear_mask=g[f"clear_k{k}"], stats=stats)
    # every rank separated from its neighbours by more than 2e-6 in the reference's float64 scores
    # was compared by index (the check above asserts equality there): at least 99 % of all ranks;
    # the others were checked as members of a near-tie group carrying their own correct score
    assert stats["ranks"] == n * k
    assert stats["exact"] >= 0.99 * n * k, stats
    assert stats["max_err"] <= 1e-6, stats  # measured: 2.4e-7 (D = 64) .. 3.6e-7 (D = 512)
    print(f"{name} k={k}: {stats}")


def test_golden_dense_every_rank_exact(pkg, device):
    """The fixture whose seed was searched until the smallest gap between consecutive float64 scores
    over ranks 1..K+1 of all rows exceeded 1e-5 (verified by the generator against the reference):
    "top-K indices exactly" holds for the whole index array."""
    g = load_golden("dense_tiefree_256x64")
    n, d, seed, k = int(g["n"]), int(g["d"]), int(g["seed"]), int(g["k"])
    assert float(g["min_rank_gap"]) > 1e-5
    x = dense_gaussian(n, d, seed)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["x_sha256"]), "input generator drifted"
    top, *_ = _run_topk(pkg, device, [x], [1.0], k, -1.0)
    assert np.array_equal(_to_np(top.idx), g["idx"])
    assert np.array_equal(_to_np(top.count), g["count"])
    assert np.abs(_to_np(top.score).astype(np.float64) - g["score"]).max() <= 1e-6


def test_golden_hybrid(pkg, device):
    g = load_golden("hybrid_2048")
    n, seed, k = int(g["n"]), int(g["seed"]), int(g["k"])
    feats = hybrid_like_blocks(n, seed, d_genre=24, d_text=200, as_features=True)
    gen = np.asarray(feats["genre_features"])
    assert hashlib.sha256(np.ascontiguousarray(gen).tobytes()).hexdigest() == str(g["genre_sha256"])
    blocks, weights = so.hybrid_blocks(feats, *g["weights"].tolist(), variant="computer")
    top, z, offsets, scales = _run_topk(pkg, device, blocks, weights, k, float(g["min_similarity"]))
    zo = so.build_z(blocks, weights)
    check_topk_against_golden(g["idx"], g["sorted_scores"], k, _to_np(top.idx), _to_np(top.score),
                              _to_np(top.count), pair_score=lambda r, cols: zo[r] @ zo[cols].T,
                              min_similarity=float(g["min_similarity"]), what="hybrid_2048")


@pytest.mark.parametrize("n,world", [(700, 2), (1000, 8), (333, 3)])
def test_column_pieces_equal_single_sweep(pkg, device, n, world):
    """A row shard that scores its own columns first and the rest later (two launches, the second
    over two ranges) gets, bit for bit, the lists of one sweep over all columns."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine, sharding

    g, t, p, ty, la = hybrid_like_blocks(n, seed=31, d_genre=16, d_text=90)
    x = np.ascontiguousarray(np.hstack([g, t, p, ty, la]), dtype=np.float32)
    offsets, scales = [0, 16, 106, 138], [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)]
    z = engine.normalize_blocks(torch.from_numpy(x).to(device), offsets, scales)
    k, ms = 20, 0.1
    full = engine.cosine_topk(z, k, ms)
    for rank in range(world):
        sh = sharding.shard_bounds(n, world, rank)
        if sh.rows == 0:
            continue
        own, after, before = sharding.column_pieces(sh)
        zq = z[sh.row0 : sh.row0 + sh.rows].contiguous()  # the rank's own shard, as after normalisation
        ses = engine.TopkSession(zq, sh.rows, sh.row0, k, ms)
        assert ses.accumulate((zq, own[1], own[0])) >= 1
        ses.accumulate((z[after[0]:], after[1], after[0]), (z, before[1], before[0]))
        top = ses.finish()
        torch.cuda.synchronize()
        sl = slice(sh.row0, sh.row0 + sh.rows)
        assert np.array_equal(_to_np(top.idx), _to_np(full.idx)[sl]), rank
        assert np.array_equal(_to_np(top.score), _to_np(full.score)[sl]), rank
        assert np.array_equal(_to_np(top.count), _to_np(full.count)[sl]), rank


def test_large_shape_properties(pkg, device):
    """N = 100k x D = 1024 (the benchmark shape): properties that need no N x N oracle --
    sampled rows against the float64 oracle, order, self exclusion, shard/full equality."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, d, k, ms = 100_000, 1024, 20, 0.1
    gen = torch.Generator(device=device)
    gen.manual_seed(5)
    x = torch.zeros((n, d), device=device)
    x[:, :32] = (torch.rand((n, 32), generator=gen, device=device) < 0.08).float()
    text = torch.randn((n, d - 64), generator=gen, device=device).abs_()
    text *= torch.rand((n, d - 64), generator=gen, device=device) < 0.03
    x[:, 32 : d - 32] = text
    del text
    x[torch.arange(n, device=device), d - 32 + torch.randint(0, 32, (n,), generator=gen, device=device)] = 1.0
    offsets, weights = [0, 32, d - 32, d], [0.4, 0.5, 0.1]
    scales = [float(np.sqrt(w)) for w in weights]
    z = engine.normalize_blocks(x, offsets, scales)
    top = engine.cosine_topk(z, k, ms)
    part = engine.cosine_topk(z, k, ms, q_row0=37_500, nq=12_500)
    torch.cuda.synchronize()
    idx, score, count = _to_np(top.idx), _to_np(top.score), _to_np(top.count)
    assert np.array_equal(_to_np(part.idx), idx[37_500:50_000]) and np.array_equal(_to_np(part.score), score[37_500:50_000])
    valid = idx >= 0
    assert np.array_equal(valid.sum(axis=1), count) and np.all(valid[:, :-1] >= valid[:, 1:])
    assert not np.any(idx == np.arange(n)[:, None])
    assert np.all(np.diff(score, axis=1)[valid[:, 1:]] <= 0) and np.all(score[valid] >= ms)
    # sampled rows against the oracle (float64 on the host)
    rows = np.random.default_rng(0).choice(n, size=48, replace=False)
    xh = x.cpu().numpy().astype(np.float64)
    zo = so.build_z([xh[:, offsets[b] : offsets[b + 1]] for b in range(3)], weights)
    del xh
    h = zo[rows] @ zo.T
    for o, r in enumerate(rows):
        check_topk_against_scores(h[o : o + 1], int(r), k, ms, idx[r : r + 1], score[r : r + 1], count[r : r + 1],
                                  what=f"N=100k row {r}")


def test_wide_features_large_k_two_ranges(pkg, device):
    """The shape class of the largest configuration (D = 2048, K = 50, a row shard in the middle of
    the catalogue, columns given as two ranges): lists in global memory, 64-bit row offsets.
    Sampled rows against the float64 oracle, all rows against a single-range sweep."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, d, k, row0, nq = 150_000, 2048, 50, 70_000, 3000
    gen = torch.Generator(device=device)
    gen.manual_seed(9)
    x = torch.randn((n, d), generator=gen, device=device)
    z = engine.normalize_blocks(x, [0, d], [1.0])
    zq = z[row0:row0 + nq].contiguous()
    ses = engine.TopkSession(zq, nq, row0, k, -1.0)
    ses.accumulate((zq, nq, row0))
    ses.accumulate((z[row0 + nq:], n - row0 - nq, row0 + nq), (z, row0, 0))
    top = ses.finish()
    one = engine.cosine_topk(z, k, -1.0, q_row0=row0, nq=nq)
    torch.cuda.synchronize()
    idx, score, count = _to_np(top.idx), _to_np(top.score), _to_np(top.count)
    assert np.array_equal(idx, _to_np(one.idx)) and np.array_equal(score, _to_np(one.score))
    assert np.all(count == k)
    rows = np.random.default_rng(1).choice(nq, size=12, replace=False)
    zo = so.l2_normalize_rows(x.cpu().numpy().astype(np.float64))
    h = zo[row0 + rows] @ zo.T
    for o, r in enumerate(rows):
        check_topk_against_scores(h[o:o + 1], int(row0 + r), k, -1.0, idx[r:r + 1], score[r:r + 1],
                                  count[r:r + 1], what=f"wide row {r}")


@pytest.mark.parametrize("k", [30, 31, 52, 53, 64, 99])
def test_list_modes_agree_with_oracle_across_k(pkg, device, k):
    """K = 30 | 31 is where the lists move from LDS to global memory: both sides of the boundary,
    and larger K, against the float64 oracle on tie-free and on tie-rich data."""
    n = 1500
    x = dense_gaussian(n, 64, seed=100 + k)
    top, _, _, _ = _run_topk(pkg, device, [x], [1.0], k, -1.0)
    h, _ = _oracle_h([x], [1.0])
    check_topk_against_scores(h, 0, k, -1.0, _to_np(top.idx), _to_np(top.score), _to_np(top.count), what=f"dense k={k}")
    g, t, p, ty, la = hybrid_like_blocks(n, seed=k, d_genre=12, d_text=80)
    blocks, weights = [g, t, np.hstack([p, ty, la])], [0.4, 0.5, 0.1]
    top, _, _, _ = _run_topk(pkg, device, blocks, weights, k, 0.15)
    h, _ = _oracle_h(blocks, weights)
    check_topk_against_scores(h, 0, k, 0.15, _to_np(top.idx), _to_np(top.score), _to_np(top.count), what=f"hybrid k={k}")


@pytest.mark.parametrize("n,d,k,ms", [(3000, 256, 20, 0.1), (1111, 64, 50, -1.0), (5000, 1024, 20, 0.1), (700, 192, 100, -1.0)])
def test_both_tile_loops_give_identical_lists(pkg, device, monkeypatch, n, d, k, ms):
    """The LDS-DMA tile loop and the register-staged loop are two schedules of the same fixed-order
    arithmetic: indices, scores and counts must agree bit for bit (and match the oracle)."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    if ms > 0:
        g, t, p, ty, la = hybrid_like_blocks(n, seed=n, d_genre=32, d_text=d - 64)
        blocks, weights = [g, t, np.hstack([p, ty, la])], [0.4, 0.5, 0.1]
    else:
        blocks, weights = [dense_gaussian(n, d, seed=n + d)], [1.0]
    x = np.ascontiguousarray(np.hstack([np.asarray(b, dtype=np.float32) for b in blocks]))
    offsets = np.concatenate([[0], np.cumsum([b.shape[1] for b in blocks])]).astype(int).tolist()
    z = engine.normalize_blocks(torch.from_numpy(x).to(device), offsets, [float(np.sqrt(w)) for w in weights])
    assert z.shape[1] % 64 == 0  # otherwise only the staged loop can run
    results = {}
    for loop in ("dma", "staged"):
        monkeypatch.setenv("TVBF_TILE_LOOP", loop)
        top = engine.cosine_topk(z, k, ms)
        torch.cuda.synchronize()
        results[loop] = (_to_np(top.idx), _to_np(top.score), _to_np(top.count))
    monkeypatch.delenv("TVBF_TILE_LOOP")
    for a_, b_ in zip(results["dma"], results["staged"]):
        assert np.array_equal(a_, b_)
    h, _ = _oracle_h(blocks, weights)
    check_topk_against_scores(h, 0, k, ms, *results["dma"], what=f"tile loops {n}x{d}")


def test_session_budgets_respect_the_merge_limit(pkg, device):
    """K = 100: one merge takes 20 lists, so the sweeps of a session share that many slots; three
    column pieces must still merge into the lists of one sweep."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, d, k = 6000, 128, 100
    z = engine.normalize_blocks(torch.from_numpy(dense_gaussian(n, d, seed=17)).to(device), [0, d], [1.0])
    full = engine.cosine_topk(z, k, -1.0)
    ses = engine.TopkSession(z, n, 0, k, -1.0, slots=64)
    assert ses.max_lists == 20
    for lo, hi in ((0, 2500), (2500, 4000), (4000, n)):
        ses.accumulate((z[lo:], hi - lo, lo))
    assert ses.used <= ses.max_lists
    top = ses.finish()
    torch.cuda.synchronize()
    for f in ("idx", "score", "count"):
        assert torch.equal(getattr(top, f), getattr(full, f)), f


def test_component_scores_of_mutual_pairs_computed_once(pkg, device, monkeypatch):
    """When the query rows are rows of the catalogue, a pair that sits in both shows' lists is
    computed by one wave and written to both slots: same numbers, bit for bit, as with every pair
    computed on its own -- for the whole catalogue and for a shard of its rows."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, k = 3000, 20
    g, t, p, ty, la = hybrid_like_blocks(n, seed=31, d_genre=16, d_text=208)
    x = np.ascontiguousarray(np.hstack([g, t, p, ty, la]), dtype=np.float32)
    offsets = [0, 16, 224, x.shape[1]]
    scales = [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)]
    z = engine.normalize_blocks(torch.from_numpy(x).to(device), offsets, scales)
    top = engine.cosine_topk(z, k, 0.05)
    idx = top.idx
    lists_of_neighbours = idx[idx.long().clamp(min=0)]  # (n, k, k)
    mutual = (lists_of_neighbours == torch.arange(n, device=device).view(n, 1, 1)).any(dim=2) & (idx >= 0)
    assert 0.2 * n * k < int(mutual.sum()) < n * k  # the case is there, and so is its opposite
    once = engine.component_scores(z, offsets, scales, idx)
    r0, nq = 1111, 700
    shard = engine.component_scores(z, offsets, scales, idx[r0:r0 + nq].contiguous(), q_row0=r0)
    monkeypatch.setenv("TVBF_K5_NO_MUTUAL", "1")
    each = engine.component_scores(z, offsets, scales, idx)
    torch.cuda.synchronize()
    assert torch.equal(once, each) and torch.equal(shard, each[r0:r0 + nq])
