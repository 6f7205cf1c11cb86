"""Ad-hoc single-show queries: memory-bound pass over Z + selection (no n x n matrix)."""

from __future__ import annotations

import numpy as np
import pytest

from oracle import similarity_oracle as so
from parity import SCORE_TOL, check_topk_against_scores
from synth import dense_gaussian, hybrid_like_blocks

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _z(device, blocks, scales):
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    x = np.ascontiguousarray(np.hstack([np.asarray(b, dtype=np.float32) for b in blocks]))
    offsets = np.concatenate([[0], np.cumsum([b.shape[1] for b in blocks])]).astype(int).tolist()
    return engine.normalize_blocks(torch.from_numpy(x).to(device), offsets, scales), offsets


def test_query_rows_match_oracle_and_batch_path(pkg, device):
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, k = 3000, 10
    g, t, p, ty, la = hybrid_like_blocks(n, seed=11, d_genre=24, d_text=300)
    blocks, weights = [g, t, np.hstack([p, ty, la])], [0.4, 0.5, 0.1]
    z, _ = _z(device, blocks, [float(np.sqrt(w)) for w in weights])
    zo = so.build_z([np.asarray(b, dtype=np.float32).astype(np.float64) for b in blocks], weights)
    rows = [0, 5, 1234, 2999, 77, 78, 79, 2000, 13]  # nine queries: two launches of the score kernel
    got = engine.query_topk(z[rows], z, k, 0.0, self_index=rows)
    batch = engine.cosine_topk(z, k, 0.0)
    torch.cuda.synchronize()
    for i, r in enumerate(rows):
        h = zo[r] @ zo.T
        check_topk_against_scores(h[None, :], r, k, 0.0, _np(got.idx)[i:i + 1], _np(got.score)[i:i + 1],
                                  _np(got.count)[i:i + 1], what=f"query row {r}")
        # same neighbours as the batch kernel wherever scores are not tied within the tolerance
        assert np.abs(_np(got.score)[i] - _np(batch.score)[r]).max() <= 4e-6
    # without a self index the show itself comes first (score 1 for a full-featured show)
    got = engine.query_topk(z[5:6], z, k, 0.0)
    torch.cuda.synchronize()
    assert _np(got.idx)[0, 0] == 5 and abs(_np(got.score)[0, 0] - 1.0) <= SCORE_TOL


def test_query_with_other_weights_rescales_only_the_query(pkg, device):
    """Z holds the plain per-block cosine form (unit scales); another weighting is the query
    vector's blocks multiplied by the weights (services/content_based_service.py:133-138)."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, k = 2000, 20
    g, t, p, ty, la = hybrid_like_blocks(n, seed=21, d_genre=16, d_text=120)
    blocks = [g, t, np.hstack([p, ty, la])]
    z, off = _z(device, blocks, [1.0, 1.0, 1.0])
    w = [0.2, 0.7, 0.1]
    row = 321
    q = z[row:row + 1].clone()
    for b in range(3):
        q[:, off[b]:off[b + 1]] *= w[b]
    got = engine.query_topk(q, z, k, 0.05, self_index=[row])
    torch.cuda.synchronize()
    cos = [so.cosine_similarity(np.asarray(b, dtype=np.float32).astype(np.float64)) for b in blocks]
    h = w[0] * cos[0][row] + w[1] * cos[1][row] + w[2] * cos[2][row]
    check_topk_against_scores(h[None, :], row, k, 0.05, _np(got.idx), _np(got.score), _np(got.count),
                              what="re-weighted query")


@pytest.mark.parametrize("n,d,k,nq", [(70, 8, 100, 3), (5000, 8192, 7, 3), (1, 4, 1, 1), (20000, 64, 100, 64)])
def test_query_shapes(pkg, device, n, d, k, nq):
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    x = dense_gaussian(n, d, seed=n + d)
    z, _ = _z(device, [x], [1.0])
    zo = so.l2_normalize_rows(x.astype(np.float64))
    rows = [int(r) for r in np.random.default_rng(3).integers(0, n, size=nq)]
    got = engine.query_topk(z[rows], z, k, -1.0, self_index=rows)
    torch.cuda.synchronize()
    for i, r in enumerate(rows):
        check_topk_against_scores((zo[r] @ zo.T)[None, :], r, k, -1.0, _np(got.idx)[i:i + 1],
                                  _np(got.score)[i:i + 1], _np(got.count)[i:i + 1], what=f"shape {n}x{d} q{i}")


def test_query_threshold_and_argument_errors(pkg, device):
    import torch

    from tvbingefriend_recommendation_service_amd import engine, native

    x = dense_gaussian(500, 32, seed=9)
    z, _ = _z(device, [x], [1.0])
    got = engine.query_topk(z[:2], z, 5, 0.99, self_index=[0, 1])  # nothing is that similar
    torch.cuda.synchronize()
    assert _np(got.count).tolist() == [0, 0] and np.all(_np(got.idx) == -1)
    with pytest.raises(ValueError):
        engine.query_topk(z[:0], z, 5, 0.0)
    with pytest.raises(ValueError):
        engine.query_topk(z[:2], z, 5, 0.0, self_index=[1])
    lib = native.load()
    rc = lib.tvbf_query_topk_f32(z.data_ptr(), 1, 64, None, z.data_ptr(), 500, 64, 64, 5, float("nan"),
                                 z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1 << 30, None)
    assert rc == -1
    assert lib.tvbf_query_workspace_bytes(65, 10, 5) == 0
