"""GPU tests of the drop-in Python surface (SimilarityComputer, the two scripts, the service).

The SimilarityComputer tests restate the reference's own unit tests
(tests/test_ml/test_similarity_computer.py of the reference) against the HIP-backed class.
"""

from __future__ import annotations

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden
from oracle import similarity_oracle as so
from parity import SCORE_TOL, check_topk_against_golden

pytestmark = pytest.mark.gpu

FEATS = ("genre_features", "text_features", "platform_features", "type_features", "language_features")
SIMS = ("genre_similarity", "text_similarity", "metadata_similarity", "hybrid_similarity")
MATRIX_TOL = 1e-12  # float64 inputs are computed in float64 (the reference's own tests use abs=1e-6 and 1 + 1e-10)


@pytest.fixture(scope="module")
def SC(pkg, device):
    from tvbingefriend_recommendation_service_amd.ml.similarity_computer import SimilarityComputer

    return SimilarityComputer


def _golden_features(g):
    f = {k: g[k] for k in FEATS}
    f["text_features"] = sp.csr_matrix(f["text_features"])
    return f


# ---- reference tests/test_ml/test_similarity_computer.py, restated ----

def test_init_weights(SC):
    c = SC()
    assert (c.genre_weight, c.text_weight, c.metadata_weight) == (0.4, 0.5, 0.1)
    c = SC(genre_weight=0.3, text_weight=0.6, metadata_weight=0.1)
    assert (c.genre_weight, c.text_weight, c.metadata_weight) == (0.3, 0.6, 0.1)


def test_genre_similarity_basic_identical_disjoint(SC):
    g = load_golden("conftest_trio")
    s = SC().compute_genre_similarity(g["genre_features"])
    assert s.shape == (3, 3) and s.dtype == np.float64
    assert np.allclose(np.diag(s), 1.0) and np.allclose(s, s.T)
    assert np.all(s >= -1e-10) and np.all(s <= 1 + 1e-10)
    s = SC().compute_genre_similarity(np.array([[1, 1, 0], [1, 1, 0]], dtype=float))
    assert s[0, 1] == pytest.approx(1.0, abs=1e-6) and s[1, 0] == pytest.approx(1.0, abs=1e-6)
    s = SC().compute_genre_similarity(np.array([[1, 0, 0], [0, 1, 0]], dtype=float))
    assert s[0, 1] == pytest.approx(0.0, abs=1e-6)


def test_text_similarity_dense_and_sparse(SC):
    s = SC().compute_text_similarity(np.array([[1.0, 0.5, 0.0], [0.5, 1.0, 0.2], [0.0, 0.2, 1.0]]))
    assert s.shape == (3, 3) and np.allclose(np.diag(s), 1.0) and np.allclose(s, s.T)
    g = load_golden("conftest_trio")
    s = SC().compute_text_similarity(sp.csr_matrix(g["text_features"]))
    assert s.shape == (3, 3) and np.all(s >= -1e-10) and np.all(s <= 1 + 1e-10)
    s32 = SC().compute_text_similarity(g["text_features"].astype(np.float32))
    assert s32.dtype == np.float32  # float32 in -> float32 out, like sklearn
    # float64 in -> float64 arithmetic: the oracle's matrix to the last bits, on dense, CSR, int64 and bool input
    rng = np.random.default_rng(8)
    x = rng.standard_normal((257, 70)) * (rng.random((257, 70)) < 0.3)
    x[5] = 0.0
    want = so.cosine_similarity(x)
    for form in (x, sp.csr_matrix(x)):
        got = SC().compute_text_similarity(form)
        assert got.dtype == np.float64 and np.abs(got - want).max() <= 1e-15 and np.array_equal(got, got.T)
    b = rng.random((64, 9)) < 0.4
    for form in (b, b.astype(np.int64)):
        assert np.abs(SC().compute_genre_similarity(form) - so.cosine_similarity(b.astype(np.float64))).max() <= 1e-15
    with pytest.raises(ValueError, match="Input contains NaN"):
        SC().compute_text_similarity(np.array([[1.0, np.nan], [0.0, 1.0]]))


def test_metadata_similarity(SC):
    g = load_golden("conftest_trio")
    s = SC().compute_metadata_similarity(g["platform_features"], g["type_features"], g["language_features"])
    assert s.shape == (3, 3) and np.allclose(np.diag(s), 1.0) and np.allclose(s, s.T)
    assert np.abs(s - g["metadata_similarity"]).max() <= MATRIX_TOL


def test_hybrid_similarity_known_answers_and_exactness(SC):
    base = np.array([[1.0, 0.8, 0.2], [0.8, 1.0, 0.3], [0.2, 0.3, 1.0]])
    h = SC(0.4, 0.5, 0.1).compute_hybrid_similarity(base, base * 0.9, base * 0.8)
    assert h.shape == (3, 3) and np.allclose(np.diag(h), 0.93, atol=1e-6)
    h = SC(2.0, 3.0, 1.0).compute_hybrid_similarity(
        np.array([[1.0, 0.6], [0.6, 1.0]]), np.array([[1.0, 0.8], [0.8, 1.0]]), np.array([[1.0, 0.4], [0.4, 1.0]]))
    assert h[0, 1] == pytest.approx((2 / 6) * 0.6 + (3 / 6) * 0.8 + (1 / 6) * 0.4, abs=1e-6)
    h = SC(1.0, 1.0, 1.0).compute_hybrid_similarity(
        np.array([[1.0, 0.3], [0.3, 1.0]]), np.array([[1.0, 0.6], [0.6, 1.0]]), np.array([[1.0, 0.9], [0.9, 1.0]]))
    assert h[0, 1] == pytest.approx((0.3 + 0.6 + 0.9) / 3, abs=1e-6)
    # float64, no fused multiply-add: bit for bit the NumPy expression of the reference
    rng = np.random.default_rng(1)
    a, b, c = (rng.random((257, 257)) for _ in range(3))
    got = SC(0.37, 0.21, 0.11).compute_hybrid_similarity(a, b, c)
    assert np.array_equal(got, so.SimilarityComputerOracle(0.37, 0.21, 0.11).compute_hybrid_similarity(a, b, c))


@pytest.mark.parametrize("name", ["conftest_trio", "edge_frame"])
def test_compute_all_similarities_matches_reference(SC, name):
    g = load_golden(name)
    out = SC().compute_all_similarities(_golden_features(g))
    assert set(out) == set(SIMS)
    for k in SIMS:
        assert out[k].shape == g[k].shape
        assert np.abs(out[k] - g[k]).max() <= MATRIX_TOL, k


def test_statistics(SC):
    c = SC()
    st = c.get_similarity_statistics(np.array([[1.0, 0.8, 0.2], [0.8, 1.0, 0.3], [0.2, 0.3, 1.0]]))
    assert set(st) == {"mean", "std", "min", "max", "median"} and all(isinstance(v, float) for v in st.values())
    assert st["min"] == 0.2 and st["max"] == 0.8 and st["median"] == 0.3
    st = c.get_similarity_statistics(np.array([[1.0, 0.5, 0.5], [0.5, 1.0, 0.5], [0.5, 0.5, 1.0]]))
    assert st["mean"] == pytest.approx(0.5, abs=1e-6) and st["max"] == pytest.approx(0.5, abs=1e-6)
    m = np.full((4, 4), 0.7)
    np.fill_diagonal(m, 1.0)
    st = c.get_similarity_statistics(m)
    assert all(st[q] == pytest.approx(0.7, abs=1e-6) for q in ("mean", "median", "min", "max"))
    assert st["std"] == pytest.approx(0.0, abs=1e-6)
    rng = np.random.default_rng(2)
    for n in (2, 3, 64, 777):  # even and odd triangle sizes, negatives included
        m = rng.standard_normal((n, n))
        got, ref = c.get_similarity_statistics(m), so.SimilarityComputerOracle().get_similarity_statistics(m)
        assert got["min"] == ref["min"] and got["max"] == ref["max"] and got["median"] == ref["median"], n
        assert got["mean"] == pytest.approx(ref["mean"], abs=1e-13)
        assert got["std"] == pytest.approx(ref["std"], rel=1e-12)
    with pytest.raises(ValueError):
        c.get_similarity_statistics(np.ones((1, 1)))


# ---- scalable path and the scripts ----

def test_gram_and_topk_share_one_arithmetic(pkg, device):
    """A dense score and the fused path's score of the same pair are the same fp32 number."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    x = np.random.default_rng(3).standard_normal((300, 72)).astype(np.float32)
    z = engine.normalize_blocks(torch.from_numpy(x).to(device), [0, 72], [1.0])
    s = engine.gram(z).cpu().numpy()
    top = engine.cosine_topk(z, 10, -1.0)
    idx, score = top.idx.cpu().numpy(), top.score.cpu().numpy()
    assert np.array_equal(np.take_along_axis(s, idx.astype(np.int64), axis=1), score)
    assert np.array_equal(s, s.T)


def _write_feature_dir(tmp_path, g):
    import pandas as pd

    text = sp.csr_matrix((g["text_data"], g["text_indices"], g["text_indptr"]), shape=tuple(g["text_shape"]))
    np.save(tmp_path / "genre_features.npy", g["genre_features"])
    sp.save_npz(tmp_path / "text_features.npz", text)
    np.save(tmp_path / "platform_features.npy", g["platform_features"])
    np.save(tmp_path / "type_features.npy", g["type_features"])
    np.save(tmp_path / "language_features.npy", g["language_features"])
    pd.DataFrame({"id": g["ids"], "name": [f"s{i}" for i in g["ids"]]}).to_csv(tmp_path / "shows_metadata.csv",
                                                                                 index=False)
    return {"genre_features": g["genre_features"], "text_features": text,
            "platform_features": g["platform_features"], "type_features": g["type_features"],
            "language_features": g["language_features"]}


def test_populate_plumbing_100_shows(pkg, device, tmp_path):
    """The reference's smallest configuration (100 shows, top-20, min 0.1) end to end:
    feature files -> compute_and_store_similarities -> sink, against the reference's result."""
    from tvbingefriend_recommendation_service_amd.scripts import populate_database as pd_amd

    g = load_golden("plumbing_100")
    feats = _write_feature_dir(tmp_path, g)
    sink = pd_amd.MemorySimilaritySink()
    cleared = []
    stats = pd_amd.compute_and_store_similarities(tmp_path, 0.4, 0.5, 0.1, top_n_per_show=20, min_similarity=0.1,
                                                  repo=sink, clear_existing=lambda: cleared.append(1))
    assert cleared == [1] and sink.calls == 1
    assert stats["top_n_per_show"] == 20 and stats["min_similarity"] == 0.1
    assert stats["total_records"] == sum(len(v) for v in sink.records.values())
    ids = g["ids"].tolist()
    n, k = len(ids), 20
    idx = np.full((n, k), -1, np.int32)
    score = np.zeros((n, k))
    count = np.zeros(n, np.int32)
    comp = np.zeros((n, k, 3))
    for r, sid in enumerate(ids):
        recs = sink.records.get(sid, [])
        count[r] = len(recs)
        for s, rec in enumerate(recs):
            assert set(rec) == {"similar_show_id", "similarity_score", "genre_score", "text_score", "metadata_score"}
            assert all(isinstance(rec[q], float) for q in rec if q != "similar_show_id")
            idx[r, s] = ids.index(rec["similar_show_id"])
            score[r, s] = rec["similarity_score"]
            comp[r, s] = [rec["genre_score"], rec["text_score"], rec["metadata_score"]]
    h = g["hybrid_similarity"]
    hh = h.copy()
    np.fill_diagonal(hh, -np.inf)
    srt = -np.sort(-hh, axis=1)[:, : k + 1]
    check_topk_against_golden(g["idx"], srt, k, idx, score, count, pair_score=lambda r, cols: h[r, cols],
                              min_similarity=0.1, what="plumbing_100")
    # component scores against the oracle's per-block cosines of the same pairs
    blocks, _ = so.hybrid_blocks(feats, 0.4, 0.5, 0.1, variant="populate")
    bn = [so.l2_normalize_rows(b) for b in blocks]
    for r in range(n):
        j = idx[r, : count[r]]
        ref_g, ref_t = bn[0][r] @ bn[0][j].T, bn[1][r] @ bn[1][j].T
        ref_m = (bn[2][r] @ bn[2][j].T + bn[3][r] @ bn[3][j].T + bn[4][r] @ bn[4][j].T) / 3
        assert np.abs(comp[r, : count[r], 0] - ref_g).max(initial=0) <= SCORE_TOL
        assert np.abs(comp[r, : count[r], 1] - ref_t).max(initial=0) <= SCORE_TOL
        assert np.abs(comp[r, : count[r], 2] - ref_m).max(initial=0) <= SCORE_TOL


def test_threshold_met_exactly_in_float64_keeps_the_pair(pkg, device):
    """A show without genres and without a summary matches others on platform, type and language
    only: 0.4 * 0 + 0.5 * 0 + 0.1 * ((1 + 1 + 1) / 3) == 0.1 in the reference's float64 arithmetic,
    and `score < min_similarity -> break` (scripts/populate_database.py:204-205) keeps such pairs,
    while the fp32 sum of the three weighted one-hot products is 0.099999994.  The production path
    (variant "populate", default weights, min 0.1) must keep exactly what the reference keeps."""
    from synth import hybrid_like_blocks

    from tvbingefriend_recommendation_service_amd.scripts import populate_database as pd_amd

    n, k, ms = 700, 20, 0.1
    feats = hybrid_like_blocks(n, seed=2024, d_genre=16, d_text=48, as_features=True)
    genre = np.array(feats["genre_features"])
    text = feats["text_features"].toarray()
    bare = np.arange(0, n, 9)  # 78 shows with no genre and no summary at all
    genre[bare] = 0
    text[bare] = 0.0
    feats["genre_features"], feats["text_features"] = genre, sp.csr_matrix(text)
    ids = list(range(n))
    want = so.populate_rows(feats, ids, 0.4, 0.5, 0.1, k, ms, order_fn=so.deterministic_order)
    got = pd_amd.compute_top_neighbours(feats, 0.4, 0.5, 0.1, top_n_per_show=k, min_similarity=ms)
    want_count = np.array([len(want.get(i, [])) for i in ids])
    assert want_count[bare].sum() > 200  # the case is really there: metadata-only neighbours at exactly 0.1
    assert np.array_equal(got.count, want_count)
    for i in bare:  # all their neighbours score exactly 0.1: index order, hence the reference's set under this order
        c = int(want_count[i])
        assert got.idx[i, :c].tolist() == [r["similar_show_id"] for r in want.get(int(i), [])]
        assert np.all(np.abs(got.score[i, :c] - 0.1) <= 1e-7) and np.all(got.idx[i, c:] == -1)
        assert np.allclose(got.comp[i, :c], [0.0, 0.0, 1.0], atol=1e-6) and np.all(got.comp[i, c:] == 0)
    for i in ids:  # and everywhere else: same neighbours, scores within the tolerance
        c = int(want_count[i])
        ref = want.get(i, [])
        assert np.abs(got.score[i, :c] - np.array([r["similarity_score"] for r in ref])).max(initial=0) <= SCORE_TOL
        assert set(got.idx[i, :c].tolist()) == {r["similar_show_id"] for r in ref} or c == k


def test_compute_similarities_script(pkg, device, tmp_path):
    from tvbingefriend_recommendation_service_amd.scripts import compute_similarities as cs

    with pytest.raises(FileNotFoundError, match="Run compute_features.py first"):
        cs.load_features(tmp_path)
    g = load_golden("plumbing_100")
    feats = _write_feature_dir(tmp_path, g)
    loaded = cs.load_features(tmp_path)
    assert set(loaded) == set(FEATS) and sp.issparse(loaded["text_features"])
    sims = cs.compute_similarities(loaded, 0.4, 0.5, 0.1)
    assert np.abs(sims["hybrid_similarity"] - g["hybrid_similarity"]).max() <= MATRIX_TOL
    cs.save_similarities(sims, tmp_path / "out", save_to_disk=False)
    assert not (tmp_path / "out").exists()
    cs.save_similarities(sims, tmp_path / "out", save_to_disk=True)
    assert sorted(p.name for p in (tmp_path / "out").iterdir()) == sorted(f"{k}.npy" for k in SIMS)
    with pytest.raises(SystemExit) as e:
        cs.main(["--input-dir", str(tmp_path), "--genre-weight", "0", "--text-weight", "0", "--metadata-weight", "0"])
    assert e.value.code == 1
    with pytest.raises(SystemExit):
        cs.main(["--input-dir", str(tmp_path / "missing")])
    out = cs.main(["--input-dir", str(tmp_path)])
    assert set(out) == set(SIMS)


def test_service_matrix_path(pkg, device, tmp_path):
    """get_recommendations_from_matrix / compute_and_store_all_similarities on stored matrices:
    float64 scores are copied, so everything but the order inside tie groups is exact."""
    from tvbingefriend_recommendation_service_amd.scripts.populate_database import MemorySimilaritySink
    from tvbingefriend_recommendation_service_amd.services.content_based_service import (
        ContentBasedRecommendationService,
    )

    g = load_golden("plumbing_100")
    feats = _write_feature_dir(tmp_path, g)
    o = so.SimilarityComputerOracle().compute_all_similarities(feats)
    for k in SIMS[:3]:
        np.save(tmp_path / f"{k}.npy", o[k])
    ids = g["ids"].tolist()
    svc = ContentBasedRecommendationService(tmp_path, 2.0, 3.0, 1.0)
    hyb = so.service_hybrid(o["genre_similarity"], o["text_similarity"], o["metadata_similarity"], 2.0, 3.0, 1.0)
    svc._load_similarity_matrices()
    assert np.array_equal(svc._hybrid_similarity.cpu().numpy(), hyb)
    for sid in ids[:25]:
        for n_, ms in ((10, 0.0), (5, 0.3), (50, 0.1), (0, 0.0)):
            got = svc.get_recommendations_from_matrix(sid, n=n_, min_similarity=ms)
            ref = so.recommendations_from_matrix(sid, ids, hyb, o["genre_similarity"], o["text_similarity"],
                                                 o["metadata_similarity"], n=n_, min_similarity=ms,
                                                 order_fn=so.deterministic_order)
            assert got == ref, (sid, n_, ms)
    assert svc.get_recommendations_from_matrix(-5) == []
    sink = MemorySimilaritySink()
    stats = svc.compute_and_store_all_similarities(top_n_per_show=20, min_similarity=0.1, repo=sink)
    assert stats["computed_shows"] == len(sink.records) and stats["top_n_per_show"] == 20
    for sid in ids:
        ref = so.recommendations_from_matrix(sid, ids, hyb, o["genre_similarity"], o["text_similarity"],
                                             o["metadata_similarity"], n=20, min_similarity=0.1,
                                             order_fn=so.deterministic_order)
        ref = [{("similar_show_id" if k == "show_id" else k): v for k, v in r.items()} for r in ref]
        assert sink.records.get(sid, []) == ref


def test_service_feature_path_matches_matrix_semantics(pkg, device, tmp_path):
    """get_recommendations_from_features: same records as the matrix lookup (scores within the
    fp32 tolerance), also with per-call weights; no matrix files are needed."""
    from parity import SCORE_TOL
    from tvbingefriend_recommendation_service_amd.services.content_based_service import (
        ContentBasedRecommendationService,
    )

    g = load_golden("plumbing_100")
    feats = _write_feature_dir(tmp_path, g)
    o = so.SimilarityComputerOracle().compute_all_similarities(feats)
    ids = g["ids"].tolist()
    svc = ContentBasedRecommendationService(tmp_path, 0.4, 0.5, 0.1)
    for weights in (None, (1.0, 1.0, 2.0), (0.0, 1.0, 0.0)):
        gw, tw, mw = weights or (0.4, 0.5, 0.1)
        hyb = so.service_hybrid(o["genre_similarity"], o["text_similarity"], o["metadata_similarity"], gw, tw, mw)
        for sid in ids[:30]:
            for n_, ms in ((10, 0.0), (5, 0.3), (0, 0.0)):
                kw = {} if weights is None else dict(genre_weight=gw, text_weight=tw, metadata_weight=mw)
                got = svc.get_recommendations_from_features(sid, n=n_, min_similarity=ms, **kw)
                ref = so.recommendations_from_matrix(sid, ids, hyb, o["genre_similarity"], o["text_similarity"],
                                                     o["metadata_similarity"], n=n_, min_similarity=ms,
                                                     order_fn=so.deterministic_order)
                row = ids.index(sid)
                # ranks whose neighbours are further than the tolerance apart must agree exactly
                ref_scores = [r["similarity_score"] for r in ref]
                assert abs(len(got) - len(ref)) <= 1, (sid, n_, ms)
                for s_, (a, b) in enumerate(zip(got, ref)):
                    assert abs(a["similarity_score"] - b["similarity_score"]) <= SCORE_TOL
                    clear = all(abs(ref_scores[s_] - x) > 2 * SCORE_TOL for i_, x in enumerate(ref_scores) if i_ != s_)
                    last = s_ == len(ref) - 1  # the first left-out neighbour may tie with the last kept one
                    if clear and not last:
                        assert a["show_id"] == b["show_id"], (sid, n_, ms, s_)
                    j = ids.index(a["show_id"])
                    assert abs(a["similarity_score"] - hyb[row, j]) <= SCORE_TOL
                    assert abs(a["genre_score"] - o["genre_similarity"][row, j]) <= SCORE_TOL
                    assert abs(a["text_score"] - o["text_similarity"][row, j]) <= SCORE_TOL
                    assert abs(a["metadata_score"] - o["metadata_similarity"][row, j]) <= SCORE_TOL
                    assert a["show_id"] != sid
    assert svc.get_recommendations_from_features(-5) == []


def test_populate_columnar_sink_equals_dict_sink(pkg, device, tmp_path):
    """An array-native sink gets the same pairs, in the same order, with the same values as the
    record dicts handed to a SimilarityRepository-like sink."""
    from tvbingefriend_recommendation_service_amd.scripts import populate_database as pd_amd

    g = load_golden("plumbing_100")
    _write_feature_dir(tmp_path, g)
    dict_sink, col_sink = pd_amd.MemorySimilaritySink(), pd_amd.ColumnarSimilaritySink()
    s1 = pd_amd.compute_and_store_similarities(tmp_path, 0.4, 0.5, 0.1, repo=dict_sink)
    s2 = pd_amd.compute_and_store_similarities(tmp_path, 0.4, 0.5, 0.1, repo=col_sink)
    assert s1["total_records"] == s2["total_records"] and s1["unique_shows"] == s2["unique_shows"]
    flat = [(sid, r["similar_show_id"], r["similarity_score"], r["genre_score"], r["text_score"], r["metadata_score"])
            for sid, items in dict_sink.records.items() for r in items]
    rows = [row for batch in col_sink.rows(1000) for row in batch]
    assert rows == flat


def test_rccl_route_one_rank_group(pkg, device):
    """The collective route of sharded_topk (asynchronous all_gather_into_tensor over RCCL, own
    columns scored meanwhile, the rest after work.wait()) on a one-rank process group: the lists
    equal the plain single-sweep result bit for bit."""
    import socket

    import torch
    import torch.distributed as dist

    from synth import hybrid_like_blocks
    from tvbingefriend_recommendation_service_amd import engine, sharding

    n, k, ms = 3000, 20, 0.1
    g, t, p, ty, la = hybrid_like_blocks(n, seed=77, d_genre=16, d_text=208)
    x = torch.from_numpy(np.ascontiguousarray(np.hstack([g, t, p, ty, la]), dtype=np.float32)).to(device)
    offsets, scales = [0, 16, 224, 256], [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)]
    want = engine.cosine_topk(engine.normalize_blocks(x, offsets, scales), k, ms)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=device)
    try:
        sh = sharding.shard_bounds(n, 1, 0)
        for _ in range(3):  # repeated steps reuse the communicator
            top, z_all = sharding.sharded_topk(x, offsets, scales, sh, k, ms, collective=True)
            torch.cuda.synchronize()
            assert z_all.shape[0] == n
            assert torch.equal(top.idx, want.idx) and torch.equal(top.score, want.score)
            assert torch.equal(top.count, want.count)
        # the symmetric route's collectives (int64 all-gather and all-to-all, int32 max-reduction)
        for _ in range(2):
            top, z_all = sharding.sharded_topk_symmetric(x, offsets, scales, sh, k, ms)
            torch.cuda.synchronize()
            assert torch.equal(top.idx, want.idx) and torch.equal(top.score, want.score)
            assert torch.equal(top.count, want.count)
        flag = torch.tensor([3.0], dtype=torch.float64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)  # the reduction bench.py uses for its timings
        assert float(flag.item()) == 3.0
        dist.barrier()
        # the drop-in's sharded call over RCCL (device tensors in every collective, lists gathered on the device)
        from tvbingefriend_recommendation_service_amd.ml.similarity_computer import SimilarityComputer

        feats = hybrid_like_blocks(n, seed=77, d_genre=16, d_text=208, as_features=True)
        sc = SimilarityComputer(0.4, 0.5, 0.1, device=device)
        single = sc.topk(feats, k=k, min_similarity=ms, variant="populate")
        sharded = sc.topk(feats, k=k, min_similarity=ms, variant="populate", group=True)
        for f in ("idx", "score", "count", "comp"):
            assert np.array_equal(getattr(single, f), getattr(sharded, f)), f
    finally:
        dist.destroy_process_group()


def test_whole_step_can_be_captured_in_a_graph(pkg, device, monkeypatch):
    """K1, the catalogue top-K, K5, the one-sweep statistics (with its gated fall-back launches) and a
    batch of single-show queries captured as ONE HIP graph: the library clears memory with kernels
    and synchronises nothing, so every replay -- on new input data, with unrelated eager work and
    synchronisations in between -- must reproduce the eager results bit for bit."""
    import torch

    from tvbingefriend_recommendation_service_amd import engine

    n, d, k, ms = 3000, 256, 20, 0.05
    offsets, scales = [0, 32, 224, 256], [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)]
    monkeypatch.setenv("TVBF_STATS_ROUTE", "band")
    gen = torch.Generator(device=device)

    def data(seed):
        gen.manual_seed(seed)
        x = torch.randn((n, d), generator=gen, device=device).abs_()
        x *= torch.rand((n, d), generator=gen, device=device) < 0.2
        return x

    x_buf = data(0)
    z_buf = torch.empty((n, d), device=device)

    def step():
        z = engine.normalize_blocks(x_buf, offsets, scales, out=z_buf)
        top = engine.catalogue_topk(z, k, ms)
        comp = engine.component_scores(z, offsets, scales, top.idx)
        stats = engine.gram_stats(z)
        q = engine.query_topk(z[:5], z, k, ms, self_index=[0, 1, 2, 3, 4])
        return top, comp, stats, q

    def flat(res):
        top, comp, stats, q = res
        return [top.idx, top.score, top.count, comp, stats, q.idx, q.score, q.count]

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        step()  # first call outside the capture (kernel attributes, library init)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        captured = flat(step())
    for seed in (11, 12, 13):
        x_buf.copy_(data(seed))
        want = [t.clone() for t in flat(step())]  # eager, same input buffer
        for t in captured:
            t.zero_()
        _ = torch.randn((128, 128), device=device) @ torch.randn((128, 128), device=device)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(captured, want)):
            assert torch.equal(a, b), f"replay with seed {seed}: output {i} differs"


def test_concurrent_calls_from_two_threads_on_two_streams(pkg, device):
    """The library keeps no state between calls (the static-scratch entry point aside): two host
    threads, each on its own stream with its own workspace and catalogue, interleave symmetric and
    plain calls and statistics -- every result equals the one computed alone."""
    import threading

    import torch

    from tvbingefriend_recommendation_service_amd import engine

    k = 20
    gen = torch.Generator(device=device)
    cats = []
    for seed, n, d in ((1, 5000, 128), (2, 3300, 256)):
        gen.manual_seed(seed)
        z = engine.normalize_blocks(torch.randn((n, d), generator=gen, device=device), [0, d], [1.0])
        cats.append((z, engine.catalogue_topk(z, k, -1.0), engine.gram_stats(z)))
    torch.cuda.synchronize()
    errors = []

    def worker(which):
        try:
            z, want, want_stats = cats[which]
            stream = torch.cuda.Stream(device=device)
            with torch.cuda.stream(stream):
                for rep in range(6):
                    got = engine.catalogue_topk(z, k, -1.0) if rep % 2 == 0 else engine.cosine_topk(z, k, -1.0)
                    st = engine.gram_stats(z)
                    stream.synchronize()
                    for f in ("idx", "score", "count"):
                        if not torch.equal(getattr(got, f), getattr(want, f)):
                            errors.append(f"thread {which} rep {rep}: {f} differs")
                    if not torch.equal(st, want_stats):
                        errors.append(f"thread {which} rep {rep}: statistics differ")
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {which}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a worker did not finish"
    assert not errors, errors


def test_populate_on_two_ranks_equals_single_gpu(pkg, device, tmp_path):
    """compute_and_store_similarities(gpus=2): two fresh rank processes (gloo, sharing this GPU:
    the exchange goes through host memory) read the feature files, each builds and sweeps its shard,
    rank 0 hands the gathered lists back -- the sink receives exactly the records of the single-GPU
    call (scripts/populate_database.py:85-259 with the shows split over two ranks)."""
    from tvbingefriend_recommendation_service_amd.scripts import populate_database as pd_amd

    g = load_golden("plumbing_100")
    _write_feature_dir(tmp_path, g)
    one, two = pd_amd.MemorySimilaritySink(), pd_amd.MemorySimilaritySink()
    s1 = pd_amd.compute_and_store_similarities(tmp_path, 0.4, 0.5, 0.1, repo=one)
    s2 = pd_amd.compute_and_store_similarities(tmp_path, 0.4, 0.5, 0.1, repo=two, gpus=2, backend="gloo")
    assert s1["total_records"] == s2["total_records"] > 0
    assert one.records == two.records
    # a larger catalogue whose shards take the symmetric sharded sweep with stored own blocks
    from synth import write_feature_files

    big = tmp_path / "big"
    big.mkdir()
    write_feature_files(big, 6000, seed=3)
    one, two = pd_amd.ColumnarSimilaritySink(), pd_amd.ColumnarSimilaritySink()
    pd_amd.compute_and_store_similarities(big, repo=one)
    pd_amd.compute_and_store_similarities(big, repo=two, gpus=3, backend="gloo")
    for c in pd_amd.TABLE_COLUMNS:
        assert np.array_equal(one.table[c], two.table[c]), c
