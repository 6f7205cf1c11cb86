"""CPU tests: the oracle against the golden vectors produced by the real reference
(tests/golden/generate_golden.py) and against the known answers of the reference's
own unit tests (tests/test_ml/test_similarity_computer.py of the reference)."""

from __future__ import annotations

import hashlib

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden
from oracle import similarity_oracle as so
from synth import dense_gaussian, hybrid_like_blocks

FEATS = ("genre_features", "text_features", "platform_features", "type_features", "language_features")
SIMS = ("genre_similarity", "text_similarity", "metadata_similarity", "hybrid_similarity")


def _features(g, sparse_text=True):
    f = {k: g[k] for k in FEATS}
    if sparse_text:
        f["text_features"] = sp.csr_matrix(f["text_features"])
    return f


@pytest.mark.parametrize("name", ["conftest_trio", "edge_frame"])
@pytest.mark.parametrize("sparse_text", [True, False])
def test_all_similarities_match_reference(name, sparse_text):
    g = load_golden(name)
    out = so.SimilarityComputerOracle().compute_all_similarities(_features(g, sparse_text))
    for k in SIMS:
        assert out[k].dtype == np.float64
        assert np.abs(out[k] - g[k]).max() <= 1e-15, k


def test_conftest_trio_known_values():
    """Values quoted by the reference fixtures: genre cos(3 genres, 2 of them) = sqrt(2/3)."""
    g = load_golden("conftest_trio")
    assert g["genre_similarity"][0, 1] == pytest.approx(np.sqrt(2 / 3), abs=1e-12)
    assert np.allclose(g["text_similarity"], np.eye(3), atol=1e-12)
    assert g["metadata_similarity"][0, 2] == pytest.approx(2 / 3, abs=1e-12)
    assert g["hybrid_similarity"][0, 1] == pytest.approx(0.4 * np.sqrt(2 / 3) + 0.1, abs=1e-12)


def test_edge_frame_zero_rows():
    """No genres / empty summary -> zero rows: similarity 0 to everything, itself included;
    a NaN type gives an all-false type row but platform 'other' and language still count."""
    g = load_golden("edge_frame")
    assert np.diag(g["genre_similarity"]).tolist() == pytest.approx([1, 1, 0, 1])
    assert np.diag(g["text_similarity"]).tolist() == pytest.approx([1, 1, 0, 1])
    assert np.diag(g["metadata_similarity"]).tolist() == pytest.approx([1, 1, 1, 1])
    out = so.SimilarityComputerOracle().compute_all_similarities(_features(g))
    assert np.all(out["genre_similarity"][2] == 0) and np.all(out["text_similarity"][:, 2] == 0)


# ---- known answers of the reference's tests/test_ml/test_similarity_computer.py ----

def test_reference_unit_test_known_answers():
    c = so.SimilarityComputerOracle()
    assert (c.genre_weight, c.text_weight, c.metadata_weight) == (0.4, 0.5, 0.1)  # :18-20
    s = c.compute_genre_similarity(np.array([[1, 1, 0], [1, 1, 0]], dtype=float))
    assert s[0, 1] == pytest.approx(1.0, abs=1e-6) and s[1, 0] == pytest.approx(1.0, abs=1e-6)  # :54-66
    s = c.compute_genre_similarity(np.array([[1, 0, 0], [0, 1, 0]], dtype=float))
    assert s[0, 1] == pytest.approx(0.0, abs=1e-6)  # :68-79
    s = c.compute_text_similarity(np.array([[1.0, 0.5, 0.0], [0.5, 1.0, 0.2], [0.0, 0.2, 1.0]]))
    assert np.allclose(np.diag(s), 1.0) and np.allclose(s, s.T)  # :85-97
    base = np.array([[1.0, 0.8, 0.2], [0.8, 1.0, 0.3], [0.2, 0.3, 1.0]])
    h = so.SimilarityComputerOracle(0.4, 0.5, 0.1).compute_hybrid_similarity(base, base * 0.9, base * 0.8)
    assert np.allclose(np.diag(h), 0.93, atol=1e-6)  # :158-177
    h = so.SimilarityComputerOracle(2.0, 3.0, 1.0).compute_hybrid_similarity(
        np.array([[1.0, 0.6], [0.6, 1.0]]), np.array([[1.0, 0.8], [0.8, 1.0]]), np.array([[1.0, 0.4], [0.4, 1.0]]))
    assert h[0, 1] == pytest.approx((2 / 6) * 0.6 + (3 / 6) * 0.8 + (1 / 6) * 0.4, abs=1e-6)  # :179-195
    h = so.SimilarityComputerOracle(1.0, 1.0, 1.0).compute_hybrid_similarity(
        np.array([[1.0, 0.3], [0.3, 1.0]]), np.array([[1.0, 0.6], [0.6, 1.0]]), np.array([[1.0, 0.9], [0.9, 1.0]]))
    assert h[0, 1] == pytest.approx((0.3 + 0.6 + 0.9) / 3, abs=1e-6)  # :197-211
    st = c.get_similarity_statistics(np.array([[1.0, 0.5, 0.5], [0.5, 1.0, 0.5], [0.5, 0.5, 1.0]]))
    assert st["mean"] == pytest.approx(0.5, abs=1e-6) and st["max"] == pytest.approx(0.5, abs=1e-6)  # :280-294
    m = np.full((4, 4), 0.7)
    np.fill_diagonal(m, 1.0)
    st = c.get_similarity_statistics(m)
    assert all(st[q] == pytest.approx(0.7, abs=1e-6) for q in ("mean", "median", "min", "max"))  # :296-312
    assert st["std"] == pytest.approx(0.0, abs=1e-6)
    assert all(isinstance(v, float) for v in st.values())


# ---- top-K selection against the reference's loop ----

@pytest.mark.parametrize("name,k", [("dense_512x64", 20), ("dense_512x64", 50), ("dense_2048x512", 20)])
def test_dense_topk_matches_reference(name, k):
    g = load_golden(name)
    n, d, seed = int(g["n"]), int(g["d"]), int(g["seed"])
    x = dense_gaussian(n, d, seed)
    assert hashlib.sha256(x.tobytes()).hexdigest() == str(g["x_sha256"])
    h = so.SimilarityComputerOracle().compute_genre_similarity(x.astype(np.float64))
    rows = range(n) if n <= 512 else range(0, n, 8)
    hr = h[list(rows)]
    for fn in (so.topk_rows, so.topk_fast):
        for o, r in enumerate(rows):
            i1, s1, c1 = fn(hr[o : o + 1], r, k, -1.0)
            # continuous data: no exact ties, so the reference's argsort order is the unique order
            assert np.array_equal(i1[0], g[f"idx_k{k}"][r]), (fn.__name__, r)
            assert np.array_equal(s1[0], g[f"score_k{k}"][r])
            assert c1[0] == g[f"count_k{k}"][r]
    st = so.SimilarityComputerOracle().get_similarity_statistics(h)
    assert np.allclose([st[q] for q in ("mean", "std", "min", "max", "median")],
                       g["stats_mean_std_min_max_median"], rtol=0, atol=1e-15)


def test_hybrid_topk_and_z_identity():
    g = load_golden("hybrid_2048")
    n, seed, k, ms = int(g["n"]), int(g["seed"]), int(g["k"]), float(g["min_similarity"])
    feats = hybrid_like_blocks(n, seed, d_genre=24, d_text=200, as_features=True)
    out = so.SimilarityComputerOracle(*g["weights"]).compute_all_similarities(feats)
    h = out["hybrid_similarity"]
    # Z identity: one product of the weighted, block-normalised features is the hybrid matrix
    z = so.build_z(*so.hybrid_blocks(feats, *g["weights"], variant="computer"))
    assert np.abs(z @ z.T - h).max() <= 5e-16
    # the production variant (mean of three cosines, raw weights) agrees when every
    # metadata block is one-hot and the weights sum to 1
    zp = so.build_z(*so.hybrid_blocks(feats, *g["weights"], variant="populate"))
    assert np.abs(zp @ zp.T - h).max() <= 5e-16
    idx, score, count = so.topk_fast(h, 0, k, ms)
    assert np.array_equal(count, g["count"])
    # sorted score sequences do not depend on the order inside tie groups; the sparse text
    # product is summed in a different order than SciPy's, hence the last-bit tolerance
    assert np.abs(score - g["score"]).max() <= 1e-15
    # same neighbour sets wherever no tie straddles the K boundary; the order inside groups of
    # equal scores is NumPy's unstable argsort in the reference and index order here
    srt = g["sorted_scores"]
    for r in range(n):
        c = count[r]
        straddle = c == k and abs(srt[r, k] - srt[r, k - 1]) <= 1e-12
        if not straddle:
            assert set(idx[r, :c].tolist()) == set(g["idx"][r, :c].tolist()), r
    for b, name in enumerate(("genre_similarity", "text_similarity", "metadata_similarity")):
        for r in range(0, n, 97):
            c = g["count"][r]
            assert np.abs(out[name][r, g["idx"][r, :c]] - g["comp"][r, :c, b]).max(initial=0) <= 1e-15
    for q in ("mean", "std", "min", "max", "median"):
        got = [so.SimilarityComputerOracle().get_similarity_statistics(out[m])[q] for m in SIMS]
        assert np.allclose(got, g[f"stats_{q}"], rtol=0, atol=1e-15), q


def test_populate_rows_match_reference_plumbing():
    """100-show catalogue: the restated production row loop against the reference's
    matrices + selection loop (stored by the golden generator)."""
    g = load_golden("plumbing_100")
    text = sp.csr_matrix((g["text_data"], g["text_indices"], g["text_indptr"]), shape=tuple(g["text_shape"]))
    feats = {"genre_features": g["genre_features"], "text_features": text,
             "platform_features": g["platform_features"], "type_features": g["type_features"],
             "language_features": g["language_features"]}
    ids = g["ids"].tolist()
    out = so.populate_rows(feats, ids, 0.4, 0.5, 0.1, 20, 0.1, order_fn=so.deterministic_order)
    h = g["hybrid_similarity"]
    for r, sid in enumerate(ids):
        c = int(g["count"][r])
        if c == 0:
            assert sid not in out  # scripts/populate_database.py:220-221
            continue
        recs = out[sid]
        assert len(recs) == c
        got_scores = np.array([x["similarity_score"] for x in recs])
        assert np.abs(got_scores - g["score"][r, :c]).max() <= 1e-12
        got_idx = [ids.index(x["similar_show_id"]) for x in recs]
        assert np.abs(h[r, got_idx] - got_scores).max() <= 1e-12
        assert all(isinstance(x[q], float) for x in recs
                   for q in ("similarity_score", "genre_score", "text_score", "metadata_score"))
        # component scores of the same pairs
        ref_pairs = {int(j): g["comp"][r, s] for s, j in enumerate(g["idx"][r, :c])}
        for x, j in zip(recs, got_idx):
            if j in ref_pairs:
                assert abs(x["genre_score"] - ref_pairs[j][0]) <= 1e-12
                assert abs(x["text_score"] - ref_pairs[j][1]) <= 1e-12
                assert abs(x["metadata_score"] - ref_pairs[j][2]) <= 1e-12


def test_service_matrix_path_semantics():
    """get_recommendations_from_matrix restated (services/content_based_service.py:161-236):
    properties pinned by the reference's tests/test_services/test_content_based_service.py:243-342."""
    rng = np.random.default_rng(0)
    g, t, m = (rng.random((6, 6)) for _ in range(3))
    ids = [10, 20, 30, 40, 50, 60]
    h = so.service_hybrid(g, t, m, 0.4, 0.5, 0.1)
    recs = so.recommendations_from_matrix(30, ids, h, g, t, m, n=3, min_similarity=0.2)
    assert len(recs) <= 3 and all(r["show_id"] != 30 for r in recs)
    assert all(r["similarity_score"] >= 0.2 for r in recs)
    sc = [r["similarity_score"] for r in recs]
    assert sc == sorted(sc, reverse=True)
    assert set(recs[0]) == {"show_id", "similarity_score", "genre_score", "text_score", "metadata_score"}
    assert so.recommendations_from_matrix(999, ids, h, g, t, m) == []


def test_walk_semantics():
    h = np.array([0.9, 0.5, 0.5, 0.05, 1.0])
    # self excluded by index even when a duplicate has the same score
    assert so.walk_row(so.deterministic_order(h), h, 4, 10, 0.1) == [0, 1, 2]
    assert so.walk_row(so.deterministic_order(h), h, 0, 2, 0.0) == [4, 1]
    # threshold applies before the K cut and to the score itself
    assert so.walk_row(so.deterministic_order(h), h, 3, 10, 0.5) == [4, 0, 1, 2]
    assert so.walk_row(so.deterministic_order(h), h, 3, 10, 1.1) == []
