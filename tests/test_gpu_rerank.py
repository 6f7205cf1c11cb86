"""GPU tests of the float64 re-rank (tvbf_block_norms_f64, tvbf_rescore_topk_f64,
tvbf_score_rows_f64; engine.exact_topk): the lists must be the reference's own float64 decisions
(scripts/populate_database.py:180-218) -- indices equal to the float64 oracle's at every rank that
is not an exact float64 tie, scores within 1e-12, counts equal."""

from __future__ import annotations

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden
from oracle import similarity_oracle as so
from parity import REFINED_GAP, REFINED_TOL, check_topk_against_scores
from synth import hybrid_like_blocks

pytestmark = pytest.mark.gpu


def _oracle_hybrid(feats, weights, variant):
    """Float64 hybrid matrix of the oracle for the variant ("computer": ml/similarity_computer.py:122-124
    with weights divided by their sum; "populate": scripts/populate_database.py:184-192)."""
    blocks, ws = so.hybrid_blocks(feats, *weights, variant=variant)
    cos = [so.cosine_similarity(b) for b in blocks]
    if variant == "populate":
        meta = (cos[2] + cos[3] + cos[4]) / 3
        return weights[0] * cos[0] + weights[1] * cos[1] + weights[2] * meta, cos
    t = sum(weights)
    return (weights[0] / t) * cos[0] + (weights[1] / t) * cos[1] + (weights[2] / t) * cos[2], cos


def _exact(pkg, device, feats, weights, variant, k, ms):
    from tvbingefriend_recommendation_service_amd.ml.similarity_computer import SimilarityComputer

    sc = SimilarityComputer(*weights, device=device)
    res = sc.topk(feats, k=k, min_similarity=ms, variant=variant)
    return res, sc.last_topk_stats


@pytest.mark.parametrize("variant", ["populate", "computer"])
@pytest.mark.parametrize("k,ms", [(20, 0.1), (5, -1.0), (50, 0.05)])
def test_rerank_equals_float64_oracle(pkg, device, variant, k, ms):
    n = 1800
    feats = hybrid_like_blocks(n, seed=4242 + k, d_genre=24, d_text=300, as_features=True)
    weights = (0.4, 0.5, 0.1)
    res, st = _exact(pkg, device, feats, weights, variant, k, ms)
    h, cos = _oracle_hybrid(feats, weights, variant)
    assert res.score.dtype == np.float64 and res.comp.dtype == np.float64
    stats = {}
    check_topk_against_scores(h, 0, k, ms, res.idx, res.score, res.count, tol=REFINED_TOL, gap=REFINED_GAP,
                              what=f"rerank {variant} k={k}", stats=stats, f64=True, check_tie_order=False)
    # equal float64 scores of the result go by index
    for r in range(n):
        c = int(res.count[r])
        same = np.diff(res.score[r, :c]) == 0
        assert np.all(np.diff(res.idx[r, :c])[same] > 0)
    # counts are the oracle's exactly (no score of this catalogue lies within 1e-12 of the threshold
    # except the exact 0.1 of metadata-only pairs, which both sides compute exactly)
    hh = h.copy()
    np.fill_diagonal(hh, -np.inf)
    assert np.array_equal(res.count, np.minimum(k, (hh >= ms).sum(1)))
    # component scores of the kept pairs
    for r in range(0, n, 7):
        c = int(res.count[r])
        j = res.idx[r, :c]
        if variant == "populate":
            want = np.stack([cos[0][r, j], cos[1][r, j], (cos[2][r, j] + cos[3][r, j] + cos[4][r, j]) / 3], axis=1)
        else:
            want = np.stack([cos[0][r, j], cos[1][r, j], cos[2][r, j]], axis=1)
        assert np.abs(res.comp[r, :c] - want).max(initial=0) <= REFINED_TOL
        assert np.all(res.comp[r, c:] == 0) and np.all(res.score[r, c:] == 0) and np.all(res.idx[r, c:] == -1)
    print(f"rerank {variant} k={k}: {stats}, stages {st}")
    assert stats["max_err"] <= REFINED_TOL


def test_rerank_on_reference_generated_goldens(pkg, device):
    """The fixtures the real SimilarityComputer / FeatureExtractor produced in the build container:
    index-equal to the reference's own float64 results at every rank that is not a float64 tie."""
    import hashlib

    from parity import check_topk_against_golden

    # 100 shows through the real FeatureExtractor; the reference's full hybrid matrix is stored
    g = load_golden("plumbing_100")
    text = sp.csr_matrix((g["text_data"], g["text_indices"], g["text_indptr"]), shape=tuple(g["text_shape"]))
    feats = {"genre_features": g["genre_features"], "text_features": text, "platform_features": g["platform_features"],
             "type_features": g["type_features"], "language_features": g["language_features"]}
    for variant in ("computer", "populate"):  # one-hot metadata: both definitions give the same matrix
        res, st = _exact(pkg, device, feats, (0.4, 0.5, 0.1), variant, 20, 0.1)
        stats = {}
        check_topk_against_scores(g["hybrid_similarity"], 0, 20, 0.1, res.idx, res.score, res.count, tol=REFINED_TOL,
                                  gap=REFINED_GAP, what=f"plumbing_100 {variant}", stats=stats, f64=True,
                                  check_tie_order=False)
        print(f"plumbing_100 {variant}: {stats}, stages {st}")
    # 2048 hybrid-like shows: the reference's lists and its 21 best float64 scores per show are stored
    g = load_golden("hybrid_2048")
    n, seed, k, ms = int(g["n"]), int(g["seed"]), int(g["k"]), float(g["min_similarity"])
    feats = hybrid_like_blocks(n, seed, d_genre=24, d_text=200, as_features=True)
    assert hashlib.sha256(np.ascontiguousarray(np.asarray(feats["genre_features"])).tobytes()).hexdigest() == str(g["genre_sha256"])
    res, st = _exact(pkg, device, feats, tuple(g["weights"].tolist()), "computer", k, ms)
    blocks, weights = so.hybrid_blocks(feats, *g["weights"].tolist(), variant="computer")
    zo = so.build_z(blocks, weights)
    stats = {}
    check_topk_against_golden(g["idx"], g["sorted_scores"], k, res.idx, res.score, res.count,
                              pair_score=lambda r, cols: zo[r] @ zo[cols].T, min_similarity=ms, tol=REFINED_TOL,
                              gap=REFINED_GAP, what="hybrid_2048", stats=stats)
    print(f"hybrid_2048: {stats}, stages {st}")
    assert stats["exact"] >= 0.6 * stats["ranks"]  # narrow discrete blocks: a third of the ranks are exact float64 ties


def test_rerank_stages_on_tie_heavy_catalogue(pkg, device):
    """Groups of identical shows larger than any candidate list force the longer sweep (stage 2) and
    the scan of all shows (stage 3); the result is still the float64 oracle's, and equal scores go by
    index.  Also: near-ties that fp32 cannot order are ordered as float64 orders them."""
    n, k, ms = 2600, 20, 0.1
    feats = hybrid_like_blocks(n, seed=99, d_genre=16, d_text=120, as_features=True)
    genre, text = np.array(feats["genre_features"]), feats["text_features"].toarray()
    plat, typ, lang = (np.array(feats[q], dtype=np.float64) for q in ("platform_features", "type_features", "language_features"))
    # 300 identical shows (every pair among them ties at 1.0), 60 identical shows, 40 bare shows with equal metadata
    for lo, hi in ((100, 400), (900, 960)):
        genre[lo:hi], text[lo:hi] = genre[lo], text[lo]
        plat[lo:hi], typ[lo:hi], lang[lo:hi] = plat[lo], typ[lo], lang[lo]
    genre[1500:1540], text[1500:1540] = 0, 0.0
    plat[1500:1540], typ[1500:1540], lang[1500:1540] = plat[1500], typ[1500], lang[1500]
    # a near-tie below fp32 resolution: two shows that differ from show 2000's twin by 1e-9 in one text weight
    text[2001], text[2002] = text[2000], text[2000]
    nz = np.nonzero(text[2000])[0]
    if len(nz) >= 2:
        text[2001, nz[0]] *= 1 + 3e-8
        text[2002, nz[1]] *= 1 + 6e-8
    feats = {"genre_features": genre, "text_features": sp.csr_matrix(text), "platform_features": plat,
             "type_features": typ.astype(bool), "language_features": lang}
    from tvbingefriend_recommendation_service_amd import engine

    h, _ = _oracle_hybrid(feats, (0.4, 0.5, 0.1), "populate")
    results = []
    # the second stage with lists of 1024 (every show of the band a candidate: the 300-group fits a list)
    # and with lists of 64 (the large groups go on to the scan of all shows)
    for name, patch in (("band", {}), ("band, short lists", {"BAND_LIST_MAX": 64})):
        with pytest.MonkeyPatch.context() as mp:
            for attr, value in patch.items():
                mp.setattr(engine, attr, value)
            res, st = _exact(pkg, device, feats, (0.4, 0.5, 0.1), "populate", k, ms)
        assert st["resweep"] >= 300, (name, st)
        assert (st["scan"] == 0) if name == "band" else (st["scan"] >= 200), (name, st)
        stats = {}
        check_topk_against_scores(h, 0, k, ms, res.idx, res.score, res.count, tol=REFINED_TOL, gap=REFINED_GAP,
                                  what=f"tie-heavy ({name})", stats=stats, f64=True, check_tie_order=False)
        # members of the 300-group: 20 neighbours at exactly 1.0, the lowest indices of the group
        for i in (100, 250, 399):
            want = [j for j in range(100, 400) if j != i][:k]
            assert res.idx[i].tolist() == want and np.all(np.abs(res.score[i] - 1.0) <= 1e-15)
        print(f"tie-heavy ({name}): {stats}, stages {st}")
        results.append(res)
    for other in results[1:]:  # both ways: the same lists, scores and component scores, bit for bit
        for f in ("idx", "score", "count", "comp"):
            assert np.array_equal(getattr(results[0], f), getattr(other, f)), f


def test_input_validation_matches_sklearn(pkg, device):
    """sklearn's cosine_similarity rejects NaN and infinity (check_pairwise_arrays, reached from
    ml/similarity_computer.py:41,58,86); so does the scalable path."""
    from tvbingefriend_recommendation_service_amd.ml.similarity_computer import SimilarityComputer

    feats = hybrid_like_blocks(300, seed=5, d_genre=8, d_text=40, as_features=True)
    bad = dict(feats)
    t = feats["text_features"].toarray()
    t[17, 3] = np.nan
    bad["text_features"] = sp.csr_matrix(t)
    with pytest.raises(ValueError, match="Input contains NaN"):
        SimilarityComputer(device=device).topk(bad, k=5)
    bad = dict(feats)
    p = np.array(feats["platform_features"])
    p[5, 0] = np.inf
    bad["platform_features"] = p
    with pytest.raises(ValueError, match="Input contains infinity"):
        SimilarityComputer(device=device).topk(bad, k=5)
    with pytest.raises(ValueError, match="Input contains infinity"):
        SimilarityComputer(device=device).get_feature_statistics(bad)


def test_negative_weights_and_long_lists(pkg, device):
    """The reference accepts any weights whose sum is not zero (ml/similarity_computer.py:112-124)
    and any --top-n (scripts/populate_database.py:309-312).  A negative weight goes on the query
    side of a plain sweep; lists longer than the sweeps keep (K > 100) are selected from float64
    score rows."""
    n = 1300
    feats = hybrid_like_blocks(n, seed=77, d_genre=16, d_text=100, as_features=True)
    for weights, variant, k, ms in (((0.6, 0.7, -0.3), "computer", 10, 0.05), ((0.5, -0.2, 0.1), "populate", 20, -1.0),
                                    ((0.4, 0.5, 0.1), "populate", 150, 0.02)):
        res, st = _exact(pkg, device, feats, weights, variant, k, ms)
        h, _ = _oracle_hybrid(feats, weights, variant)
        stats = {}
        check_topk_against_scores(h, 0, k, ms, res.idx, res.score, res.count, tol=REFINED_TOL, gap=REFINED_GAP,
                                  what=f"weights {weights} k={k}", stats=stats, f64=True, check_tie_order=False)
        print(f"weights {weights} {variant} k={k}: {stats}, stages {st}")
        if k > 100:
            assert st["scan"] == n and st["candidates"] == 0


@pytest.mark.parametrize("n", [1, 2, 3, 17, 130])
def test_small_and_degenerate_catalogues(pkg, device, n):
    """Catalogues smaller than K and than a tile, shows with all-zero blocks (sklearn's norm guard: the
    block's cosines are 0), empty text rows, duplicated shows (exact ties at 1.0), and thresholds that
    keep everything or nothing: lists, counts and float64 scores are the oracle's."""
    feats = hybrid_like_blocks(max(n, 4), seed=77 + n, d_genre=12, d_text=40, as_features=True)
    feats = {name: a[:n] for name, a in feats.items()}
    genre = np.array(feats["genre_features"])
    text = sp.lil_matrix(feats["text_features"].astype(np.float64))
    if n >= 2:
        genre[0] = 0                      # a show without genres: its genre cosines are 0
        text[1] = 0                       # a show without text
    if n >= 17:
        genre[5], genre[6] = genre[4], genre[4]   # three shows equal in every block: ties at the top
        text[5], text[6] = text[4], text[4]
        for name in ("platform_features", "type_features", "language_features"):
            a = np.array(feats[name])
            a[5], a[6] = a[4], a[4]
            feats[name] = a
    feats["genre_features"], feats["text_features"] = genre, sp.csr_matrix(text)
    weights = (0.4, 0.5, 0.1)
    for variant in ("populate", "computer"):
        h, _ = _oracle_hybrid(feats, weights, variant)
        for k, ms in ((20, 0.1), (3, -1.0), (5, 0.999), (4, 1.5)):
            res, st = _exact(pkg, device, feats, weights, variant, k, ms)
            assert res.idx.shape == (n, k) and res.score.dtype == np.float64
            hh = h.copy()
            np.fill_diagonal(hh, -np.inf)
            near = np.abs(hh - ms) <= REFINED_TOL   # a pair this close to the threshold may fall either way
            lo = np.minimum(k, ((hh >= ms) & ~near).sum(1))
            hi = np.minimum(k, ((hh >= ms) | near).sum(1))
            assert np.all(res.count >= lo) and np.all(res.count <= hi), (n, variant, k, ms)
            check_topk_against_scores(h, 0, k, ms, res.idx, res.score, res.count, tol=REFINED_TOL, gap=REFINED_GAP,
                                      what=f"degenerate n={n} {variant} k={k} ms={ms}", f64=True, check_tie_order=False)
            for r in range(n):
                c = int(res.count[r])
                assert np.all(res.idx[r, :c] != r) and np.all(res.idx[r, c:] == -1) and np.all(res.score[r, c:] == 0)
                same = np.diff(res.score[r, :c]) == 0
                assert np.all(np.diff(res.idx[r, :c])[same] > 0)   # equal float64 scores go by index
            if ms > 1.0 + 1e-9:
                assert not res.count.any()


@pytest.mark.parametrize("weights", [(4.0, 5.0, 1.0), (6.0, -4.5, 0.3)])
def test_weights_far_from_a_unit_sum(pkg, device, weights):
    """The production loop applies the weights as given (scripts/populate_database.py:190-192): scores --
    and the fp32 error of the sweep's scores -- grow with the weights' magnitudes, and so does the
    allowance the re-rank certifies with.  Lists are still the float64 oracle's."""
    n, k = 1500, 20
    feats = hybrid_like_blocks(n, seed=515, d_genre=24, d_text=300, as_features=True)
    h, _ = _oracle_hybrid(feats, weights, "populate")
    ms = float(np.quantile(h, 0.9))
    res, st = _exact(pkg, device, feats, weights, "populate", k, ms)
    stats = {}
    check_topk_against_scores(h, 0, k, ms, res.idx, res.score, res.count, tol=10 * REFINED_TOL, gap=10 * REFINED_GAP,
                              what=f"weights {weights}", stats=stats, f64=True, check_tie_order=False)
    print(f"weights {weights}: {stats}, stages {st}")
    assert stats["exact"] > 0
