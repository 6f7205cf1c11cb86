"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run once in the build container (the reference tree is not available on the
GPU box, only the .npz files travel):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python /root/repo/tests/golden/generate_golden.py

It imports ``SimilarityComputer`` (ml/similarity_computer.py:12) and
``FeatureExtractor`` (ml/feature_extractor.py:17) from the reference, feeds
them the inputs described below and stores inputs (or the seed that rebuilds
them, plus a checksum) together with the reference's outputs.  Top-K index
sets are derived from the reference's own matrices with the reference's
selection loop (``np.argsort(row)[::-1]`` + skip self + threshold + K cut,
scripts/populate_database.py:195-218).
"""

from __future__ import annotations

import hashlib
import os
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import scipy.sparse as sp

from tvbingefriend_recommendation_service.ml.feature_extractor import FeatureExtractor
from tvbingefriend_recommendation_service.ml.similarity_computer import SimilarityComputer

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from synth import dense_gaussian, hybrid_like_blocks, synthetic_shows  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_walk(h_row, self_idx, top_n, min_similarity):
    """scripts/populate_database.py:195-218 applied to one row of the reference's matrix."""
    kept = []
    for j in np.argsort(h_row)[::-1]:
        if j == self_idx:
            continue
        if float(h_row[j]) < min_similarity:
            break
        if len(kept) >= top_n:
            break
        kept.append(int(j))
    return kept


def ref_topk(h, k, min_similarity):
    n = h.shape[0]
    idx = np.full((n, k), -1, np.int32)
    sc = np.zeros((n, k), np.float64)
    cnt = np.zeros(n, np.int32)
    for i in range(n):
        kept = ref_walk(h[i], i, k, min_similarity)
        cnt[i] = len(kept)
        idx[i, : len(kept)] = kept
        sc[i, : len(kept)] = h[i, kept]
    return idx, sc, cnt


def sorted_scores(h, k):
    """The k+1 largest scores of every row (self excluded), descending: lets a
    checker see every gap between consecutive ranks, the K boundary included."""
    hh = h.copy()
    np.fill_diagonal(hh, -np.inf)
    return -np.sort(-hh, axis=1)[:, : k + 1]


def conftest_trio():
    """The hand-written 3-show fixtures of the reference's tests/conftest.py:137-200."""
    genre = np.array([[1, 1, 1, 0, 0], [1, 1, 0, 0, 0], [0, 0, 0, 1, 0]], dtype=float)
    data = np.array([0.5, 0.3, 0.7, 0.4, 0.6, 0.8])
    row = np.array([0, 0, 1, 1, 2, 2])
    col = np.array([0, 3, 1, 4, 2, 5])
    text = sp.csr_matrix((data, (row, col)), shape=(3, 10))
    platform = np.array([[1, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=float)
    typ = np.array([[1, 0], [1, 0], [1, 0]], dtype=float)
    lang = np.array([[1, 0], [1, 0], [1, 0]], dtype=float)
    feats = dict(genre_features=genre, text_features=text, platform_features=platform,
                 type_features=typ, language_features=lang)
    out = SimilarityComputer().compute_all_similarities(feats)
    np.savez_compressed(
        HERE / "conftest_trio.npz",
        genre_features=genre, text_features=text.toarray(), platform_features=platform,
        type_features=typ, language_features=lang, **out,
    )


def edge_frame():
    """Four shows through the real FeatureExtractor: no genres / empty summary /
    NaN type / None platform each appear once."""
    df = pd.DataFrame(
        {
            "id": [11, 12, 13, 14],
            "genres": [["Drama", "Crime"], ["Drama"], [], ["Comedy", "Drama"]],
            "summary_clean": [
                "a chemistry teacher turns to crime with a former student",
                "a criminal lawyer and a former student of crime",
                "",
                "office workers and a teacher share a comedy of crime",
            ],
            "type": ["Scripted", "Scripted", np.nan, "Reality"],
            "language": ["English", "English", "Spanish", "English"],
            "platform": ["AMC", "AMC", None, "NBC"],
        }
    )
    fx = FeatureExtractor(max_text_features=50, text_min_df=1, text_max_df=1.0)
    f = fx.extract_all_features(df)
    feats = {k: f[k] for k in ("genre_features", "text_features", "platform_features",
                              "type_features", "language_features")}
    out = SimilarityComputer().compute_all_similarities(feats)
    np.savez_compressed(
        HERE / "edge_frame.npz",
        ids=df["id"].to_numpy(),
        genre_features=np.asarray(feats["genre_features"]),
        text_features=feats["text_features"].toarray(),
        platform_features=np.asarray(feats["platform_features"]),
        type_features=np.asarray(feats["type_features"]),
        language_features=np.asarray(feats["language_features"]),
        **out,
    )


def dense_tie_free(name, n, d, seed, ks=(20, 50)):
    """Continuous random fp32 features, single block: almost every gap between
    consecutive ranks is far above fp32 rounding, so exact index parity is well
    defined on nearly all rows; the stored sorted scores mark the rest."""
    x = dense_gaussian(n, d, seed)
    sc = SimilarityComputer()
    h = sc.compute_genre_similarity(x.astype(np.float64))  # the f64 path of the reference
    save = {"n": n, "d": d, "seed": seed, "x_sha256": sha(x)}
    for k in ks:
        idx, s, cnt = ref_topk(h, k, -1.0)
        save[f"idx_k{k}"] = idx
        save[f"score_k{k}"] = s
        save[f"count_k{k}"] = cnt
        save[f"sorted_scores_k{k}"] = sorted_scores(h, k)
        save[f"clear_k{k}"] = clear_mask(save[f"sorted_scores_k{k}"], k)
    save["stats_mean_std_min_max_median"] = np.array(
        [sc.get_similarity_statistics(h)[q] for q in ("mean", "std", "min", "max", "median")])
    np.savez_compressed(HERE / f"{name}.npz", **save)


GAP_CLEAR = 2e-6  # ranks whose score gaps to both neighbours exceed this are held to index equality


def clear_mask(srt, k):
    """Per rank 1..k: True where the gaps to the previous and to the next rank (rank k+1 included)
    both exceed GAP_CLEAR in the reference's float64 scores, so that fp32 rounding (about 3e-7)
    cannot reorder them.  ``srt`` are the k+1 largest scores of every row."""
    gaps = srt[:, :-1] - srt[:, 1:]  # gap between rank r and r+1, r = 1..k
    ok = gaps > GAP_CLEAR
    prev_ok = np.concatenate([np.ones((srt.shape[0], 1), bool), ok[:, :-1]], axis=1)
    return prev_ok & ok


def dense_verified_gaps(name, n, d, k, first_seed, min_gap=1e-5, tries=5000):
    """A fixture on which "top-K indices exactly" is well defined for EVERY rank: the seed is
    searched until the smallest gap between consecutive float64 scores over ranks 1..k+1 of all rows
    exceeds ``min_gap`` (verified here, stored with the fixture)."""
    sc = SimilarityComputer()
    for seed in range(first_seed, first_seed + tries):
        x = dense_gaussian(n, d, seed)
        xn = x.astype(np.float64)
        xn /= np.linalg.norm(xn, axis=1, keepdims=True)
        srt = sorted_scores(xn @ xn.T, k)  # cheap pre-check before the reference is asked
        if (srt[:, :-1] - srt[:, 1:]).min() <= min_gap * 1.01:
            continue
        h = sc.compute_genre_similarity(x.astype(np.float64))
        srt = sorted_scores(h, k)
        smallest = float((srt[:, :-1] - srt[:, 1:]).min())
        if smallest <= min_gap:
            continue
        idx, s, cnt = ref_topk(h, k, -1.0)
        np.savez_compressed(HERE / f"{name}.npz", n=n, d=d, seed=seed, k=k, x_sha256=sha(x), idx=idx, score=s,
                            count=cnt, sorted_scores=srt, min_rank_gap=smallest)
        print(f"{name}: seed {seed}, smallest gap over ranks 1..{k + 1}: {smallest:.3e}")
        return
    raise RuntimeError("no seed with the required gap found")


def hybrid_ties(name, n, seed, k=20, min_similarity=0.1, weights=(0.4, 0.5, 0.1)):
    """Block-structured features with many exact ties, through the real SimilarityComputer."""
    feats = hybrid_like_blocks(n, seed, d_genre=24, d_text=200, as_features=True)
    sc = SimilarityComputer(*weights)
    out = sc.compute_all_similarities(feats)
    h = out["hybrid_similarity"]
    idx, s, cnt = ref_topk(h, k, min_similarity)
    comp = np.zeros((n, k, 3))
    for i in range(n):
        j = idx[i, : cnt[i]]
        comp[i, : cnt[i], 0] = out["genre_similarity"][i, j]
        comp[i, : cnt[i], 1] = out["text_similarity"][i, j]
        comp[i, : cnt[i], 2] = out["metadata_similarity"][i, j]
    srt = sorted_scores(h, k)
    stats = {q: np.array([sc.get_similarity_statistics(out[m])[q] for m in
                          ("genre_similarity", "text_similarity", "metadata_similarity",
                           "hybrid_similarity")]) for q in ("mean", "std", "min", "max", "median")}
    np.savez_compressed(
        HERE / f"{name}.npz", n=n, seed=seed, k=k, min_similarity=min_similarity,
        weights=np.array(weights), idx=idx, score=s, count=cnt, comp=comp,
        sorted_scores=srt,
        genre_sha256=sha(np.asarray(feats["genre_features"])),
        **{f"stats_{q}": v for q, v in stats.items()},
    )


def plumbing_100():
    """100 synthetic shows -> real FeatureExtractor -> real SimilarityComputer ->
    reference selection loop (top-20, min 0.1): the counterpart of the reference's
    `run_pipeline.py --max-shows 100` smoke configuration."""
    df = synthetic_shows(100, seed=7)
    fx = FeatureExtractor()
    f = fx.extract_all_features(df)
    feats = {k: f[k] for k in ("genre_features", "text_features", "platform_features",
                              "type_features", "language_features")}
    sc = SimilarityComputer()
    out = sc.compute_all_similarities(feats)
    idx, s, cnt = ref_topk(out["hybrid_similarity"], 20, 0.1)
    comp = np.zeros((100, 20, 3))
    for i in range(100):
        j = idx[i, : cnt[i]]
        comp[i, : cnt[i], 0] = out["genre_similarity"][i, j]
        comp[i, : cnt[i], 1] = out["text_similarity"][i, j]
        comp[i, : cnt[i], 2] = out["metadata_similarity"][i, j]
    np.savez_compressed(
        HERE / "plumbing_100.npz",
        ids=df["id"].to_numpy(),
        genre_features=np.asarray(feats["genre_features"]),
        text_data=feats["text_features"].data, text_indices=feats["text_features"].indices,
        text_indptr=feats["text_features"].indptr, text_shape=np.array(feats["text_features"].shape),
        platform_features=np.asarray(feats["platform_features"]),
        type_features=np.asarray(feats["type_features"]),
        language_features=np.asarray(feats["language_features"]),
        idx=idx, score=s, count=cnt, comp=comp,
        hybrid_similarity=out["hybrid_similarity"],
    )


def main():
    os.chdir("/tmp")
    conftest_trio()
    edge_frame()
    dense_tie_free("dense_512x64", 512, 64, seed=41001)
    dense_tie_free("dense_2048x512", 2048, 512, seed=41002)
    dense_verified_gaps("dense_tiefree_256x64", 256, 64, k=20, first_seed=42000)
    hybrid_ties("hybrid_2048", 2048, seed=41003)
    plumbing_100()
    for p in sorted(HERE.glob("*.npz")):
        print(p.name, p.stat().st_size)


if __name__ == "__main__":
    main()
