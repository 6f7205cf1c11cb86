"""Seeded synthetic inputs shared by the golden generator and the tests (NumPy only)."""

from __future__ import annotations

import numpy as np


def dense_gaussian(n: int, d: int, seed: int) -> np.ndarray:
    """Dense fp32 standard-normal features: one block, no ties."""
    return np.random.default_rng(seed).standard_normal((n, d), dtype=np.float32)


def hybrid_like_blocks(n: int, seed: int, d_genre: int = 32, d_text: int = 960,
                       text_density: float = 0.03, as_features: bool = False):
    """Block-structured features shaped like the reference's feature files:
    genre multi-hot (Bernoulli 0.08), non-negative sparse text rows with unit L2
    norm (TF-IDF-like), platform one-hot(21), type one-hot(5), language one-hot(6).
    A few rows are given no genres / no text, as real catalogues have.
    """
    rng = np.random.default_rng(seed)
    genre = (rng.random((n, d_genre)) < 0.08).astype(np.float64)
    mask = rng.random((n, d_text)) < text_density
    text = np.abs(rng.standard_normal((n, d_text))) * mask
    empty = rng.random(n) < 0.02
    text[empty] = 0.0
    genre[rng.random(n) < 0.02] = 0.0
    nrm = np.sqrt((text * text).sum(axis=1))
    nrm[nrm == 0] = 1.0
    text /= nrm[:, None]

    def one_hot(width, p=None):
        a = np.zeros((n, width))
        a[np.arange(n), rng.choice(width, size=n, p=p)] = 1.0
        return a

    pp = np.arange(21, 0, -1, dtype=np.float64)
    platform = one_hot(21, pp / pp.sum())
    typ = one_hot(5, np.array([0.6, 0.2, 0.1, 0.05, 0.05]))
    lang = one_hot(6, np.array([0.7, 0.1, 0.08, 0.06, 0.04, 0.02]))
    if as_features:
        import scipy.sparse as sp

        return {
            "genre_features": genre.astype(np.int64),
            "text_features": sp.csr_matrix(text),
            "platform_features": platform,
            "type_features": typ.astype(bool),
            "language_features": lang,
        }
    return genre, text, platform, typ, lang


_WORDS = (
    "crime drama family detective murder city love war doctor hospital lawyer school teacher "
    "student police agent secret space alien future past king queen dragon magic comedy office "
    "friends neighbours cooking travel music dance competition reality island survival money "
    "business politics president journalist mystery island ghost vampire zombie hero villain "
    "robot science history documentary nature animals ocean mountain racing football"
).split()


def synthetic_shows(n: int, seed: int):
    """A small synthetic catalogue with the columns FeatureExtractor.extract_all_features
    reads (ml/feature_extractor.py:200-229 of the reference)."""
    import pandas as pd

    rng = np.random.default_rng(seed)
    genres_all = ["Drama", "Comedy", "Crime", "Thriller", "Action", "Romance", "Science-Fiction",
                  "Fantasy", "Mystery", "Family", "Horror", "Adventure"]
    platforms = ["NBC", "CBS", "ABC", "HBO", "Netflix", "BBC One", "FOX", "AMC", None]
    types = ["Scripted", "Reality", "Animation", "Documentary", "Talk Show"]
    langs = ["English", "Japanese", "Spanish", "German", "French", "Korean", "Italian"]
    rows = []
    for i in range(n):
        g = list(rng.choice(genres_all, size=int(rng.integers(0, 4)), replace=False))
        words = rng.choice(_WORDS, size=int(rng.integers(0, 25)))
        rows.append(
            {
                "id": 1000 + 3 * i,
                "name": f"Show {i}",
                "genres": g,
                "summary_clean": " ".join(words),
                "type": types[int(rng.choice(len(types), p=[0.6, 0.15, 0.1, 0.1, 0.05]))],
                "language": langs[int(rng.choice(len(langs), p=[0.6, 0.1, 0.08, 0.07, 0.06, 0.05, 0.04]))],
                "platform": platforms[int(rng.integers(0, len(platforms)))],
                "rating_avg": float(rng.uniform(4, 9.5)),
            }
        )
    return pd.DataFrame(rows)


def write_feature_files(path, n: int, seed: int = 7, d_text: int = 500) -> dict:
    """A synthetic catalogue in the reference's on-disk form (scripts/compute_features.py:115-129:
    genre multi-hot int64, TF-IDF CSR float64 with unit rows, platform one-hot of 21 float64, type
    dummies bool, language one-hot of 6 float64) plus shows_metadata.csv; returns the arrays."""
    import scipy.sparse as sp

    rng = np.random.default_rng(seed)
    genre = (rng.random((n, 24)) < 0.1).astype(np.int64)
    nnz_per_row = rng.integers(8, 48, size=n)
    nnz_per_row[rng.random(n) < 0.02] = 0  # shows without a summary: all-zero TF-IDF rows
    indptr = np.concatenate([[0], np.cumsum(nnz_per_row)])
    # distinct sorted columns per row: the smallest keys of a random key per (row, column)
    order = np.argsort(rng.random((n, d_text)), axis=1)[:, :48] if n else np.empty((0, 48), int)
    keep = np.arange(48)[None, :] < nnz_per_row[:, None]
    cols = np.sort(np.where(keep, order, d_text), axis=1)[keep] if n else np.empty(0, int)
    vals = np.abs(rng.standard_normal(indptr[-1])) + 0.05
    text = sp.csr_matrix((vals, cols, indptr), shape=(n, d_text), dtype=np.float64)
    norms = np.sqrt(np.asarray(text.multiply(text).sum(axis=1)).ravel())
    norms[norms == 0] = 1.0
    text = sp.csr_matrix(sp.diags(1.0 / norms) @ text)

    def one_hot(width, dtype, p=None):
        out = np.zeros((n, width), dtype=dtype)
        out[np.arange(n), rng.choice(width, size=n, p=p)] = 1
        return out

    p_platform = np.r_[np.full(20, 0.7 / 20), 0.3]
    feats = {"genre_features": genre, "text_features": text,
             "platform_features": one_hot(21, np.float64, p_platform),
             "type_features": one_hot(8, bool), "language_features": one_hot(6, np.float64, [0.6, 0.1, 0.1, 0.08, 0.07, 0.05])}
    for key, value in feats.items():
        if sp.issparse(value):
            sp.save_npz(path / f"{key}.npz", value, compressed=False)
        else:
            np.save(path / f"{key}.npy", value)
    ids = np.arange(1, n + 1) * 3 + 11
    (path / "shows_metadata.csv").write_text("id,name\n" + "\n".join(f"{i},show {i}" for i in ids) + "\n")
    return feats
