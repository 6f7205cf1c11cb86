"""Feature blocks -> one row-major fp32 matrix, laid out for the kernels.

The reference hands five arrays around (ml/feature_extractor.py:244-252 and
the files of scripts/compute_features.py:115-129): genre multi-hot (int64),
TF-IDF text (CSR float64), platform / type / language one-hots (float64 /
bool / float64).  The similarity path only ever uses them through cosine
similarities that are then mixed with three weights, so they are packed as

    X = [ genre | text | metadata... ]   (n, d) fp32

with every block starting on a multiple of 4 columns (16-byte aligned rows
and block starts for the kernels; the few padding columns are zero and change
neither norms nor dot products).  After block-wise normalisation and scaling
by sqrt(weight) the hybrid similarity is a single matrix product Z @ Z.T.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

BLOCK_ALIGN = 4  # floats

FEATURE_KEYS = ("genre_features", "text_features", "platform_features", "type_features",
                "language_features")


def to_dense_f32(x) -> np.ndarray:
    """Dense C-contiguous fp32 copy of a dense array or SciPy sparse matrix."""
    if hasattr(x, "toarray") and hasattr(x, "tocsr"):  # scipy.sparse without importing it
        x = x.toarray()
    x = np.asarray(x)
    if x.ndim != 2:
        raise ValueError(f"feature blocks must be 2-D, got shape {x.shape}")
    return np.ascontiguousarray(x, dtype=np.float32)


@dataclass
class BlockLayout:
    """Column layout of the packed matrix.

    offsets  nb+1 ascending multiples of 4; block b occupies [offsets[b], offsets[b+1])
    widths   real (unpadded) width of each block
    weights  hybrid weight of each block (scale = sqrt(weight))
    groups   for each block, which reported component it belongs to
             (0 = genre_score, 1 = text_score, 2 = metadata_score)
    """

    offsets: list[int]
    widths: list[int]
    weights: list[float]
    groups: list[int] = field(default_factory=list)

    @property
    def d(self) -> int:
        return self.offsets[-1]

    @property
    def nb(self) -> int:
        return len(self.weights)

    def scales(self) -> list[float]:
        for w in self.weights:
            if w < 0 or not np.isfinite(w):
                raise ValueError(
                    f"similarity weights must be finite and >= 0 (got {w}): the weighted "
                    "feature matrix needs sqrt(weight)")
        return [float(np.sqrt(w)) for w in self.weights]

    def component_scales(self) -> list[float]:
        """Scales for the per-block score kernel: a block with weight 0 is given
        scale 1 there so that its plain cosine can still be reported."""
        return [s if s > 0 else 1.0 for s in self.scales()]


def pack_blocks(blocks, weights, groups=None) -> tuple[np.ndarray, BlockLayout]:
    """Concatenate dense blocks column-wise with 4-float aligned block starts."""
    dense = [to_dense_f32(b) for b in blocks]
    n = dense[0].shape[0]
    if any(b.shape[0] != n for b in dense):
        raise ValueError("all feature blocks must have the same number of rows")
    offsets = [0]
    for b in dense:
        w = b.shape[1]
        offsets.append(offsets[-1] + (w + BLOCK_ALIGN - 1) // BLOCK_ALIGN * BLOCK_ALIGN)
    x = np.zeros((n, max(offsets[-1], BLOCK_ALIGN)), dtype=np.float32)
    if offsets[-1] == 0:  # all blocks empty: keep one zero column group so shapes stay valid
        offsets[-1] = BLOCK_ALIGN
    for b, o in zip(dense, offsets):
        x[:, o : o + b.shape[1]] = b
    layout = BlockLayout(offsets, [b.shape[1] for b in dense], [float(w) for w in weights],
                         list(groups) if groups is not None else list(range(len(dense))))
    return x, layout


def pack_features(features: dict, genre_weight: float, text_weight: float, metadata_weight: float,
                  variant: str = "computer") -> tuple[np.ndarray, BlockLayout]:
    """Pack the reference's five feature arrays.

    variant "computer" -- SimilarityComputer semantics (ml/similarity_computer.py):
        metadata = cosine of hstack(platform, type, language) (:84-86); weights
        divided by their sum (:112-115).
    variant "populate" -- the production row loop (scripts/populate_database.py):
        metadata = mean of the three cosines (:184-187) and raw weights (:190-192);
        platform / type / language become three blocks weighted metadata_weight / 3.
    """
    missing = [k for k in FEATURE_KEYS if k not in features]
    if missing:
        raise KeyError(missing[0])
    g = to_dense_f32(features["genre_features"])
    t = to_dense_f32(features["text_features"])
    p = to_dense_f32(features["platform_features"])
    ty = to_dense_f32(features["type_features"])
    la = to_dense_f32(features["language_features"])
    if variant == "computer":
        total = genre_weight + text_weight + metadata_weight
        ws = [genre_weight / total, text_weight / total, metadata_weight / total]
        return pack_blocks([g, t, np.hstack([p, ty, la])], ws, [0, 1, 2])
    if variant == "populate":
        ws = [genre_weight, text_weight, metadata_weight / 3, metadata_weight / 3, metadata_weight / 3]
        return pack_blocks([g, t, p, ty, la], ws, [0, 1, 2, 2, 2])
    raise ValueError(f"unknown variant {variant!r}")


def _is_sparse(x) -> bool:
    return hasattr(x, "toarray") and hasattr(x, "tocsr")


def block_sources(features: dict, genre_weight: float, text_weight: float, metadata_weight: float,
                  variant: str = "computer"):
    """The five feature arrays in their stored form plus the column each one starts at.

    Same layout as :func:`pack_features` (so both routes produce the same Z), but nothing is
    densified or converted on the host: a SciPy sparse block stays a CSR triple, dense blocks
    keep their dtype (int64 genre, bool type, float64 one-hots; anything else becomes fp32).
    Returns (sources, layout); a source is ("dense", array, z_col) or
    ("csr", indptr, indices, data, width, z_col).
    """
    missing = [k for k in FEATURE_KEYS if k not in features]
    if missing:
        raise KeyError(missing[0])
    names = ("genre_features", "text_features", "platform_features", "type_features", "language_features")
    arrays = []
    for name in names:
        a = features[name]
        if _is_sparse(a):
            a = a.tocsr()
            if not a.has_canonical_format:
                a = a.copy()
                a.sum_duplicates()
        else:
            a = np.asarray(a)
            if a.ndim != 2:
                raise ValueError(f"feature blocks must be 2-D, got shape {a.shape}")
            if a.dtype not in (np.float32, np.float64, np.int64, np.bool_, np.uint8):
                a = a.astype(np.float32)
            a = np.ascontiguousarray(a)
        arrays.append(a)
    n = arrays[0].shape[0]
    if any(a.shape[0] != n for a in arrays):
        raise ValueError("all feature blocks must have the same number of rows")
    widths = [a.shape[1] for a in arrays]
    pad = lambda w: (w + BLOCK_ALIGN - 1) // BLOCK_ALIGN * BLOCK_ALIGN  # noqa: E731
    if variant == "computer":
        total = genre_weight + text_weight + metadata_weight
        weights = [genre_weight / total, text_weight / total, metadata_weight / total]
        meta_w = widths[2] + widths[3] + widths[4]
        offsets = [0, pad(widths[0])]
        offsets.append(offsets[-1] + pad(widths[1]))
        offsets.append(offsets[-1] + pad(meta_w))
        z_cols = [0, offsets[1], offsets[2], offsets[2] + widths[2], offsets[2] + widths[2] + widths[3]]
        layout = BlockLayout(offsets, [widths[0], widths[1], meta_w], weights, [0, 1, 2])
    elif variant == "populate":
        weights = [genre_weight, text_weight, metadata_weight / 3, metadata_weight / 3, metadata_weight / 3]
        offsets = [0]
        for w in widths:
            offsets.append(offsets[-1] + pad(w))
        z_cols = offsets[:-1]
        layout = BlockLayout(offsets, widths, weights, [0, 1, 2, 2, 2])
    else:
        raise ValueError(f"unknown variant {variant!r}")
    if layout.offsets[-1] == 0:
        layout.offsets[-1] = BLOCK_ALIGN
    sources = []
    for a, zc in zip(arrays, z_cols):
        if _is_sparse(a):
            sources.append(("csr", a.indptr, a.indices, a.data, a.shape[1], zc))
        else:
            sources.append(("dense", a, zc))
    return sources, layout


def fold_components(comp: np.ndarray, layout: BlockLayout) -> np.ndarray:
    """(nq, k, nb) per-block cosines -> (nq, k, 3) genre / text / metadata scores
    (metadata = mean of its blocks, scripts/populate_database.py:187)."""
    out = np.zeros(comp.shape[:2] + (3,), dtype=np.float64)
    cnt = np.zeros(3)
    for b, g in enumerate(layout.groups):
        out[..., g] += comp[..., b]
        cnt[g] += 1
    cnt[cnt == 0] = 1
    return out / cnt
