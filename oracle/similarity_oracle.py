"""CPU oracle for the batch similarity + top-K hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, in plain NumPy/SciPy, the arithmetic of the reference's
hot path so that the HIP kernels can be checked against it.  It is never
imported by the product package: only tests, the smoke check and the
``cpu_baseline`` leg of the benchmark may use it.

Parity status: PINNED.  ``tests/golden/generate_golden.py`` ran the real
reference modules (``SimilarityComputer``, ``FeatureExtractor``) in the build
container and stored their outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks every function below against those
vectors and against the known answers of the reference's own unit tests
(tests/test_ml/test_similarity_computer.py).

Third-party arithmetic restated here: ``sklearn.metrics.pairwise.
cosine_similarity`` of scikit-learn 1.7.2 (pinned in the reference's
poetry.lock:5998-5999; call sites ml/similarity_computer.py:41,58,86 and
scripts/populate_database.py:180-186):

* dtype: float32 only if both inputs are float32, else float64
  (sklearn/metrics/pairwise.py:51-72);
* rows are L2-normalised on a copy; dense norms smaller than 10*eps of the
  dtype are replaced by 1.0 (sklearn/preprocessing/_data.py:118-123,
  2010-2015), sparse rows are divided only when their sum of squares is
  non-zero (inplace_csr_row_normalize_l2);
* the result is the dense product Xn @ Yn.T (pairwise.py:1736).
"""

from __future__ import annotations

import numpy as np

try:  # SciPy is only needed for CSR text features.
    import scipy.sparse as sp
except Exception:  # pragma: no cover - scipy is present in the build image
    sp = None


# --------------------------------------------------------------------------
# sklearn.metrics.pairwise.cosine_similarity, restated
# --------------------------------------------------------------------------

def _pair_dtype(x, y):
    """sklearn/metrics/pairwise.py:51-72 -- float32 iff both are float32."""
    xd = getattr(x, "dtype", None)
    yd = getattr(y, "dtype", None)
    if xd == np.float32 and yd == np.float32:
        return np.float32
    return np.float64


def l2_normalize_rows(x, dtype=None):
    """Row L2 normalisation with sklearn's zero guard.

    Dense: norms < 10*eps(dtype) become 1.0 (preprocessing/_data.py:118-123).
    CSR: a row is scaled only if its sum of squares is non-zero.
    Returns a new array / matrix of ``dtype``.
    """
    if sp is not None and sp.issparse(x):
        dtype = dtype or (np.float32 if x.dtype == np.float32 else np.float64)
        x = sp.csr_matrix(x, dtype=dtype, copy=True)
        sq = np.asarray(x.multiply(x).sum(axis=1)).ravel()
        norms = np.sqrt(sq)
        norms[sq == 0.0] = 1.0
        inv = sp.diags(1.0 / norms)
        return (inv @ x).tocsr()
    x = np.asarray(x)
    dtype = dtype or (np.float32 if x.dtype == np.float32 else np.float64)
    x = np.array(x, dtype=dtype, copy=True)
    if x.ndim != 2:
        raise ValueError("expected a 2-D array")
    norms = np.sqrt(np.einsum("ij,ij->i", x, x))
    norms[norms < 10 * np.finfo(dtype).eps] = 1.0
    x /= norms[:, None]
    return x


def cosine_similarity(x, y=None):
    """Dense cosine similarity matrix, sklearn semantics (pairwise.py:1728-1738)."""
    same = y is None or y is x
    if y is None:
        y = x
    dtype = _pair_dtype(x, y)
    xn = l2_normalize_rows(x, dtype)
    yn = xn if same else l2_normalize_rows(y, dtype)
    out = xn @ yn.T
    if sp is not None and sp.issparse(out):
        out = out.toarray()
    return np.asarray(out)


# --------------------------------------------------------------------------
# ml/similarity_computer.py, restated
# --------------------------------------------------------------------------

class SimilarityComputerOracle:
    """Mirror of the reference ``SimilarityComputer`` (ml/similarity_computer.py:12)."""

    def __init__(self, genre_weight=0.4, text_weight=0.5, metadata_weight=0.1):
        # ml/similarity_computer.py:26-28
        self.genre_weight = genre_weight
        self.text_weight = text_weight
        self.metadata_weight = metadata_weight

    def compute_genre_similarity(self, genre_features):
        # ml/similarity_computer.py:41
        return cosine_similarity(genre_features)

    def compute_text_similarity(self, text_features):
        # ml/similarity_computer.py:58 (dense or CSR input)
        return cosine_similarity(text_features)

    def compute_metadata_similarity(self, platform_features, type_features, language_features):
        # ml/similarity_computer.py:84-86 -- cosine of the concatenation
        meta = np.hstack([platform_features, type_features, language_features])
        return cosine_similarity(meta)

    def normalized_weights(self):
        # ml/similarity_computer.py:112-115
        total = self.genre_weight + self.text_weight + self.metadata_weight
        return (self.genre_weight / total, self.text_weight / total, self.metadata_weight / total)

    def compute_hybrid_similarity(self, genre_similarity, text_similarity, metadata_similarity):
        # ml/similarity_computer.py:112-124
        gw, tw, mw = self.normalized_weights()
        return gw * genre_similarity + tw * text_similarity + mw * metadata_similarity

    def compute_all_similarities(self, features):
        # ml/similarity_computer.py:147-169
        g = self.compute_genre_similarity(features["genre_features"])
        t = self.compute_text_similarity(features["text_features"])
        m = self.compute_metadata_similarity(
            features["platform_features"], features["type_features"], features["language_features"]
        )
        h = self.compute_hybrid_similarity(g, t, m)
        return {
            "genre_similarity": g,
            "text_similarity": t,
            "metadata_similarity": m,
            "hybrid_similarity": h,
        }

    def get_similarity_statistics(self, similarity_matrix):
        # ml/similarity_computer.py:182-190 -- strict upper triangle, ddof=0
        m = np.asarray(similarity_matrix)
        ut = m[np.triu_indices_from(m, k=1)]
        return {
            "mean": float(ut.mean()),
            "std": float(ut.std()),
            "min": float(ut.min()),
            "max": float(ut.max()),
            "median": float(np.median(ut)),
        }


# --------------------------------------------------------------------------
# Top-K selection
# --------------------------------------------------------------------------

def reference_order(scores):
    """Literal restatement of ``np.argsort(scores)[::-1]``
    (scripts/populate_database.py:195, services/content_based_service.py:209).
    The order inside groups of equal scores is whatever NumPy's unstable sort
    produces and is not portable."""
    return np.argsort(scores)[::-1]


def deterministic_order(scores):
    """Total order used for comparison: score descending, then index ascending."""
    scores = np.asarray(scores)
    return np.lexsort((np.arange(scores.shape[0]), -scores))


def walk_row(order, scores, self_idx, top_n, min_similarity, break_n_after_append=False):
    """The selection walk shared by both reference top-K loops.

    scripts/populate_database.py:198-208 tests ``score < min`` then ``len >= n``
    before appending; services/content_based_service.py:213-234 tests
    ``score < min``, appends, then tests ``len >= n``.  Both keep the same
    elements; the flag only mirrors the order of the two tests.
    Returns the list of kept column indices.
    """
    kept = []
    for j in order:
        j = int(j)
        if j == self_idx:
            continue
        if float(scores[j]) < min_similarity:
            break
        if not break_n_after_append and len(kept) >= top_n:
            break
        kept.append(j)
        if break_n_after_append and len(kept) >= top_n:
            break
    return kept


def topk_rows(hybrid_rows, row0, k, min_similarity, exclude_self=True, order_fn=deterministic_order):
    """Top-K of each row of ``hybrid_rows`` (rows are global rows row0, row0+1, ...).

    Returns (idx[nq,k] int32 padded with -1, score[nq,k] float64 padded with 0,
    count[nq] int32).
    """
    hybrid_rows = np.asarray(hybrid_rows)
    nq = hybrid_rows.shape[0]
    idx = np.full((nq, k), -1, dtype=np.int32)
    score = np.zeros((nq, k), dtype=np.float64)
    count = np.zeros(nq, dtype=np.int32)
    for r in range(nq):
        h = hybrid_rows[r]
        kept = walk_row(order_fn(h), h, row0 + r if exclude_self else -1, k, min_similarity)
        count[r] = len(kept)
        idx[r, : len(kept)] = kept
        score[r, : len(kept)] = h[kept]
    return idx, score, count


def topk_fast(hybrid_rows, row0, k, min_similarity, exclude_self=True):
    """Vectorised equivalent of ``topk_rows`` with the deterministic order
    (one partition per row, then a small sort of the survivors); used on large
    inputs and for the CPU baseline."""
    h = np.array(hybrid_rows, dtype=np.float64, copy=True)
    nq, n = h.shape
    if exclude_self:
        r = np.arange(nq)
        c = row0 + r
        ok = (c >= 0) & (c < n)
        h[r[ok], c[ok]] = -np.inf
    kk = min(k, n)
    idx = np.full((nq, k), -1, dtype=np.int32)
    score = np.zeros((nq, k), dtype=np.float64)
    count = np.zeros(nq, dtype=np.int32)
    if kk == 0:
        return idx, score, count
    kth = np.partition(h, n - kk, axis=1)[:, n - kk]  # kk-th largest value of each row
    for r in range(nq):
        row = h[r]
        cand = np.nonzero(row >= kth[r])[0]
        cand = cand[np.lexsort((cand, -row[cand]))][:kk]
        sc = row[cand]
        keep = (sc >= min_similarity) & np.isfinite(sc)
        m = int(np.count_nonzero(np.cumprod(keep)))
        count[r] = m
        idx[r, :m] = cand[:m]
        score[r, :m] = sc[:m]
    return idx, score, count


def tie_groups(scores_row, kept_idx, self_idx, tol):
    """For one row: split the kept list into groups of consecutive ranks whose
    score differs by at most ``tol`` from its predecessor, and report whether
    the last group also ties with a column that was *not* kept (a tie that
    straddles the K boundary).  Returns (groups, boundary_tied_candidates)."""
    groups = []
    cur = []
    for j in kept_idx:
        if cur and abs(scores_row[cur[-1]] - scores_row[j]) > tol:
            groups.append(cur)
            cur = []
        cur.append(int(j))
    if cur:
        groups.append(cur)
    boundary = []
    if kept_idx is not None and len(kept_idx) > 0:
        last = scores_row[kept_idx[-1]]
        near = np.nonzero(np.abs(scores_row - last) <= tol)[0]
        kept_set = set(int(x) for x in kept_idx)
        boundary = [int(j) for j in near if int(j) not in kept_set and int(j) != self_idx]
    return groups, boundary


# --------------------------------------------------------------------------
# scripts/populate_database.py:170-218, restated
# --------------------------------------------------------------------------

def _row(x, i):
    return x[i : i + 1]


def populate_rows(features, show_ids, genre_weight=0.4, text_weight=0.5, metadata_weight=0.1,
                  top_n_per_show=20, min_similarity=0.1, rows=None, order_fn=reference_order):
    """Row loop of ``compute_and_store_similarities``.

    Per show: five one-row cosine similarities (populate_database.py:180-186),
    metadata = mean of the three (:187), hybrid with the *raw* weights
    (:190-192), descending argsort (:195), walk (:198-218); shows without a
    surviving neighbour are absent from the dict (:220-221).
    """
    g, t = features["genre_features"], features["text_features"]
    p, ty, la = features["platform_features"], features["type_features"], features["language_features"]
    out = {}
    n = len(show_ids)
    for i in (range(n) if rows is None else rows):
        gs = cosine_similarity(_row(g, i), g)[0]
        ts = cosine_similarity(_row(t, i), t)[0]
        ps = cosine_similarity(_row(p, i), p)[0]
        tys = cosine_similarity(_row(ty, i), ty)[0]
        ls = cosine_similarity(_row(la, i), la)[0]
        ms = (ps + tys + ls) / 3
        h = genre_weight * gs + text_weight * ts + metadata_weight * ms
        kept = walk_row(order_fn(h), h, i, top_n_per_show, min_similarity)
        recs = [
            {
                "similar_show_id": show_ids[j],
                "similarity_score": float(h[j]),
                "genre_score": float(gs[j]),
                "text_score": float(ts[j]),
                "metadata_score": float(ms[j]),
            }
            for j in kept
        ]
        if recs:
            out[show_ids[i]] = recs
    return out


# --------------------------------------------------------------------------
# services/content_based_service.py:113-140, 161-236, restated
# --------------------------------------------------------------------------

def service_hybrid(genre_sim, text_sim, metadata_sim, genre_weight, text_weight, metadata_weight):
    # content_based_service.py:133-138
    total = genre_weight + text_weight + metadata_weight
    return ((genre_weight / total) * genre_sim + (text_weight / total) * text_sim
            + (metadata_weight / total) * metadata_sim)


def recommendations_from_matrix(show_id, show_ids, hybrid, genre_sim, text_sim, metadata_sim,
                                n=10, min_similarity=0.0, order_fn=reference_order):
    """``get_recommendations_from_matrix`` (content_based_service.py:161-236)."""
    id_to_index = {sid: i for i, sid in enumerate(show_ids)}
    if show_id not in id_to_index:
        return []  # content_based_service.py:199-201
    i = id_to_index[show_id]
    row = hybrid[i]
    kept = walk_row(order_fn(row), row, i, n, min_similarity, break_n_after_append=True)
    return [
        {
            "show_id": show_ids[j],
            "similarity_score": float(row[j]),
            "genre_score": float(genre_sim[i, j]),
            "text_score": float(text_sim[i, j]),
            "metadata_score": float(metadata_sim[i, j]),
        }
        for j in kept
    ]


# --------------------------------------------------------------------------
# The hybrid feature matrix Z with H = Z @ Z.T
# --------------------------------------------------------------------------

def dense(x, dtype=np.float64):
    if sp is not None and sp.issparse(x):
        return np.asarray(x.toarray(), dtype=dtype)
    return np.asarray(x, dtype=dtype)


def hybrid_blocks(features, genre_weight, text_weight, metadata_weight, variant="computer"):
    """Blocks and per-block weights whose scaled, normalised concatenation Z
    satisfies Z @ Z.T == hybrid similarity.

    variant "computer": metadata is the cosine of the concatenation and the
    weights are divided by their sum (ml/similarity_computer.py:84-86,112-124).
    variant "populate": metadata is the mean of three cosines and the weights
    are used raw (scripts/populate_database.py:184-192).
    Returns (list of dense float64 blocks, list of weights).
    """
    g = dense(features["genre_features"])
    t = dense(features["text_features"])
    p = dense(features["platform_features"])
    ty = dense(features["type_features"])
    la = dense(features["language_features"])
    if variant == "computer":
        total = genre_weight + text_weight + metadata_weight
        return [g, t, np.hstack([p, ty, la])], [genre_weight / total, text_weight / total,
                                                metadata_weight / total]
    if variant == "populate":
        return [g, t, p, ty, la], [genre_weight, text_weight, metadata_weight / 3,
                                   metadata_weight / 3, metadata_weight / 3]
    raise ValueError(variant)


def build_z(blocks, weights, dtype=np.float64):
    """Z = [sqrt(w_b) * normalise(block_b)]_b (requires w_b >= 0)."""
    cols = []
    for b, w in zip(blocks, weights):
        if w < 0:
            raise ValueError("negative weights have no real square root")
        cols.append(np.sqrt(w) * l2_normalize_rows(np.asarray(b, dtype=np.float64)))
    return np.ascontiguousarray(np.hstack(cols), dtype=dtype)


def topk_from_z(z, k, min_similarity, row0=0, rows=None, chunk=2048, exclude_self=True):
    """Chunked Z[rows] @ Z.T + deterministic top-K without holding N x N."""
    n = z.shape[0]
    rows = range(n) if rows is None else rows
    rows = np.asarray(list(rows))
    idx = np.full((len(rows), k), -1, dtype=np.int32)
    score = np.zeros((len(rows), k), dtype=np.float64)
    count = np.zeros(len(rows), dtype=np.int32)
    zt = z.T
    for s in range(0, len(rows), chunk):
        rr = rows[s : s + chunk]
        h = z[rr] @ zt
        for o, r in enumerate(rr):
            i1, s1, c1 = topk_fast(h[o : o + 1], row0 + int(r), k, min_similarity, exclude_self)
            idx[s + o], score[s + o], count[s + o] = i1[0], s1[0], c1[0]
    return idx, score, count
