"""Tie-aware comparison of a top-K result with the float64 oracle.

Bar (stated once, used by every parity test):
  * scores: |result - oracle_f64| <= SCORE_TOL = 1e-5;
  * indices: exactly the oracle's at every rank whose float64 score differs from
    both neighbouring ranks (rank K+1 included) by more than GAP_TOL = 2e-6 --
    several times the largest |fp32 - float64| score error these tests see
    (about 3e-7 at D = 512, 4e-7 at D = 2048; every test asserts <= SCORE_TOL and
    the golden tests assert <= 1e-6), so rounding cannot swap such ranks; inside
    a group of closer scores the result must hold members of that group, each
    with its own correct score, no duplicates;
  * order: score descending, equal fp32 scores by column index ascending;
  * count: the number of scores >= min_similarity capped at K (a score within
    SCORE_TOL of the threshold may fall on either side).
The reference itself orders equal scores by NumPy's unstable argsort
(scripts/populate_database.py:195), so exact index equality inside tie groups
is not defined by the reference either.
"""

from __future__ import annotations

import numpy as np

SCORE_TOL = 1e-5
GAP_TOL = 2e-6
# Refined mode -- results of the float64 re-rank (engine.exact_topk, SimilarityComputer.topk):
# scores within REFINED_TOL of the oracle's float64 scores and indices exactly the oracle's at every
# rank whose float64 score differs from both neighbouring ranks by more than REFINED_GAP.  What is
# left out are exact float64 ties (and pairs of scores a few units in the last place apart, whose
# order depends on the summation order of the BLAS that computed them): there the reference's own
# order comes from an unstable argsort.  Counts must equal the oracle's where no score lies within
# REFINED_TOL of the threshold.
REFINED_TOL = 1e-12
REFINED_GAP = 1e-12


def check_topk_against_scores(h_rows, row0, k, min_similarity, got_idx, got_score, got_count,
                              exclude_self=True, tol=SCORE_TOL, what="", check_tie_order=True, gap=GAP_TOL,
                              stats=None, f64=False):
    """``h_rows`` holds the oracle's full float64 score rows for the query rows.
    ``check_tie_order`` also demands index order among scores that are equal in fp32
    (the kernels' rule; a float64 stand-in cannot be held to it).  ``stats`` (a dict) collects
    ``ranks`` compared, ``exact`` (ranks held to index equality) and ``max_err`` (largest score error).
    ``f64``: the result's scores are float64 (their own order is checked as float64).
    Returns the number of ranks held to (and meeting) index equality."""
    h_rows = np.asarray(h_rows, dtype=np.float64)
    nq, n = h_rows.shape
    got_idx = np.asarray(got_idx)
    got_score = np.asarray(got_score, dtype=np.float64)
    got_count = np.asarray(got_count)
    assert got_idx.shape == (nq, k) and got_score.shape == (nq, k) and got_count.shape == (nq,), what
    exact_positions = 0
    for r in range(nq):
        h = h_rows[r].copy()
        if exclude_self and 0 <= row0 + r < n:
            h[row0 + r] = -np.inf
        order = np.lexsort((np.arange(n), -h))
        srt = h[order]
        lo = min(k, int(np.count_nonzero(h >= min_similarity + tol)))
        hi = min(k, int(np.count_nonzero(h >= min_similarity - tol)))
        c = int(got_count[r])
        assert lo <= c <= hi, f"{what} row {r}: count {c} not in [{lo},{hi}]"
        gi = got_idx[r, :c]
        gs = got_score[r, :c]
        assert np.all(got_idx[r, c:] == -1), f"{what} row {r}: unused slots must be -1"
        assert np.all(got_score[r, c:] == 0), f"{what} row {r}: unused scores must be 0"
        assert np.all((gi >= 0) & (gi < n)), f"{what} row {r}: index out of range"
        assert len(set(gi.tolist())) == c, f"{what} row {r}: duplicate neighbour"
        if exclude_self:
            assert row0 + r not in set(gi.tolist()), f"{what} row {r}: self not excluded"
        # every reported score is the oracle's score of that very column
        assert np.all(np.abs(gs - h[gi]) <= tol), (
            f"{what} row {r}: score error {np.abs(gs - h[gi]).max():.3e}")
        # rank-wise: the sorted score sequence is unique whatever the tie order
        assert np.all(np.abs(gs - srt[:c]) <= tol), (
            f"{what} row {r}: rank-wise score mismatch {np.abs(gs - srt[:c]).max():.3e}")
        # order of the result itself
        d = np.diff(gs if f64 else gs.astype(np.float32))
        assert np.all(d <= 0), f"{what} row {r}: scores not descending"
        same = d == 0
        if check_tie_order:
            assert np.all(np.diff(gi)[same] > 0), f"{what} row {r}: equal scores not in index order"
        # exact index parity away from (near-)ties
        ext = np.full(c + 2, -np.inf)
        ext[0] = np.inf
        m = min(c + 1, n)
        ext[1 : 1 + m] = srt[:m]
        clear = (ext[:c] - ext[1 : c + 1] > gap) & (ext[1 : c + 1] - ext[2 : c + 2] > gap)
        assert np.array_equal(gi[clear], order[:c][clear]), (
            f"{what} row {r}: index mismatch at ranks {np.nonzero(gi[clear] != order[:c][clear])[0]}")
        exact_positions += int(clear.sum())
        if stats is not None:
            stats["ranks"] = stats.get("ranks", 0) + c
            stats["exact"] = stats.get("exact", 0) + int(clear.sum())
            stats["max_err"] = max(stats.get("max_err", 0.0), float(np.abs(gs - h[gi]).max(initial=0.0)))
    return exact_positions


def check_topk_against_golden(ref_idx, ref_sorted, k, got_idx, got_score, got_count, pair_score,
                              min_similarity=None, tol=SCORE_TOL, what="", gap=GAP_TOL, clear_mask=None,
                              stats=None):
    """Compare with a stored reference result.

    ``ref_idx`` (nq,k) are the reference's kept columns (-1 padded),
    ``ref_sorted`` (nq,k+1) the k+1 largest float64 scores of every row (self
    excluded); ``pair_score(r, cols)`` returns the oracle's float64 scores of
    row r against ``cols`` and is used for members of tie groups.
    ``clear_mask`` (nq,k) is the fixture's own record of the ranks held to index
    equality; it must agree with the mask derived here from ``ref_sorted``.
    """
    ref_idx = np.asarray(ref_idx)
    ref_sorted = np.asarray(ref_sorted, dtype=np.float64)
    nq = ref_idx.shape[0]
    got_idx = np.asarray(got_idx)
    got_score = np.asarray(got_score, dtype=np.float64)
    got_count = np.asarray(got_count)
    exact_positions = 0
    for r in range(nq):
        srt = ref_sorted[r]
        if min_similarity is None:
            lo = hi = int(np.count_nonzero(ref_idx[r] >= 0))
        else:
            lo = min(k, int(np.count_nonzero(srt[:k] >= min_similarity + tol)))
            hi = min(k, int(np.count_nonzero(srt[:k] >= min_similarity - tol)))
        c = int(got_count[r])
        assert lo <= c <= hi, f"{what} row {r}: count {c} not in [{lo},{hi}]"
        gi, gs = got_idx[r, :c], got_score[r, :c]
        assert np.all(got_idx[r, c:] == -1), f"{what} row {r}: unused slots must be -1"
        assert len(set(gi.tolist())) == c and r not in set(gi.tolist()), f"{what} row {r}: dup/self"
        assert np.all(np.abs(gs - srt[:c]) <= tol), (
            f"{what} row {r}: rank-wise score mismatch {np.abs(gs - srt[:c]).max():.3e}")
        ext = np.concatenate([[np.inf], srt, [-np.inf]])
        clear = (ext[:c] - ext[1 : c + 1] > gap) & (ext[1 : c + 1] - ext[2 : c + 2] > gap)
        if clear_mask is not None:
            assert np.array_equal(clear, np.asarray(clear_mask)[r, :c]), f"{what} row {r}: stored clear mask differs"
        assert np.array_equal(gi[clear], ref_idx[r, :c][clear]), f"{what} row {r}: index mismatch"
        exact_positions += int(clear.sum())
        if stats is not None:
            stats["ranks"] = stats.get("ranks", 0) + c
            stats["exact"] = stats.get("exact", 0) + int(clear.sum())
            stats["max_err"] = max(stats.get("max_err", 0.0), float(np.abs(gs - srt[:c]).max(initial=0.0)))
        if not np.all(clear):
            amb = np.nonzero(~clear)[0]
            assert np.all(np.abs(pair_score(r, gi[amb]) - gs[amb]) <= tol), (
                f"{what} row {r}: a tie-group member carries the wrong score")
    return exact_positions
