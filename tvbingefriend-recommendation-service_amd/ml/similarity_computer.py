"""Drop-in ``SimilarityComputer`` whose arithmetic runs on the MI355X.

Same constructor, the same six methods, the same argument and return types and
the same dict keys as the reference class
(tvbingefriend_recommendation_service/ml/similarity_computer.py:12-190), so
scripts/compute_similarities.py:107-125 and scripts/populate_database.py:145-147
can import this class instead.  The dense-matrix methods are meant for the
sizes the reference can hold in memory; :meth:`topk` is the scalable entry
point that never forms an n x n matrix.

Every method uploads its inputs, runs HIP kernels through the C ABI and
downloads the result.  There is no CPU fallback: without a GPU or without the
built library the methods raise ``NativeLibraryError``.
"""

from __future__ import annotations

import logging
from dataclasses import dataclass

import numpy as np

from .. import features as feat
from .. import native

logger = logging.getLogger(__name__)


@dataclass
class TopKResult:
    """Neighbours of every show, in row order of the feature matrices.

    idx    (n, k) int32 column index of each neighbour, -1 in unused slots
    score  (n, k) float64 hybrid similarity, descending
    count  (n,)   int32 neighbours kept (score >= min_similarity, at most k)
    comp   (n, k, 3) float64 genre / text / metadata cosine of each kept pair
    """

    idx: np.ndarray
    score: np.ndarray
    count: np.ndarray
    comp: np.ndarray


def _torch_device(device):
    import torch

    if not torch.cuda.is_available():
        raise native.NativeLibraryError(
            "no GPU visible: the similarity path runs on MI355X only (there is no CPU fallback)")
    return torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")


# noinspection PyMethodMayBeStatic
class SimilarityComputer:
    """Compute similarity matrices from feature arrays."""

    def __init__(self, genre_weight: float = 0.4, text_weight: float = 0.5,
                 metadata_weight: float = 0.1, device=None):
        self.genre_weight = genre_weight
        self.text_weight = text_weight
        self.metadata_weight = metadata_weight
        self.device = device

    # ---- dense-matrix API (reference: ml/similarity_computer.py:30-130) ----

    def _cosine_matrix(self, features_2d) -> np.ndarray:
        """cosine_similarity(X) with sklearn's dtype rule (metrics/pairwise.py:51-72): float32 input
        is computed and returned in float32 -- one normalise launch and one MFMA tile sweep --,
        everything else in float64: float64 block norms and float64 dot products of the stored
        values (dense or CSR, nothing densified), one wavefront per pair."""
        import torch

        from .. import engine

        dev = _torch_device(self.device)
        if getattr(features_2d, "dtype", None) == np.float32:
            x, layout = feat.pack_blocks([features_2d], [1.0])
            z = engine.normalize_blocks(torch.from_numpy(x).to(dev), layout.offsets, [1.0])
            return engine.gram(z).cpu().numpy()
        if feat._is_sparse(features_2d):
            a = features_2d.tocsr()
            if not a.has_canonical_format:
                a = a.copy()
                a.sum_duplicates()
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            n, d = a.shape
            host = [("csr", a.indptr, a.indices, a.data, d, 0)]
        else:
            a = np.asarray(features_2d)
            if a.ndim != 2:
                raise ValueError(f"Expected 2D array, got {a.ndim}D array instead")
            if a.dtype not in (np.float64, np.int64, np.bool_, np.uint8):
                a = a.astype(np.float64)
            n, d = a.shape
            host = [("dense", np.ascontiguousarray(a), 0)]
        if n == 0 or d == 0:
            return np.zeros((n, n), dtype=np.float64)
        sources = engine.upload_sources(host, dev)
        offsets = [0, d]
        norms = engine.block_norms(sources, n, offsets)  # NaN / infinity raise ValueError, as sklearn's validation does
        out = torch.empty((n, n), dtype=torch.float64, device=dev)
        for r0 in range(0, n, 32768):
            ids = torch.arange(r0, min(n, r0 + 32768), dtype=torch.int32, device=dev)
            out[r0:r0 + ids.numel()] = engine.score_rows(sources, n, offsets, [0], [1.0, 0.0, 0.0], norms, ids,
                                                         exclude_self=False)
        # (x_i / |x_i|) . x_j / |x_j| and its transpose differ in the last bit: keep one of the two
        upper = torch.triu(out)
        return (upper + torch.triu(out, 1).T).cpu().numpy()

    def compute_genre_similarity(self, genre_features: np.ndarray) -> np.ndarray:
        logger.info("Computing genre similarity...")
        similarity = self._cosine_matrix(genre_features)
        logger.info(
            f" Genre similarity: {similarity.shape}, range [{similarity.min():.3f}, {similarity.max():.3f}]"
        )
        return similarity

    def compute_text_similarity(self, text_features) -> np.ndarray:
        logger.info("Computing text similarity...")
        similarity = self._cosine_matrix(text_features)
        logger.info(
            f" Text similarity: {similarity.shape}, range [{similarity.min():.3f}, {similarity.max():.3f}]"
        )
        return similarity

    def compute_metadata_similarity(self, platform_features: np.ndarray, type_features: np.ndarray,
                                    language_features: np.ndarray) -> np.ndarray:
        logger.info("Computing metadata similarity...")
        metadata_features = np.hstack([platform_features, type_features, language_features])
        similarity = self._cosine_matrix(metadata_features)
        logger.info(
            f" Metadata similarity: {similarity.shape}, range [{similarity.min():.3f}, {similarity.max():.3f}]"
        )
        return similarity

    def compute_hybrid_similarity(self, genre_similarity: np.ndarray, text_similarity: np.ndarray,
                                  metadata_similarity: np.ndarray) -> np.ndarray:
        import torch

        from .. import engine

        logger.info("Computing hybrid similarity...")
        total_weight = self.genre_weight + self.text_weight + self.metadata_weight
        genre_w = self.genre_weight / total_weight
        text_w = self.text_weight / total_weight
        metadata_w = self.metadata_weight / total_weight
        logger.info(f"  Weights - Genre: {genre_w:.2f}, Text: {text_w:.2f}, Metadata: {metadata_w:.2f}")

        dev = _torch_device(self.device)
        mats = [np.ascontiguousarray(m, dtype=np.float64) for m in
                (genre_similarity, text_similarity, metadata_similarity)]
        if not (mats[0].shape == mats[1].shape == mats[2].shape):
            raise ValueError("similarity matrices must have one shape")
        g, t, m = (torch.from_numpy(a).to(dev) for a in mats)
        hybrid_similarity = engine.weighted_sum3(g, t, m, genre_w, text_w, metadata_w).cpu().numpy()
        logger.info(
            f" Hybrid similarity: {hybrid_similarity.shape}, range [{hybrid_similarity.min():.3f}, "
            f"{hybrid_similarity.max():.3f}]"
        )
        return hybrid_similarity

    def compute_all_similarities(self, features: dict) -> dict:
        logger.info("=" * 60)
        logger.info("COMPUTING ALL SIMILARITIES")
        logger.info("=" * 60)
        genre_similarity = self.compute_genre_similarity(features["genre_features"])
        text_similarity = self.compute_text_similarity(features["text_features"])
        metadata_similarity = self.compute_metadata_similarity(
            features["platform_features"], features["type_features"], features["language_features"]
        )
        hybrid_similarity = self.compute_hybrid_similarity(
            genre_similarity, text_similarity, metadata_similarity
        )
        logger.info("=" * 60)
        logger.info("SIMILARITY COMPUTATION COMPLETE")
        logger.info("=" * 60)
        return {
            "genre_similarity": genre_similarity,
            "text_similarity": text_similarity,
            "metadata_similarity": metadata_similarity,
            "hybrid_similarity": hybrid_similarity,
        }

    def get_similarity_statistics(self, similarity_matrix: np.ndarray) -> dict:
        """mean / std / min / max / median over the strict upper triangle
        (reference: ml/similarity_computer.py:171-190)."""
        import torch

        from .. import engine

        m = np.ascontiguousarray(similarity_matrix, dtype=np.float64)
        if m.ndim != 2 or m.shape[0] != m.shape[1]:
            raise ValueError("input array must be square")  # np.triu_indices_from would need 2-D
        if m.shape[0] < 2:
            # NumPy's .min() of the empty upper triangle raises the same way
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        dev = _torch_device(self.device)
        out = engine.upper_tri_stats(torch.from_numpy(m).to(dev)).cpu().numpy()
        return {"mean": float(out[0]), "std": float(out[1]), "min": float(out[2]),
                "max": float(out[3]), "median": float(out[4])}

    def _device_sources(self, features: dict, variant: str):
        """The five stored feature arrays on the device, their column layout, and the number of shows."""
        from .. import engine

        dev = _torch_device(self.device)
        sources, layout = feat.block_sources(features, self.genre_weight, self.text_weight,
                                             self.metadata_weight, variant)
        n = int(sources[0][1].shape[0]) - (1 if sources[0][0] == "csr" else 0)
        return engine.upload_sources(sources, dev), layout, n

    def get_feature_statistics(self, features: dict) -> dict:
        """The four statistics blocks that scripts/compute_similarities.py:119-131 logs (genre,
        text, metadata, hybrid), each what :meth:`get_similarity_statistics` returns for that
        matrix, computed from the features without forming any n x n matrix."""
        from .. import engine

        dev_sources, layout, n = self._device_sources(features, "computer")
        if n < 2:
            raise ValueError("zero-size array to reduction operation minimum which has no identity")
        engine.block_norms(dev_sources, n, layout.offsets)  # sklearn's input validation: NaN / infinity raise
        out = {}
        hybrid_scales = layout.scales()
        for name, scales in (("genre_similarity", [1.0, 0.0, 0.0]), ("text_similarity", [0.0, 1.0, 0.0]),
                             ("metadata_similarity", [0.0, 0.0, 1.0]), ("hybrid_similarity", hybrid_scales)):
            # a block with scale 0 contributes nothing to the product: one layout serves all four
            z = engine.assemble_normalize(dev_sources, n, layout.offsets, scales)
            v = engine.gram_stats(z).cpu().numpy()
            out[name] = {"mean": float(v[0]), "std": float(v[1]), "min": float(v[2]),
                         "max": float(v[3]), "median": float(v[4])}
        return out

    # ---- scalable entry point: no n x n matrix ----

    def group_weights(self, variant: str) -> list:
        """The three weights of the reference's hybrid expression: raw for the production loop
        (scripts/populate_database.py:190-192), divided by their sum for this class
        (ml/similarity_computer.py:112-115)."""
        if variant == "populate":
            return [self.genre_weight, self.text_weight, self.metadata_weight]
        total = self.genre_weight + self.text_weight + self.metadata_weight
        return [self.genre_weight / total, self.text_weight / total, self.metadata_weight / total]

    def topk(self, features: dict, k: int = 20, min_similarity: float = 0.1,
             variant: str = "computer", group=None) -> TopKResult:
        """Top-k hybrid neighbours of every show straight from the feature blocks.

        ``variant`` selects which of the reference's two definitions is followed:
        "computer" (this class: metadata = cosine of the concatenation, weights
        divided by their sum) or "populate" (scripts/populate_database.py:184-192:
        metadata = mean of three cosines, raw weights).

        The fp32 MFMA sweep collects a few more than k candidates per show a hair below the
        threshold; their scores are then recomputed in float64 from the stored arrays, so the
        order, the scores and the ``score < min_similarity`` cut are the reference's own float64
        decisions (scripts/populate_database.py:190-218); equal float64 scores go by index.

        ``group``: a torch.distributed process group (or True for the default group) whose ranks
        all make this call with the same features -- one process per GPU.  Every rank builds the
        normalised matrix from the features it holds, computes and re-ranks the lists of its own
        rows (exchanging bound keys and column-side lists with the others on the symmetric route),
        and the lists are gathered: every rank returns the whole result, identical to the
        single-GPU call's.
        """
        from .. import engine

        if group is not None:
            return self._topk_sharded(features, k, min_similarity, variant, None if group is True else group)
        # the stored arrays go to the device as they are (CSR text stays CSR); Z is built there
        dev_sources, layout, n = self._device_sources(features, variant)
        if min(layout.weights) >= 0:
            # Z = [sqrt(w_b) * unit block]: Z Z^T is the hybrid matrix, every pair computed once
            z = engine.assemble_normalize(dev_sources, n, layout.offsets, layout.scales())
            z_q = None
        else:
            # a negative weight has no square root: unit blocks on the column side, the weights on the
            # query side only (a plain sweep; the reference accepts any weights, ml/similarity_computer.py:112-124)
            z = engine.assemble_normalize(dev_sources, n, layout.offsets, [1.0] * layout.nb)
            z_q = engine.assemble_normalize(dev_sources, n, layout.offsets, [float(w) for w in layout.weights])
        # the float64 norms are computed inside; NaN / infinity raise ValueError there, as sklearn does
        top = engine.exact_topk(z, dev_sources, layout.offsets, layout.groups, self.group_weights(variant),
                                k, min_similarity, n=n, z_q=z_q)
        self.last_topk_stats = dict(top.stats)
        return TopKResult(top.idx.cpu().numpy(), top.score.cpu().numpy(), top.count.cpu().numpy(),
                          top.comp.cpu().numpy())

    def _topk_sharded(self, features: dict, k: int, min_similarity: float, variant: str, group) -> TopKResult:
        """:meth:`topk` as one rank of a row-sharded job (scripts/populate_database.py:170-218 with the
        shows split over the ranks).  The symmetric sharded sweep when the shape allows it (every
        block between two shards computed by one of them), else the plain sharded sweep."""
        import torch
        import torch.distributed as dist

        from .. import engine, sharding

        world, rank = dist.get_world_size(group), dist.get_rank(group)
        dev_sources, layout, n = self._device_sources(features, variant)
        if min(layout.weights) < 0:
            raise ValueError("negative similarity weights are served by the single-GPU call only")
        dev = dev_sources[0].data.device
        norms = engine.block_norms(dev_sources, n, layout.offsets)
        sh = sharding.shard_bounds(n, world, rank)
        ld = engine.padded_ld(layout.d)
        # The stored features of all shows are on this rank anyway (the float64 re-rank reads the
        # candidates' rows), so it builds the normalised rows of ALL shows itself -- 0.15 ms for
        # 100,000 x 1024, less than any all-gather of them -- and only the exchanges that carry results
        # of other ranks' work stay collectives (bound keys, column-side lists, the final lists).
        z_all = torch.zeros((sh.padded_total, ld), dtype=torch.float32, device=dev)
        engine.assemble_normalize(dev_sources, n, layout.offsets, layout.scales(), out=z_all[:n])
        z_shard = z_all[rank * sh.per_rank:(rank + 1) * sh.per_rank]
        kc = engine.candidate_count(k)
        floor = min_similarity - engine.error_margin(ld, self.group_weights(variant))
        if sharding.symmetric_route_fits(world, kc, ld):
            cand, z_all = sharding.sharded_topk_symmetric_z(z_shard, sh, kc, floor, group=group, z_all=z_all)
        else:
            cand, z_all = sharding.sharded_topk_z(z_shard, sh, kc, floor, group=group, z_all=z_all)
        top = engine.exact_topk(z_all, dev_sources, layout.offsets, layout.groups, self.group_weights(variant), k,
                                min_similarity, n=n, q_row0=sh.row0, nq=sh.rows, cand=cand, norms=norms)
        self.last_topk_stats = dict(top.stats)
        via_host = dist.get_backend(group) == "gloo"  # gloo: CPU tensors only
        parts = []
        for t in (top.idx, top.score, top.count, top.comp):
            t = t.cpu() if via_host else t
            parts.append(sharding.gather_lists(t, sh, group).cpu().numpy())
        return TopKResult(*parts)
