"""MI355X counterpart of the matrix-based methods of the reference's
``ContentBasedRecommendationService`` (services/content_based_service.py):
``_load_similarity_matrices`` (:113-140), ``_load_show_mappings`` (:142-159),
``get_recommendations_from_matrix`` (:161-236) and
``compute_and_store_all_similarities`` (:262-338), plus
``get_recommendations_from_features``: the same lookup computed from the feature
files, for catalogues whose n x n matrices cannot be stored.

The three dense matrices are loaded from disk as in the reference; the hybrid
matrix is formed on the GPU with the same float64 arithmetic (bit for bit),
stays in HBM, and rows are ranked by a selection kernel instead of a full
argsort per show.  Database access, blob download and metadata sync are the
reference's own and are not re-implemented: the sink is injected.
"""

from __future__ import annotations

import logging
from pathlib import Path

import numpy as np

from .. import native

logger = logging.getLogger(__name__)


class ContentBasedRecommendationService:
    """Serve recommendations from pre-computed similarity matrices."""

    def __init__(self, processed_data_dir: Path | None = None, genre_weight: float = 0.4,
                 text_weight: float = 0.5, metadata_weight: float = 0.1, use_blob: bool | None = None,
                 blob_prefix: str = "processed", device=None, repo_factory=None):
        if processed_data_dir is None:
            processed_data_dir = Path.cwd() / "data" / "processed"
        if use_blob:
            raise NotImplementedError(
                "blob download belongs to the reference's storage layer; download the files first "
                "and pass processed_data_dir")
        self.processed_data_dir = Path(processed_data_dir)
        self.genre_weight = genre_weight
        self.text_weight = text_weight
        self.metadata_weight = metadata_weight
        self.use_blob = False
        self.blob_prefix = blob_prefix
        self.device = device
        self._repo_factory = repo_factory

        self._genre_similarity = None      # torch float64 (n, n) on the GPU
        self._text_similarity = None
        self._metadata_similarity = None
        self._hybrid_similarity = None
        self._feature_z = None             # torch fp32 (n, ld): per-block unit rows, kept in HBM
        self._feature_sources = None       # the stored feature arrays on the device
        self._feature_norms = None         # (n, nb) float64 block norms
        self._feature_layout = None
        self._show_id_to_index: dict | None = None
        self._index_to_show_id: dict | None = None
        logger.info("Initialized ContentBasedRecommendationService")
        logger.info(f"Weights - Genre: {genre_weight}, Text: {text_weight}, Metadata: {metadata_weight}")

    def _get_data_dir(self) -> Path:
        return self.processed_data_dir

    def _device(self):
        import torch

        if not torch.cuda.is_available():
            raise native.NativeLibraryError(
                "no GPU visible: the similarity path runs on MI355X only (there is no CPU fallback)")
        return torch.device(self.device if self.device is not None else f"cuda:{torch.cuda.current_device()}")

    def _load_similarity_matrices(self):
        """np.load the three matrices, form the hybrid with normalised weights on the GPU."""
        if self._hybrid_similarity is not None:
            return
        import torch

        from .. import engine

        logger.info("Loading similarity matrices...")
        data_dir = self._get_data_dir()
        dev = self._device()
        mats = []
        for name in ("genre_similarity", "text_similarity", "metadata_similarity"):
            m = np.ascontiguousarray(np.load(data_dir / f"{name}.npy"), dtype=np.float64)
            mats.append(torch.from_numpy(m).to(dev))
        self._genre_similarity, self._text_similarity, self._metadata_similarity = mats
        total_weight = self.genre_weight + self.text_weight + self.metadata_weight
        self._hybrid_similarity = engine.weighted_sum3(
            mats[0], mats[1], mats[2], self.genre_weight / total_weight,
            self.text_weight / total_weight, self.metadata_weight / total_weight)
        logger.info(f"✓ Loaded similarity matrices: {tuple(self._hybrid_similarity.shape)}")

    def _load_show_mappings(self):
        """Row order of shows_metadata.csv["id"] defines the matrix index of every show."""
        if self._show_id_to_index is not None:
            return
        import pandas as pd

        metadata_df = pd.read_csv(self._get_data_dir() / "shows_metadata.csv")
        self._show_id_to_index = {show_id: idx for idx, show_id in enumerate(metadata_df["id"])}
        self._index_to_show_id = {idx: show_id for show_id, idx in self._show_id_to_index.items()}
        logger.info(f"✓ Loaded mappings for {len(self._show_id_to_index)} shows")

    def _rank_rows(self, row0: int, rows: int, n: int, min_similarity: float):
        """Selection kernel + gathers of the three component matrices for rows [row0, row0+rows)."""
        import torch

        from .. import engine

        idx, score, count = engine.dense_rows_topk(self._hybrid_similarity, n, min_similarity,
                                                   row0=row0, rows=rows)
        safe = idx.clamp_min(0).long()
        sl = slice(row0, row0 + rows)
        comps = [torch.gather(m[sl], 1, safe) for m in
                 (self._genre_similarity, self._text_similarity, self._metadata_similarity)]
        return (idx.cpu().numpy(), score.cpu().numpy(), count.cpu().numpy(),
                [c.cpu().numpy() for c in comps])

    def get_recommendations_from_matrix(self, show_id: int, n: int = 10,
                                        min_similarity: float = 0.0) -> list:
        """Top-n neighbours of one show straight from the matrices (no database)."""
        self._load_similarity_matrices()
        self._load_show_mappings()
        if show_id not in self._show_id_to_index:
            logger.warning(f"Show ID {show_id} not found in similarity matrix")
            return []
        if n < 1:
            # the reference appends the best neighbour before it tests the length (:223-234)
            n_eff = 1
        else:
            n_eff = n
        show_idx = self._show_id_to_index[show_id]
        idx, score, count, comps = self._rank_rows(show_idx, 1, n_eff, min_similarity)
        return self._records(idx[0], score[0], int(count[0]), [c[0] for c in comps], "show_id")

    def _load_feature_matrix(self):
        """The five feature files of scripts/compute_features.py:115-129 -> block-normalised Z
        on the GPU (unit scale per block, so any weighting can be applied to a query vector); the
        stored arrays and their float64 block norms stay on the device for the float64 re-rank."""
        if self._feature_z is not None:
            return
        from .. import engine, features as feat
        from ..scripts.compute_similarities import load_features

        dev = self._device()
        sources, layout = feat.block_sources(load_features(self._get_data_dir()), 1.0, 1.0, 1.0, "computer")
        n = int(sources[0][1].shape[0]) - (1 if sources[0][0] == "csr" else 0)
        self._feature_sources = engine.upload_sources(sources, dev)
        self._feature_norms = engine.block_norms(self._feature_sources, n, layout.offsets)
        self._feature_z = engine.assemble_normalize(self._feature_sources, n, layout.offsets, [1.0] * layout.nb)
        self._feature_layout = layout
        logger.info(f"✓ Feature matrix on device: {tuple(self._feature_z.shape)}")

    def get_recommendations_from_features(self, show_id: int, n: int = 10, min_similarity: float = 0.0,
                                          genre_weight: float | None = None, text_weight: float | None = None,
                                          metadata_weight: float | None = None) -> list:
        """Same records as :meth:`get_recommendations_from_matrix`, computed from the feature
        files with one pass over the catalogue instead of stored n x n matrices; the weights
        default to the service's and may differ per call (normalised by their sum as in the
        reference, services/content_based_service.py:133-138; negative weights are fine: they
        scale the query vector only).  Scores, order and the threshold cut are float64."""
        from .. import engine

        self._load_feature_matrix()
        self._load_show_mappings()
        if show_id not in self._show_id_to_index:
            logger.warning(f"Show ID {show_id} not found in similarity matrix")
            return []
        gw = self.genre_weight if genre_weight is None else genre_weight
        tw = self.text_weight if text_weight is None else text_weight
        mw = self.metadata_weight if metadata_weight is None else metadata_weight
        total = gw + tw + mw
        weights = [gw / total, tw / total, mw / total]
        z, layout = self._feature_z, self._feature_layout
        row = self._show_id_to_index[show_id]
        q = z[row:row + 1].clone()
        for b in range(layout.nb):
            q[:, layout.offsets[b]:layout.offsets[b + 1]] *= weights[b]
        k = max(n, 1)
        floor = min_similarity - engine.error_margin(z.shape[1], weights)
        cand = None
        if k <= native.TVBF_MAX_K:
            cand = engine.query_topk(q, z, engine.candidate_count(k), floor, self_index=[row])
        top = engine.exact_topk(z, self._feature_sources, layout.offsets, layout.groups, weights, k, min_similarity,
                                q_row0=row, nq=1, cand=cand, norms=self._feature_norms,
                                resweep=lambda ids, kk: engine.query_topk(q, z, kk, floor))
        comp = top.comp.cpu().numpy()[0]
        return self._records(top.idx.cpu().numpy()[0], top.score.cpu().numpy()[0], int(top.count.cpu().numpy()[0]),
                             [comp[:, 0], comp[:, 1], comp[:, 2]], "show_id")

    def _records(self, idx, score, count, comps, id_key):
        return [
            {
                id_key: self._index_to_show_id[int(idx[s])],
                "similarity_score": float(score[s]),
                "genre_score": float(comps[0][s]),
                "text_score": float(comps[1][s]),
                "metadata_score": float(comps[2][s]),
            }
            for s in range(count)
        ]

    def compute_and_store_all_similarities(self, top_n_per_show: int = 20, min_similarity: float = 0.1,
                                           repo=None) -> dict:
        """Rank every row in one launch and store the lists through ``repo``
        (``bulk_store_all_similarities`` / ``get_similarity_stats``; default: the factory given
        to the constructor, else the reference's SimilarityRepository on its own session)."""
        self._load_similarity_matrices()
        self._load_show_mappings()
        total_shows = len(self._show_id_to_index)
        logger.info(f"Computing similarities for {total_shows} shows...")
        idx, score, count, comps = self._rank_rows(0, total_shows, max(top_n_per_show, 1), min_similarity)
        all_similarities = {}
        for r in range(total_shows):
            c = int(count[r])
            if c:
                all_similarities[self._index_to_show_id[r]] = self._records(
                    idx[r], score[r], c, [m[r] for m in comps], "similar_show_id")
        logger.info(f"✓ Computed similarities for {len(all_similarities)} shows")

        db = None
        if repo is None and self._repo_factory is not None:
            repo = self._repo_factory()
        if repo is None:
            from tvbingefriend_recommendation_service.models.database import SessionLocal
            from tvbingefriend_recommendation_service.repos import SimilarityRepository

            db = SessionLocal()
            repo = SimilarityRepository(db)
        try:
            total_records = repo.bulk_store_all_similarities(all_similarities)
            stats = repo.get_similarity_stats()
            stats["computed_shows"] = len(all_similarities)
            stats["top_n_per_show"] = top_n_per_show
            stats["min_similarity"] = min_similarity
            logger.info(f"Total records stored: {total_records}")
            return stats
        finally:
            if db is not None:
                db.close()
