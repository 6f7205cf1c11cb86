"""MI355X counterpart of the reference's scripts/compute_similarities.py.

Same functions, same file contract and error messages
(scripts/compute_similarities.py:29-81 load_features, :84-133
compute_similarities, :136-177 save_similarities, :180-262 main); the
matrices come from the HIP-backed ``SimilarityComputer``.
"""

from __future__ import annotations

import argparse
import logging
import sys
from pathlib import Path

import numpy as np

from ..ml.similarity_computer import SimilarityComputer

logger = logging.getLogger(__name__)

REQUIRED_FILES = [
    "genre_features.npy",
    "text_features.npz",
    "platform_features.npy",
    "type_features.npy",
    "language_features.npy",
]


def load_features(input_dir: Path) -> dict:
    """Load the five feature files written by compute_features.py."""
    from scipy.sparse import load_npz

    input_dir = Path(input_dir)
    logger.info("=" * 70)
    logger.info("LOADING FEATURES")
    logger.info("=" * 70)
    for filename in REQUIRED_FILES:
        filepath = input_dir / filename
        if not filepath.exists():
            raise FileNotFoundError(
                f"Feature file not found: {filepath}\n" "Run compute_features.py first."
            )
    features = {}
    for filename in REQUIRED_FILES:
        key = filename.split(".")[0]
        loader = load_npz if filename.endswith(".npz") else np.load
        features[key] = loader(input_dir / filename)
        logger.info(f"✓ Loaded {key}: {features[key].shape}")
    return features


# The reference holds four dense float64 n x n matrices at once (32 n^2 bytes: ~205 GB at 80,000
# shows, scripts/compute_similarities.py:155).  Above this many bytes the script takes the
# matrix-free route: the statistics block from the features (one sweep per matrix, nothing stored)
# and the top-K neighbour lists instead of matrices.
MATRIX_BYTES_LIMIT = 8 << 30


def matrix_bytes(features: dict) -> int:
    n = int(features["genre_features"].shape[0])
    return 32 * n * n


def _log_statistics(all_stats: dict) -> None:
    logger.info("\n" + "=" * 70)
    logger.info("SIMILARITY STATISTICS")
    logger.info("=" * 70)
    for sim_name in ["genre_similarity", "text_similarity", "metadata_similarity", "hybrid_similarity"]:
        stats = all_stats[sim_name]
        logger.info(f"\n{sim_name}:")
        logger.info(f"  Mean: {stats['mean']:.4f}")
        logger.info(f"  Std:  {stats['std']:.4f}")
        logger.info(f"  Min:  {stats['min']:.4f}")
        logger.info(f"  Max:  {stats['max']:.4f}")
        logger.info(f"  Median: {stats['median']:.4f}")


def compute_similarities(features: dict, genre_weight: float = 0.4, text_weight: float = 0.5,
                         metadata_weight: float = 0.1, matrices: bool | None = None, top_n: int = 20,
                         min_similarity: float = 0.1) -> dict:
    """All four similarity matrices plus the statistics log block
    (scripts/compute_similarities.py:84-133 of the reference).

    ``matrices`` False -- or None and 32 n^2 bytes above ``MATRIX_BYTES_LIMIT`` -- takes the
    matrix-free route for catalogues whose matrices cannot be held: the same statistics block,
    computed from the features without storing a matrix, and a dict
    ``{"statistics": {...four blocks...}, "top_neighbours": TopKResult}`` (the ``top_n`` best
    neighbours per show above ``min_similarity``: what the pipeline stores downstream,
    scripts/populate_database.py:309-317) instead of the four n x n arrays."""
    logger.info("=" * 70)
    logger.info("COMPUTING SIMILARITIES")
    logger.info("=" * 70)
    computer = SimilarityComputer(
        genre_weight=genre_weight, text_weight=text_weight, metadata_weight=metadata_weight
    )
    if matrices is None:
        matrices = matrix_bytes(features) <= MATRIX_BYTES_LIMIT
    if not matrices:
        n = int(features["genre_features"].shape[0])
        logger.info(f"Matrix-free route: {n} shows ({matrix_bytes(features) / 2**30:.1f} GiB of float64 matrices not formed)")
        all_stats = computer.get_feature_statistics(features)
        _log_statistics(all_stats)
        top = computer.topk(features, k=top_n, min_similarity=min_similarity)
        logger.info(f"✓ Top-{top_n} neighbours of {n} shows: {int(top.count.sum())} pairs at or above {min_similarity}")
        return {"statistics": all_stats, "top_neighbours": top}
    similarities = computer.compute_all_similarities(features)
    _log_statistics({name: computer.get_similarity_statistics(similarities[name]) for name in
                     ["genre_similarity", "text_similarity", "metadata_similarity", "hybrid_similarity"]})
    return similarities


def save_similarities(similarities: dict, output_dir: Path, save_to_disk: bool = False):
    """Optionally write the dense matrices (off by default, as in the reference)."""
    logger.info("\n" + "=" * 70)
    logger.info("SIMILARITY STORAGE")
    logger.info("=" * 70)
    if not save_to_disk:
        logger.info("⊘ Skipping similarity matrix storage (storage optimization)")
        logger.info("  Use --save-similarities flag to save for analysis/debugging")
        return
    if "top_neighbours" in similarities:
        raise ValueError("the matrix-free route forms no similarity matrices to save "
                         "(drop --no-matrices, or raise MATRIX_BYTES_LIMIT, for catalogues whose matrices fit)")
    output_dir = Path(output_dir)
    output_dir.mkdir(parents=True, exist_ok=True)
    total_size = 0
    for name, matrix in similarities.items():
        output_path = output_dir / f"{name}.npy"
        np.save(output_path, matrix)
        size = output_path.stat().st_size
        total_size += size
        logger.info(f"✓ Saved {name}.npy ({size / 1024 / 1024:.2f} MB)")
    logger.info(f"\n✓ All similarities saved to {output_dir}")
    logger.info(f"  Total size: {total_size / 1024 / 1024:.2f} MB")


def main(argv=None):
    parser = argparse.ArgumentParser(description="Compute similarity matrices from feature matrices")
    parser.add_argument("--input-dir", type=str, default="data/processed")
    parser.add_argument("--output-dir", type=str, default="data/processed")
    parser.add_argument("--genre-weight", type=float, default=0.4)
    parser.add_argument("--text-weight", type=float, default=0.5)
    parser.add_argument("--metadata-weight", type=float, default=0.1)
    parser.add_argument("--save-similarities", action="store_true")
    parser.add_argument("--no-matrices", action="store_true",
                        help="statistics and top-N neighbours straight from the features, no n x n matrix "
                             "(taken by itself when the matrices would exceed MATRIX_BYTES_LIMIT)")
    parser.add_argument("--top-n", type=int, default=20)
    parser.add_argument("--min-similarity", type=float, default=0.1)
    args = parser.parse_args(argv)

    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
    total_weight = args.genre_weight + args.text_weight + args.metadata_weight
    if total_weight <= 0:
        logger.error("Error: Sum of weights must be greater than 0")
        sys.exit(1)
    try:
        features = load_features(Path(args.input_dir))
        similarities = compute_similarities(
            features=features, genre_weight=args.genre_weight, text_weight=args.text_weight,
            metadata_weight=args.metadata_weight, matrices=False if args.no_matrices else None,
            top_n=args.top_n, min_similarity=args.min_similarity,
        )
        save_similarities(similarities, Path(args.output_dir), save_to_disk=args.save_similarities)
        logger.info("✓ SIMILARITY COMPUTATION COMPLETE")
        return similarities
    except Exception as e:  # same contract as the reference script: log and exit 1
        logger.error(f"Error during similarity computation: {str(e)}", exc_info=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
