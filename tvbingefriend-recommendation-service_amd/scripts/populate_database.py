"""MI355X counterpart of ``compute_and_store_similarities`` of the reference's
scripts/populate_database.py:85-259 -- the production top-K path.

The reference walks the shows one at a time: five one-row cosine similarities,
the weighted sum, a full argsort and a Python loop per show (:170-218).  Here
the whole catalogue goes through three kernel launches (normalise, fused
similarity + top-K, per-block scores) and only the K kept neighbours per show
come back to the host, where they are turned into the same
``{show_id: [record, ...]}`` dicts and handed to the same sink in the same
batches of 5000 shows (:166, :224-234).
"""

from __future__ import annotations

import logging
from pathlib import Path

import numpy as np

from ..ml.similarity_computer import SimilarityComputer, TopKResult
from .compute_similarities import load_features

logger = logging.getLogger(__name__)

FLUSH_EVERY = 5000  # shows per bulk_store_all_similarities call (populate_database.py:166)


_records_ext = None  # the C builder of the record dicts: module, False (not available) or None (not looked for yet)


def _load_records_ext():
    """csrc/_tvbf_records.so (built by native.build_records_ext from csrc/tvbf_records.c), or False.
    Host-side formatting only: without it the Python loop below builds the same objects."""
    global _records_ext
    if _records_ext is None:
        import importlib.util
        from .. import native

        _records_ext = False
        path = native.RECORDS_EXT_PATH
        if path.exists():
            try:
                spec = importlib.util.spec_from_file_location("_tvbf_records", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                _records_ext = mod
            except (ImportError, OSError) as e:  # another Python version, a stale build
                logger.warning("record builder %s not loadable (%s): using the Python loop", path, e)
    return _records_ext


def neighbour_records(result: TopKResult, show_ids, start: int, stop: int, use_ext: bool = True) -> dict:
    """Rows [start, stop) of a top-K result as the reference's record dicts
    (populate_database.py:210-221): Python ints/floats, shows without neighbours omitted.
    Built in C when the extension is there and the show ids are integers (about a third of the
    time); otherwise the arrays are turned into nested lists once (one C loop each) so that the
    Python loop only builds the dicts -- element-wise indexing of the arrays cost a third more."""
    out = {}
    if stop <= start:
        return out
    ext = _load_records_ext() if use_ext else False
    if ext:
        try:
            ids = np.asarray(show_ids)
        except (TypeError, ValueError, OverflowError):
            ids = None
        if ids is not None and ids.ndim == 1 and ids.dtype.kind in "iu" and ids.dtype.itemsize <= 8 and (
                ids.dtype.kind == "i" or ids.size == 0 or int(ids.max()) <= np.iinfo(np.int64).max):
            ids = np.ascontiguousarray(ids, dtype=np.int64)
            k = int(np.asarray(result.idx).shape[1])
            return ext.records(ids, np.ascontiguousarray(result.idx, dtype=np.int32),
                               np.ascontiguousarray(result.score, dtype=np.float64),
                               np.ascontiguousarray(result.comp, dtype=np.float64),
                               np.ascontiguousarray(result.count, dtype=np.int32), k, int(start), int(stop))
    idx = np.asarray(result.idx[start:stop])
    counts = np.asarray(result.count[start:stop]).tolist()
    similar = np.asarray(show_ids)[np.where(idx < 0, 0, idx)].tolist()  # -1 padding is never read
    score = np.asarray(result.score[start:stop], dtype=np.float64).tolist()
    comp = np.asarray(result.comp[start:stop], dtype=np.float64)
    genre, text, meta = comp[:, :, 0].tolist(), comp[:, :, 1].tolist(), comp[:, :, 2].tolist()
    for r, c in enumerate(counts):
        if c == 0:
            continue
        out[show_ids[start + r]] = [
            {
                "similar_show_id": sid,
                "similarity_score": h,
                "genre_score": g,
                "text_score": t,
                "metadata_score": m,
            }
            for sid, h, g, t, m in zip(similar[r][:c], score[r][:c], genre[r][:c], text[r][:c], meta[r][:c])
        ]
    return out


# column order of the reference's show_similarities table (models/show_similarity.py:19-29)
TABLE_COLUMNS = ("show_id", "similar_show_id", "similarity_score", "genre_score", "text_score",
                 "metadata_score")


def neighbour_table(result: TopKResult, show_ids) -> dict:
    """All kept pairs as six flat columns, in (show, rank) order -- the rows the reference
    builds one ORM object at a time (repos/similarity_repository.py:97-108), here as arrays
    for ``executemany`` / bulk loaders.  No Python object per pair is created."""
    ids = np.asarray(show_ids)
    k = result.idx.shape[1]
    count = np.asarray(result.count)
    # flat masks over the (show, rank) grid, row-major: shows ascending, ranks ascending
    keep = (np.arange(k)[None, :] < count[:, None]).reshape(-1)
    whole = bool(keep.all())  # every list full: the grid IS the table

    def kept(a, width=None):
        flat = np.asarray(a).reshape(-1) if width is None else np.asarray(a).reshape(-1, width)
        return flat if whole else flat[keep]

    comp = kept(result.comp, 3)
    return {
        "show_id": np.repeat(ids[: len(count)], np.minimum(count, k)),
        "similar_show_id": ids[kept(result.idx)],
        "similarity_score": kept(result.score).astype(np.float64),
        "genre_score": comp[:, 0].astype(np.float64),
        "text_score": comp[:, 1].astype(np.float64),
        "metadata_score": comp[:, 2].astype(np.float64),
    }


def iter_table_rows(table: dict, batch_size: int = 1000):
    """Batches of plain tuples (column order ``TABLE_COLUMNS``) for a DB-API ``executemany``;
    the batch size mirrors the reference's commit interval (repos/similarity_repository.py:75)."""
    n = len(table["show_id"])
    cols = [table[c] for c in TABLE_COLUMNS]
    for start in range(0, n, batch_size):
        stop = min(n, start + batch_size)
        yield list(zip(*[c[start:stop].tolist() for c in cols]))


def write_table_csv(table: dict, path, header: bool = True) -> int:
    """The table as a CSV file for a bulk loader (``LOAD DATA INFILE`` / ``COPY``); returns the
    number of rows.  Scores are written with 17 significant digits (exact for float64)."""
    n = len(table["show_id"])
    with open(path, "w", encoding="utf-8", newline="") as f:
        if header:
            f.write(",".join(TABLE_COLUMNS) + "\n")
        block = 65536
        for start in range(0, n, block):
            stop = min(n, start + block)
            cols = [table[c][start:stop] for c in TABLE_COLUMNS]
            lines = ["%d,%d,%.17g,%.17g,%.17g,%.17g" % row for row in zip(*[c.tolist() for c in cols])]
            f.write("\n".join(lines) + "\n")
    return n


def compute_top_neighbours(features: dict, genre_weight: float = 0.4, text_weight: float = 0.5,
                           metadata_weight: float = 0.1, top_n_per_show: int = 20,
                           min_similarity: float = 0.1, device=None, group=None) -> TopKResult:
    """Columnar result of the production path (metadata = mean of three cosines,
    raw weights: populate_database.py:184-192) for array-native sinks.  ``group``: the ranks of a
    row-sharded job (see SimilarityComputer.topk); every rank returns the whole result."""
    sc = SimilarityComputer(genre_weight=genre_weight, text_weight=text_weight,
                            metadata_weight=metadata_weight, device=device)
    return sc.topk(features, k=top_n_per_show, min_similarity=min_similarity, variant="populate", group=group)


RANK_WORKER = ("import tvbf_amd; from tvbingefriend_recommendation_service_amd.scripts.populate_database "
               "import _rank_main; _rank_main()")


def _rank_main() -> None:
    """Body of one rank process started by :func:`compute_top_neighbours_on_gpus`: joins the process
    group, loads the feature files, takes part in the sharded computation; rank 0 leaves the
    result in the file named by TVBF_RESULT."""
    import json
    import os

    import torch
    import torch.distributed as dist

    cfg = json.loads(os.environ["TVBF_JOB"])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device(f"cuda:{int(os.environ['LOCAL_RANK']) % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    if cfg["backend"] == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    try:
        features = load_features(Path(cfg["input_dir"]))
        result = compute_top_neighbours(features, cfg["genre_weight"], cfg["text_weight"], cfg["metadata_weight"],
                                        cfg["top_n_per_show"], cfg["min_similarity"], device=dev, group=True)
        if rank == 0:
            np.savez(os.environ["TVBF_RESULT"], idx=result.idx, score=result.score, count=result.count,
                     comp=result.comp)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def compute_top_neighbours_on_gpus(input_dir: Path, gpus: int, genre_weight: float = 0.4, text_weight: float = 0.5,
                                   metadata_weight: float = 0.1, top_n_per_show: int = 20,
                                   min_similarity: float = 0.1, backend: str = "nccl", timeout: float = 3600.0) -> TopKResult:
    """The production path row-sharded over ``gpus`` GPUs of this node: starts one fresh process per
    GPU (rank r on GPU r, rendezvous on 127.0.0.1 -- children of this process, which itself never
    touches a GPU for it), every rank reads the feature files, builds its shard of the normalised
    matrix, takes part in one all-gather and computes the lists of its rows; rank 0 hands the gathered
    result back through a temporary file.  ``backend`` "gloo" lets several ranks share one GPU
    (rehearsal: the exchange goes through host memory)."""
    import json
    import os
    import socket
    import subprocess
    import sys
    import tempfile
    import time

    if gpus < 2:
        raise ValueError("gpus must be at least 2 (the single-GPU path needs no ranks)")
    root = str(Path(__file__).resolve().parents[2])
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    job = json.dumps({"input_dir": str(input_dir), "genre_weight": genre_weight, "text_weight": text_weight,
                      "metadata_weight": metadata_weight, "top_n_per_show": top_n_per_show,
                      "min_similarity": min_similarity, "backend": backend})
    with tempfile.TemporaryDirectory() as tmp:
        result_path = str(Path(tmp) / "result.npz")
        procs = []
        for r in range(gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TVBF_JOB=job, TVBF_RESULT=result_path,
                       PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
            procs.append(subprocess.Popen([sys.executable, "-c", RANK_WORKER], env=env))
        deadline = time.time() + timeout
        failed = None
        while any(p.poll() is None for p in procs) and failed is None:
            failed = next((p.returncode for p in procs if p.poll() not in (None, 0)), None)
            if failed is None and time.time() > deadline:
                failed = -1
            time.sleep(0.05)
        if failed is not None:  # one rank failed (or time ran out): end the others, by their own PIDs
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            raise RuntimeError(f"a rank of the sharded job failed (exit code {failed})")
        with np.load(result_path) as r:
            return TopKResult(r["idx"], r["score"], r["count"], r["comp"])


def compute_and_store_similarities(input_dir: Path, genre_weight: float = 0.4, text_weight: float = 0.5,
                                   metadata_weight: float = 0.1, top_n_per_show: int = 20,
                                   min_similarity: float = 0.1, repo=None, clear_existing=None,
                                   gpus: int | None = None, backend: str = "nccl") -> dict:
    """Compute the top-N neighbours of every show and store them.

    ``gpus`` > 1 shards the shows over that many GPUs of this node.  Inside a launcher's rank
    (torch.distributed initialised by the caller) the given ranks are used: every rank calls this
    function, rank 0 alone feeds the sink, the others return the statistics without storing.
    Otherwise fresh rank processes are started for the computation
    (:func:`compute_top_neighbours_on_gpus`) and this process feeds the sink.

    A sink that also has ``store_similarity_table(table: dict) -> int`` receives the six flat
    columns of :func:`neighbour_table` once instead of record dicts in batches.

    ``repo`` is the sink: any object with ``bulk_store_all_similarities(dict,
    clear_existing=False) -> int`` and ``get_similarity_stats() -> dict`` (the
    reference's ``SimilarityRepository``, repos/similarity_repository.py:72,243).
    ``clear_existing`` is called once before the first write (the reference
    deletes all rows of show_similarities there, populate_database.py:160-161).
    When ``repo`` is None the reference's own repository and session are used,
    which needs the reference package and its database to be importable.
    """
    import pandas as pd

    input_dir = Path(input_dir)
    logger.info("COMPUTING AND STORING SIMILARITIES")
    logger.info(f"Genre weight: {genre_weight}  Text weight: {text_weight}  Metadata weight: {metadata_weight}")
    logger.info(f"Top N per show: {top_n_per_show}  Min similarity: {min_similarity}")

    features = load_features(input_dir)
    show_ids = pd.read_csv(input_dir / "shows_metadata.csv")["id"].tolist()
    n = len(show_ids)
    if features["genre_features"].shape[0] != n:
        raise ValueError("shows_metadata.csv and the feature files disagree on the number of shows")

    in_group = False
    if gpus is not None and gpus > 1:
        import torch.distributed as dist

        in_group = dist.is_available() and dist.is_initialized()
        if in_group and dist.get_world_size() != gpus:
            raise ValueError(f"gpus={gpus} but the process group has {dist.get_world_size()} ranks")
    store = not in_group or dist.get_rank() == 0  # of a launcher's ranks only rank 0 touches the sink
    db = None
    if repo is None and store:  # drop-in inside the reference's own process
        from tvbingefriend_recommendation_service.models import ShowSimilarity
        from tvbingefriend_recommendation_service.models.database import SessionLocal
        from tvbingefriend_recommendation_service.repos import SimilarityRepository

        db = SessionLocal()
        repo = SimilarityRepository(db)
        if clear_existing is None:
            def clear_existing():
                db.query(ShowSimilarity).delete()
                db.commit()
    try:
        if clear_existing is not None and store:
            logger.info("Clearing existing similarities from database...")
            clear_existing()

        if gpus is not None and gpus > 1:
            if in_group:
                result = compute_top_neighbours(features, genre_weight, text_weight, metadata_weight,
                                                top_n_per_show, min_similarity, group=True)
            else:
                result = compute_top_neighbours_on_gpus(input_dir, gpus, genre_weight, text_weight, metadata_weight,
                                                        top_n_per_show, min_similarity, backend=backend)
        else:
            result = compute_top_neighbours(features, genre_weight, text_weight, metadata_weight,
                                            top_n_per_show, min_similarity)
        total_records_stored = 0
        if not store:  # a rank other than 0 of a launcher's job: rank 0 stores
            return {"total_records": int(np.minimum(result.count, top_n_per_show).sum()),
                    "top_n_per_show": top_n_per_show, "min_similarity": min_similarity}
        if hasattr(repo, "store_similarity_table"):  # array-native sink: no per-pair Python objects
            total_records_stored = int(repo.store_similarity_table(neighbour_table(result, show_ids)))
        else:
            ids_once = np.asarray(show_ids)  # integer ids as one array: the record builder takes it as it is
            if ids_once.ndim != 1 or ids_once.dtype.kind not in "iu":
                ids_once = show_ids
            for start in range(0, n, FLUSH_EVERY):
                stop = min(n, start + FLUSH_EVERY)
                batch = neighbour_records(result, ids_once, start, stop)
                if batch:
                    total_records_stored += repo.bulk_store_all_similarities(batch, clear_existing=False)
                logger.info(f"  Processed {stop}/{n} shows... (stored {total_records_stored} similarity records)")
        logger.info(f"✓ Stored {total_records_stored} total similarity records")

        stats = repo.get_similarity_stats()
        stats["total_records"] = total_records_stored
        stats["top_n_per_show"] = top_n_per_show
        stats["min_similarity"] = min_similarity
        return stats
    finally:
        if db is not None:
            db.close()


class ColumnarSimilaritySink:
    """Array-native sink: keeps the six columns of :func:`neighbour_table`; ``rows()`` and
    ``to_csv()`` feed an ``executemany`` or a bulk loader."""

    def __init__(self):
        self.table = {c: np.empty(0) for c in TABLE_COLUMNS}

    def store_similarity_table(self, table: dict) -> int:
        self.table = table
        return len(table["show_id"])

    def rows(self, batch_size: int = 1000):
        return iter_table_rows(self.table, batch_size)

    def to_csv(self, path, header: bool = True) -> int:
        return write_table_csv(self.table, path, header)

    def get_similarity_stats(self) -> dict:
        total = len(self.table["show_id"])
        shows = len(np.unique(self.table["show_id"])) if total else 0
        return {"total_records": total, "unique_shows": shows,
                "avg_similarities_per_show": (total / shows) if shows else 0, "last_computed": None}


class MemorySimilaritySink:
    """In-memory stand-in for ``SimilarityRepository`` with the two methods the
    path calls; used by the plumbing test and handy for dry runs."""

    def __init__(self):
        self.records: dict = {}
        self.calls = 0

    def clear(self):
        self.records.clear()

    def bulk_store_all_similarities(self, all_similarities: dict, batch_size: int = 1000,
                                    clear_existing: bool = True) -> int:
        if clear_existing:
            self.records.clear()
        self.calls += 1
        total = 0
        for show_id, items in all_similarities.items():
            self.records[show_id] = list(items)
            total += len(items)
        return total

    def get_similarity_stats(self) -> dict:
        total = sum(len(v) for v in self.records.values())
        shows = len(self.records)
        # same keys as repos/similarity_repository.py:258-263 of the reference
        return {"total_records": total, "unique_shows": shows,
                "avg_similarities_per_show": (total / shows) if shows else 0, "last_computed": None}
