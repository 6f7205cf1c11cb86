"""Row sharding of the similarity path over the ranks of one node.

Rank r owns query rows [r*P, min((r+1)*P, n)) with P = ceil(n / world).  Every
rank normalises its own shard, one all-gather (RCCL over xGMI when the
tensors live on GPUs) gives every rank the full normalised matrix, and each
rank then computes the top-K lists of its own rows against all n columns.  No
reduction is needed: lists belong to rows and rows are disjoint.  Results do
not depend on the world size because every score is the same fixed-order fp32
sum and the selection applies one total order.
"""

from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    n: int          # global number of shows
    per_rank: int   # P: rows every rank contributes to the all-gather (last ranks zero-padded)
    row0: int       # first global row owned by this rank
    rows: int       # real rows owned by this rank (may be 0 on trailing ranks)

    @property
    def padded_total(self) -> int:
        return self.per_rank * self.world


def shard_bounds(n: int, world: int, rank: int) -> Shard:
    if n < 1 or world < 1 or not 0 <= rank < world:
        raise ValueError("bad n / world / rank")
    per = (n + world - 1) // world
    row0 = min(rank * per, n)
    rows = max(0, min(n, (rank + 1) * per) - row0)
    return Shard(rank, world, n, per, row0, rows)


def all_gather_rows(z_shard: torch.Tensor, shard: Shard, group=None) -> torch.Tensor:
    """Gather the (per_rank, ld) blocks of all ranks into one (per_rank*world, ld)
    tensor, rank-major; rows >= n are padding.  One collective."""
    if z_shard.dim() != 2 or z_shard.shape[0] != shard.per_rank or not z_shard.is_contiguous():
        raise ValueError(f"z_shard must be a contiguous ({shard.per_rank}, ld) tensor")
    if shard.world == 1:
        return z_shard
    out = torch.empty((shard.padded_total, z_shard.shape[1]), dtype=z_shard.dtype, device=z_shard.device)
    if z_shard.is_cuda:
        dist.all_gather_into_tensor(out, z_shard, group=group)
    else:  # gloo: CPU tensors only
        parts = list(out.chunk(shard.world, dim=0))
        dist.all_gather(parts, z_shard, group=group)
    return out


def gather_lists(local: torch.Tensor, shard: Shard, group=None) -> torch.Tensor:
    """Concatenate per-rank outputs (rows, ...) rank-major into (n, ...) on every rank.
    GPU tensors travel over the process group's own backend; with gloo pass CPU tensors."""
    if shard.world == 1:
        return local
    pad = torch.zeros((shard.per_rank,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(shard.world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat(parts, dim=0)[: shard.n]


def column_pieces(shard: Shard):
    """Column ranges a rank scores, in order: its own shard first (needs no communication), then
    the columns after and before it (available once the all-gather has completed).
    Each piece is (first_row_in_gathered_matrix, n_rows); rows of the gathered matrix are global
    column indices, padding rows (>= n) are left out."""
    own = (shard.row0, shard.rows)
    after = (shard.row0 + shard.rows, max(0, shard.n - (shard.row0 + shard.rows)))
    before = (0, shard.row0)
    return own, after, before


def sharded_topk(x_shard: torch.Tensor, offsets, scales, shard: Shard, k: int, min_similarity: float,
                 group=None, normalize_fn=None, session_fn=None, collective=None):
    """normalise own rows -> all-gather (asynchronous) -> top-K of own rows against all n columns.

    The rank first scores its rows against its own shard, which overlaps the collective, then
    against the remaining columns; the pieces are merged on the device.  ``x_shard`` holds this
    rank's rows padded to (per_rank, d).  ``normalize_fn(x, offsets, scales)`` and
    ``session_fn(z_q, nq, q_row0, k, min_similarity)`` default to the HIP kernels
    (engine.normalize_blocks / engine.TopkSession); tests of the host logic inject CPU stand-ins.
    ``collective`` (default: world > 1) forces the all-gather route also for a one-rank group,
    which is how the RCCL route is exercised on a single GPU.
    Returns (TopK of the rank's ``shard.rows`` rows, z_all).
    """
    if normalize_fn is None or session_fn is None:
        from . import engine

        normalize_fn = normalize_fn or engine.normalize_blocks
        session_fn = session_fn or engine.TopkSession
    if x_shard.shape[0] != shard.per_rank:
        raise ValueError("x_shard must be padded to per_rank rows")
    z_shard = normalize_fn(x_shard, offsets, scales)
    if collective is None:
        collective = shard.world > 1
    if not collective:
        if shard.world != 1:
            raise ValueError("a sharded job needs the collective")
        session = session_fn(z_shard, shard.rows, 0, k, min_similarity)
        session.accumulate((z_shard, shard.rows, 0))
        return session.finish(), z_shard
    z_all = torch.empty((shard.padded_total, z_shard.shape[1]), dtype=z_shard.dtype, device=z_shard.device)
    if z_shard.is_cuda:
        work = dist.all_gather_into_tensor(z_all, z_shard, group=group, async_op=True)
    else:  # gloo: CPU tensors only
        work = dist.all_gather(list(z_all.chunk(shard.world, dim=0)), z_shard, group=group, async_op=True)
    own, after, before = column_pieces(shard)
    session = session_fn(z_shard, shard.rows, shard.row0, k, min_similarity)
    session.accumulate((z_shard, own[1], own[0]))            # runs while the shards travel
    work.wait()                                              # the current stream now waits for the gather
    session.accumulate((z_all[after[0]:], after[1], after[0]), (z_all, before[1], before[0]))
    return session.finish(), z_all
