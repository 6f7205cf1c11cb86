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


def sharded_topk(x_shard: torch.Tensor, offsets, scales, shard: Shard, k: int, min_similarity: float,
                 group=None, normalize_fn=None, topk_fn=None):
    """normalise own rows -> all-gather -> top-K of own rows against all n columns.

    ``x_shard`` holds this rank's rows padded to (per_rank, d).  ``normalize_fn`` and
    ``topk_fn`` default to the HIP kernels; tests of the host logic inject CPU stand-ins.
    Returns (TopK of the rank's ``shard.rows`` rows, z_all).
    """
    if normalize_fn is None or topk_fn is None:
        from . import engine

        normalize_fn = normalize_fn or engine.normalize_blocks
        topk_fn = topk_fn or engine.cosine_topk
    if x_shard.shape[0] != shard.per_rank:
        raise ValueError("x_shard must be padded to per_rank rows")
    z_shard = normalize_fn(x_shard, offsets, scales)
    z_all = all_gather_rows(z_shard, shard, group)
    top = topk_fn(z_all, k, min_similarity, q_row0=shard.row0, nq=shard.rows, n=shard.n)
    return top, z_all
