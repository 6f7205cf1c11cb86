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
    return sharded_topk_z(z_shard, shard, k, min_similarity, group=group, session_fn=session_fn, collective=collective)


def sharded_topk_z(z_shard: torch.Tensor, shard: Shard, k: int, min_similarity: float, group=None, session_fn=None,
                   collective=None, z_all: torch.Tensor | None = None):
    """:func:`sharded_topk` from the rank's normalised rows ``z_shard`` ((per_rank, ld), padding
    rows zero) on: all-gather, own columns meanwhile, the rest, merge.  ``z_all`` ((padded_total, ld),
    rank-major, padding rows zero): the normalised rows of ALL shows when the rank already holds them
    (it built them itself from features that are resident on every rank) -- no all-gather then."""
    if session_fn is None:
        from . import engine

        session_fn = engine.TopkSession
    if z_shard.shape[0] != shard.per_rank:
        raise ValueError("z_shard must be padded to per_rank rows")
    if collective is None:
        collective = shard.world > 1
    if not collective:
        if shard.world != 1:
            raise ValueError("a sharded job needs the collective")
        session = session_fn(z_shard, shard.rows, 0, k, min_similarity)
        session.accumulate((z_shard, shard.rows, 0))
        return session.finish(), z_shard
    if z_all is not None:
        if z_all.shape != (shard.padded_total, z_shard.shape[1]):
            raise ValueError("z_all must hold per_rank * world rows")
        own, after, before = column_pieces(shard)
        session = session_fn(z_shard, shard.rows, shard.row0, k, min_similarity)
        session.accumulate((z_shard, own[1], own[0]))
        session.accumulate((z_all[after[0]:], after[1], after[0]), (z_all, before[1], before[0]))
        return session.finish(), z_all
    z_all = torch.empty((shard.padded_total, z_shard.shape[1]), dtype=z_shard.dtype, device=z_shard.device)
    if _via_host(z_shard, group):  # gloo ranks that use GPUs: through host memory, synchronously
        host = z_shard.cpu()
        parts = [torch.empty_like(host) for _ in range(shard.world)]
        dist.all_gather(parts, host, group=group)
        z_all.copy_(torch.cat(parts, dim=0))
        work = None
    elif z_shard.is_cuda:
        work = dist.all_gather_into_tensor(z_all, z_shard, group=group, async_op=True)
    else:  # gloo: CPU tensors only
        work = dist.all_gather(list(z_all.chunk(shard.world, dim=0)), z_shard, group=group, async_op=True)
    own, after, before = column_pieces(shard)
    session = session_fn(z_shard, shard.rows, shard.row0, k, min_similarity)
    session.accumulate((z_shard, own[1], own[0]))            # runs while the shards travel
    if work is not None:
        work.wait()                                          # the current stream now waits for the gather
    session.accumulate((z_all[after[0]:], after[1], after[0]), (z_all, before[1], before[0]))
    return session.finish(), z_all


# ---------------------------------------------------------------------------
# Symmetric sharded sweep: of two shards only one computes the block between them.
# ---------------------------------------------------------------------------
PAD_BOUND = 0xFF800000FFFFFFFF - (1 << 64)  # int64 view of the largest key (+inf): a padding row accepts nothing


def half_peers(world: int, rank: int) -> list[int]:
    """The shards whose block with shard ``rank`` is computed BY ``rank`` in full: the
    (world - 1) // 2 shards that follow it cyclically.  For an even world the block with the shard
    at distance world / 2 (the partner) is split between the two, see partner_split()."""
    return [(rank + 1 + i) % world for i in range((world - 1) // 2)]


def partner(world: int, rank: int):
    """The shard at distance world / 2 for an even world, else None."""
    return (rank + world // 2) % world if world % 2 == 0 else None


def partner_split(shard: Shard, upper_rank: int) -> int:
    """Rows [0, h0) of the UPPER partner's shard are columns of the lower partner's sweep; its rows
    [h0, rows) are query rows of a second sweep of the upper partner against the lower partner's
    columns.  h0 is half the shard, rounded up to whole 128-row blocks."""
    rows = shard_bounds(shard.n, shard.world, upper_rank).rows
    return min(rows, ((rows + 1) // 2 + 127) // 128 * 128)


def half_pieces(shard: Shard):
    """Column ranges of the main sweep's column side as at most two (first_row, n_rows) pieces of
    the gathered matrix: the shards of half_peers() and, for the lower partner of an even world,
    the first partner_split() rows of its partner (cyclically contiguous); padding rows left out."""
    W, r, P, n = shard.world, shard.rank, shard.per_rank, shard.n
    peers = half_peers(W, r)
    tail = 0  # rows of the partner that follow the full peers
    pt = partner(W, r)
    if pt is not None and r < W // 2:
        tail = partner_split(shard, pt)
    if not peers and not tail:
        return (0, 0), (0, 0)
    first = (r + 1) % W
    end_shard = first + len(peers)  # exclusive, not wrapped

    def span(a_row, b_row):
        a_row, b_row = min(a_row, n), min(b_row, n)
        return (a_row, max(0, b_row - a_row))

    if end_shard <= W:  # no wrap (a lower partner never wraps: its partner comes right after the peers)
        return span(first * P, end_shard * P + tail), (0, 0)
    return span(first * P, W * P), span(0, (end_shard - W) * P)


class SymmetricShardJob:
    """One rank's side of the symmetric sharded sweep, as phases between which the caller runs the
    collectives (sharded_topk_symmetric does; a test can drive several simulated ranks in lockstep):

        phase_own()            own rows x own columns (overlaps the all-gather of Z)
        own_bounds()           -> (per_rank,) int64 keys to all-gather
        phase_half(z_all, b)   own rows x the columns of half_pieces(), rows AND columns fed; the
                               upper partner of an even world also sweeps its rows [h0, rows)
                               against its partner's columns
        lists_for_peers(b)     -> (world, per_rank, k) int64: per peer the lists of ITS rows found here
        flag                   -> overflow words, to be max-reduced over the ranks
        finish(z_all, recv, flag) gated plain re-run + merge of own lists and the received ones
    """

    def __init__(self, z_shard: torch.Tensor, shard: Shard, k: int, min_similarity: float, *, col_cap: int = 1024,
                 slots: int = 96, workspace: torch.Tensor | None = None, colside=None, own_matrix=None):
        from . import engine

        self.engine, self.shard, self.k, self.min_similarity = engine, shard, int(k), float(min_similarity)
        self.z_shard = z_shard
        self.dev = z_shard.device
        self.session = engine.TopkSession(z_shard, shard.rows, shard.row0, k, min_similarity, slots=slots,
                                          workspace=workspace)
        self.side = colside if colside is not None else engine.ColumnSide(shard.padded_total, self.dev, cap=col_cap)
        if colside is not None:
            self.side.begin()
        # scores of the own block as a stored matrix (engine.own_matrix_bytes() bytes): a caller that
        # keeps such a buffer passes it in; None = the fused sweeps
        self.own_matrix = own_matrix
        self.normal_slots = 0
        self.per_sweep = _sweep_slots(shard.world, k)
        if self.per_sweep < 6:
            raise ValueError("k too large for the symmetric sharded route with this many ranks: use sharded_topk")
        # upper partner: second sweep over the rows [h0, rows) of this shard
        self.sub, self.h0 = None, 0
        pt = partner(shard.world, shard.rank)
        if pt is not None and shard.rank >= shard.world // 2:
            self.h0 = partner_split(shard, shard.rank)
            if shard.rows - self.h0 > 0:
                self.sub = engine.TopkSession(z_shard[self.h0:], shard.rows - self.h0, shard.row0 + self.h0, k,
                                              min_similarity, slots=16)

    def phase_own(self):
        own, _, _ = column_pieces(self.shard)
        need = self.engine.own_matrix_bytes(self.shard.rows, self.z_shard.shape[1]) if self.own_matrix is not None else 0
        if need and self.own_matrix.numel() * self.own_matrix.element_size() >= need:
            self.session.accumulate_own_matrix(self.own_matrix)  # a block of a few ten thousand rows: stored matrix
        elif self.shard.rows >= 2048:   # large enough for the symmetric sweep inside the block
            self.session.accumulate_own_symmetric(self.side, max_slots=self.per_sweep)
        else:
            self.session.accumulate((self.z_shard, own[1], own[0]), max_slots=self.per_sweep)

    def own_bounds(self) -> torch.Tensor:
        b = torch.full((self.shard.per_rank,), PAD_BOUND, dtype=torch.int64, device=self.dev)
        if self.shard.rows:
            b[: self.shard.rows] = self.session.bounds()
        return b

    def phase_half(self, z_all: torch.Tensor, bounds_all: torch.Tensor):
        a, b = half_pieces(self.shard)
        self.session.accumulate((z_all[a[0]:], a[1], a[0]), (z_all[b[0]:], b[1], b[0]), max_slots=self.per_sweep,
                                colside=(self.side, bounds_all))
        self.normal_slots = self.session.used
        if self.sub is not None:
            sh = self.shard
            pt = partner(sh.world, sh.rank)
            c0 = min(pt * sh.per_rank, sh.n)
            cols = min((pt + 1) * sh.per_rank, sh.n) - c0
            self.sub.bounds().copy_(self.session.bounds()[self.h0:])  # start from what the own block found
            self.sub.accumulate((z_all[c0:], cols, c0), max_slots=self.per_sweep, colside=(self.side, bounds_all))

    def lists_for_peers(self, bounds_all: torch.Tensor) -> torch.Tensor:
        sh = self.shard
        send = torch.zeros((sh.world, sh.per_rank, self.k), dtype=torch.int64, device=self.dev)  # zero = empty slot
        if sh.rows == 0:
            return send
        # the fed shards follow this one cyclically (the partner last), and a show's global index
        # is also its row in the (world * per_rank)-row view of `send`: one reduction per run of shards
        fed = len(half_peers(sh.world, sh.rank)) + (1 if partner(sh.world, sh.rank) is not None else 0)
        flat = send.view(sh.world * sh.per_rank, self.k)
        first = sh.rank + 1
        for p0, p1 in ((first, min(first + fed, sh.world)), (0, first + fed - sh.world)):
            a, b = min(p0 * sh.per_rank, sh.n), min(p1 * sh.per_rank, sh.n)
            if p1 > p0 and b > a:
                self.side.reduce(bounds_all, a, b - a, self.k, self.min_similarity, flat[a:b])
        return send

    @property
    def tiles(self) -> int:
        """128 x 128 score tiles this rank's sweeps compute (the gated re-run not counted)."""
        return self.session.tiles + (self.sub.tiles if self.sub is not None else 0)

    @property
    def flag(self) -> torch.Tensor:
        """Overflow words of this rank's column side (summary + one per show), to be max-reduced."""
        return self.side.flags

    def finish(self, z_all: torch.Tensor, recv: torch.Tensor, flag: torch.Tensor, out=None):
        """``flag``: the overflow words after the max-reduction (int32 device tensor).  The plain
        sweep over all columns is launched in any case; it only touches row blocks that hold a
        flagged show and returns at once when there is none."""
        sh = self.shard
        own, after, before = column_pieces(sh)
        first = self.session.used
        self.session.accumulate((self.z_shard, own[1], own[0]), gate=flag)
        self.session.accumulate((z_all[after[0]:], after[1], after[0]), (z_all, before[1], before[0]), gate=flag)
        extra = recv
        if self.sub is not None and self.sub.used:
            # the second sweep's lists, placed at the rows they belong to
            mine = torch.zeros((self.sub.used, sh.per_rank, self.k), dtype=torch.int64, device=self.dev)
            mine[:, self.h0 : sh.rows] = self.sub.slot_lists()
            extra = torch.cat([recv, mine], dim=0)
        return self.session.finish(out, extra=extra, gate=flag, alt=(first, self.session.used - first),
                                   slots=self.normal_slots)


def symmetric_route_fits(world: int, k: int, ld: int) -> bool:
    """Whether sharded_topk_symmetric takes the shape: LDS-DMA rows and a merge wide enough for the
    lists of three sweeps (own block, half sweep, second sweep of an upper partner; at least six
    each) plus the `world` received ones."""
    return ld % 64 == 0 and _sweep_slots(world, k) >= 6


def _sweep_slots(world: int, k: int) -> int:
    """List slots each of a rank's three sweeps may use so that one merge takes them all."""
    from . import native

    return min(16, (int(native.load().tvbf_topk_max_slots(int(k))) - world) // 3)


def _via_host(t: torch.Tensor, group) -> bool:
    return t.is_cuda and dist.get_backend(group) == "gloo"  # gloo: CPU tensors only


def gather_words(local: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """all-gather of equally sized 1-D tensors, rank-major."""
    if _via_host(local, group):
        host = local.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return torch.cat(parts).to(local.device)
    out = torch.empty((world * local.numel(),), dtype=local.dtype, device=local.device)
    if local.is_cuda:
        dist.all_gather_into_tensor(out, local, group=group)
    else:
        dist.all_gather(list(out.chunk(world)), local, group=group)
    return out


def exchange_lists(send: torch.Tensor, group=None) -> torch.Tensor:
    """all-to-all of (world, ...) tensors: slice p goes to rank p; returns what the ranks sent here.
    RCCL runs one all_to_all_single; gloo has no all-to-all, so CPU tensors (and GPU tensors of a
    gloo group, through host memory) are all-gathered whole and the slices for this rank picked."""
    if send.is_cuda and not _via_host(send, group):
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=group)
        return recv
    host = send.cpu() if send.is_cuda else send
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host.contiguous(), group=group)
    return torch.stack([parts[p][rank] for p in range(world)]).to(send.device)


def pack_lists_and_flags(send: torch.Tensor, flags: torch.Tensor, shard: Shard) -> torch.Tensor:
    """One all-to-all payload per peer: the K-lists found here for the peer's shows followed by the
    overflow words of those shows (``flags``: the [4 + padded_total] int32 words of a column side).
    (world, per_rank, k) int64 + int32 words -> (world, per_rank * k + ceil(per_rank / 2)) int64."""
    W, P = shard.world, shard.per_rank
    words = flags[4 : 4 + W * P].reshape(W, P)
    if P % 2:
        words = torch.nn.functional.pad(words, (0, 1))
    return torch.cat([send.reshape(W, -1), words.contiguous().view(torch.int64)], dim=1)


def unpack_lists_and_flags(recv: torch.Tensor, shard: Shard, k: int):
    """What the peers sent here (pack_lists_and_flags, after the all-to-all): the lists as a
    (world, per_rank, k) view and the overflow words of THIS rank's shows, max-reduced over the
    senders, placed in a [4 + padded_total] int32 tensor ([0] = some show of this rank is flagged).
    A rank re-runs only its own rows, so it needs no other rank's words."""
    W, P = shard.world, shard.per_rank
    lists = recv[:, : P * k].unflatten(1, (P, k))
    words = recv[:, P * k :].contiguous().view(torch.int32)[:, :P].max(dim=0).values
    flag = torch.zeros(4 + W * P, dtype=torch.int32, device=recv.device)
    flag[4 + shard.rank * P : 4 + (shard.rank + 1) * P] = words
    flag[0] = words.max() if P else 0
    return lists, flag


def reduce_flag(flag: torch.Tensor, group=None) -> torch.Tensor:
    """element-wise max over the ranks of a small tensor (a copy; the input is left alone)."""
    if _via_host(flag, group):
        host = flag.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        return host.to(flag.device)
    out = flag.clone()
    dist.all_reduce(out, op=dist.ReduceOp.MAX, group=group)
    return out


def sharded_topk_symmetric(x_shard: torch.Tensor, offsets, scales, shard: Shard, k: int, min_similarity: float,
                           group=None, normalize_fn=None, job_fn=None):
    """sharded_topk with every block between two shards computed once: normalise own rows ->
    all-gather of Z (asynchronous) -> own block -> all-gather of the per-row bounds -> blocks of the
    cyclically following half of the shards, rows and columns fed -> one all-to-all of the column-side
    lists (world x per_rank x k keys) with the overflow words of the receiver's shows -> merge.  Same lists as sharded_topk, bit for bit.
    Needs a row length that is a multiple of 64 (the engine raises otherwise) and a K for which one
    merge can take the rank's own lists plus one per rank: symmetric_route_fits()."""
    if normalize_fn is None:
        from . import engine

        normalize_fn = engine.normalize_blocks
    if x_shard.shape[0] != shard.per_rank:
        raise ValueError("x_shard must be padded to per_rank rows")
    z_shard = normalize_fn(x_shard, offsets, scales)
    return sharded_topk_symmetric_z(z_shard, shard, k, min_similarity, group=group, job_fn=job_fn)


def sharded_topk_symmetric_z(z_shard: torch.Tensor, shard: Shard, k: int, min_similarity: float, group=None,
                             job_fn=None, z_all: torch.Tensor | None = None):
    """:func:`sharded_topk_symmetric` from the rank's normalised rows ``z_shard`` ((per_rank, ld),
    padding rows zero) on.  ``z_all``: as in :func:`sharded_topk_z` -- the Z all-gather is left out,
    the exchanges of the symmetric route (bound keys, column-side lists) stay."""
    job_fn = job_fn or SymmetricShardJob
    if z_shard.shape[0] != shard.per_rank:
        raise ValueError("z_shard must be padded to per_rank rows")
    if z_all is not None:
        if z_all.shape != (shard.padded_total, z_shard.shape[1]):
            raise ValueError("z_all must hold per_rank * world rows")
        return _symmetric_exchange(job_fn, z_shard, z_all, None, shard, k, min_similarity, group)
    z_all = torch.empty((shard.padded_total, z_shard.shape[1]), dtype=z_shard.dtype, device=z_shard.device)
    host_route = _via_host(z_shard, group)
    if host_route:
        host = z_shard.cpu()
        parts = [torch.empty_like(host) for _ in range(shard.world)]
        dist.all_gather(parts, host, group=group)
        z_all.copy_(torch.cat(parts, dim=0))
        work = None
    elif z_shard.is_cuda:
        work = dist.all_gather_into_tensor(z_all, z_shard, group=group, async_op=True)
    else:
        work = dist.all_gather(list(z_all.chunk(shard.world, dim=0)), z_shard, group=group, async_op=True)
    return _symmetric_exchange(job_fn, z_shard, z_all, work, shard, k, min_similarity, group)


def _symmetric_exchange(job_fn, z_shard, z_all, work, shard: Shard, k: int, min_similarity: float, group):
    """The phases of the symmetric sharded sweep around their collectives; ``work``: the handle of
    the Z all-gather in flight (None: z_all is complete)."""
    if job_fn is SymmetricShardJob and z_shard.is_cuda:
        from . import engine

        mbytes = engine.own_matrix_bytes(shard.rows, z_shard.shape[1])
        if mbytes > engine.MEMORY_HEADROOM * engine.available_bytes(z_shard.device):
            mbytes = 0  # not enough memory for the stored own block: the fused own-block sweeps give the same lists
        own_matrix = torch.empty((mbytes + 7) // 8, dtype=torch.int64, device=z_shard.device) if mbytes else None
        job = job_fn(z_shard, shard, k, min_similarity, own_matrix=own_matrix)
    else:
        job = job_fn(z_shard, shard, k, min_similarity)
    job.phase_own()                                       # runs while the shards travel
    bounds_all = gather_words(job.own_bounds(), shard.world, group)
    if work is not None:
        work.wait()
    job.phase_half(z_all, bounds_all)
    # one all-to-all carries the column-side lists and the overflow words of the receiver's shows
    packed = pack_lists_and_flags(job.lists_for_peers(bounds_all), job.flag, shard)
    lists, flag = unpack_lists_and_flags(exchange_lists(packed, group), shard, k)
    return job.finish(z_all, lists, flag), z_all
