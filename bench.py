#!/usr/bin/env python3
"""Benchmark of the similarity + top-K hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over the whole synthetic catalogue, as the
drop-in runs it (--path exact, the default): build the normalised, weighted
matrix from the stored features (K1; when N > 1 every rank builds all rows from
the features resident on it, or with --z-exchange gather its own shard followed
by one all-gather) -> fused all-pairs similarity + per-row top-(K + slack) in fp32
(K2-K4) -> float64 re-rank of the candidates from the stored features (order,
scores, threshold cut and per-block scores of the reference's float64
arithmetic; rows that cannot be certified get a longer list or a scan).
--path fp32 stops at the fp32 lists (+ K5), the step of the earlier rounds; a
few such steps are timed after the main region and reported beside it.
Inputs (the stored features of ALL shows, as every rank of the reference's
pipeline would load them from the feature files) are resident in HBM when the
timed region starts.  Metric: ordered show-pairs per second = n^2 * steps /
time, over the whole job.  The total catalogue is fixed as the GPU count grows
(strong scaling): rank r owns rows [r*n/N, (r+1)*n/N).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FP32_MFMA_PEAK_TFLOPS = 157.3  # 256 CU x 256 flop/clk x 2.4 GHz, v_mfma_f32_32x32x2_f32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--path", choices=["exact", "fp32"], default="exact",
                    help="exact: the drop-in's step (fp32 sweep for K + slack candidates, float64 re-rank); "
                         "fp32: the fp32 lists and K5 only")
    ap.add_argument("--n", "--shows", dest="n", type=int, default=100_000, help="shows")
    ap.add_argument("--d", "--dims", dest="d", type=int, default=1024, help="feature columns (genre 32 | text d-64 | metadata 32)")
    ap.add_argument("--k", "--neighbours", dest="k", type=int, default=20)
    ap.add_argument("--data", choices=["hybrid", "dense"], default="hybrid")
    ap.add_argument("--min-similarity", type=float, default=None)
    ap.add_argument("--seed", type=int, default=30001)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather-compare", action="store_true",
                    help="several GPUs, Z built on every rank: skip the few steps of the all-gather variant timed "
                         "after the main region (reported as gather_step)")
    ap.add_argument("--no-fp32-compare", action="store_true",
                    help="skip the few fp32-only steps timed after the main region (profiling runs: one kind of step per run)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-to-host measurement (one GPU only)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo is a rehearsal mode: the shards are exchanged through host memory")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--check", action="store_true",
                    help="after the timed steps, recompute this rank's lists with the plain sweep and compare bitwise")
    ap.add_argument("--z-exchange", choices=["auto", "local", "gather"], default="auto",
                    help="several GPUs: how a rank gets the normalised rows of the other shards.  local: it builds "
                         "them itself from the stored features, which are resident on every rank (auto on the exact "
                         "path, whose re-rank needs them there anyway); gather: one all-gather of the shards "
                         "(auto on the fp32 path)")
    ap.add_argument("--sweep", choices=["auto", "plain", "symmetric"], default="auto",
                    help="single GPU: 'symmetric' computes every pair of shows once (identical lists); "
                         "auto takes it whenever the library accepts the shape")
    return ap.parse_args()


def make_shard(args, shard, dev):
    """Synthetic features of this rank's rows, generated on the GPU.

    hybrid: [genre 32 multi-hot p=0.08 | text d-64 columns, 3 % dense, non-negative, unit
    L2 rows | platform one-hot 21 + type one-hot 5 + language one-hot 6]; dense: standard normal.
    Returns (x (per_rank, d) fp32 with zero padding rows, offsets, weights).
    """
    import torch

    g = torch.Generator(device=dev)
    g.manual_seed(args.seed + shard.rank)
    rows, d = shard.rows, args.d
    x = torch.zeros((shard.per_rank, d), dtype=torch.float32, device=dev)
    if args.data == "dense":
        x[:rows] = torch.randn((rows, d), generator=g, device=dev)
        return x, [0, d], [1.0]
    dt = d - 64
    x[:rows, :32] = (torch.rand((rows, 32), generator=g, device=dev) < 0.08).float()
    text = torch.randn((rows, dt), generator=g, device=dev).abs_()
    text *= (torch.rand((rows, dt), generator=g, device=dev) < 0.03)
    text /= text.norm(dim=1, keepdim=True).clamp_min(1e-30)
    x[:rows, 32 : 32 + dt] = text
    del text
    base = 32 + dt
    r = torch.arange(rows, device=dev)
    for width in (21, 5, 6):
        pick = torch.randint(0, width, (rows,), generator=g, device=dev)
        x[r, base + pick] = 1.0
        base += width
    return x, [0, 32, 32 + dt, d], [0.4, 0.5, 0.1]


def stored_sources(engine, x, offsets):
    """The packed hybrid rows as the five arrays the reference stores (scripts/compute_features.py:115-129:
    genre int64, TF-IDF CSR float64, platform float64, type bool, language float64) on the device -- what the
    drop-in uploads from the feature files -- as sources of K1's assembly and of the float64 re-rank."""
    import torch

    g0, t0, m0 = offsets[0], offsets[1], offsets[2]
    dt = m0 - t0
    csr = x[:, t0:m0].to_sparse_csr()
    meta = x[:, m0:]
    return [engine.DenseSource(x[:, g0:t0].to(torch.int64), g0),
            engine.CsrSource(csr.crow_indices().contiguous(), csr.col_indices().contiguous(),
                             csr.values().to(torch.float64), dt, t0),
            engine.DenseSource(meta[:, :21].to(torch.float64), m0),
            engine.DenseSource(meta[:, 21:26] != 0, m0 + 21),
            engine.DenseSource(meta[:, 26:].to(torch.float64), m0 + 26)]


def cpu_baseline(x_host, offsets, weights, k, min_similarity, budget_s):
    """The oracle (a NumPy port of the reference path) timed on the host cores on a bounded
    sample: query rows in chunks of 512 against all n columns, float64 like the reference,
    until `budget_s` seconds of work are done."""
    import numpy as np

    from oracle import similarity_oracle as so

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    n = x_host.shape[0]
    t0 = time.perf_counter()
    blocks = [x_host[:, offsets[b] : offsets[b + 1]].astype(np.float64) for b in range(len(weights))]
    z = so.build_z(blocks, weights)
    del blocks
    t_norm = time.perf_counter() - t0
    rows, t_rows = 0, 0.0
    while rows < n and t_rows < budget_s:
        hi = min(n, rows + 512)
        t0 = time.perf_counter()
        so.topk_from_z(z, k, min_similarity, rows=range(rows, hi), chunk=512)
        t_rows += time.perf_counter() - t0
        rows = hi
    # the normalisation pass is paid once per catalogue: charge the sample its share
    t = t_rows + t_norm * rows / n
    # second mode: the reference's production loop taken literally (one show at a time, five
    # one-row cosine similarities that re-normalise the whole block, a full argsort and the walk:
    # scripts/populate_database.py:170-218), on a handful of rows; hybrid data only
    row_loop = None
    if len(weights) == 3:
        import scipy.sparse as sp

        g, tx, m = (x_host[:, offsets[b] : offsets[b + 1]].astype(np.float64) for b in range(3))
        feats = {"genre_features": g, "text_features": sp.csr_matrix(tx), "platform_features": m[:, :21],
                 "type_features": m[:, 21:26], "language_features": m[:, 26:]}
        del tx
        sample, t_loop = 0, 0.0
        while sample < min(n, 64) and t_loop < budget_s / 3:
            t0 = time.perf_counter()
            so.populate_rows(feats, list(range(n)), weights[0], weights[1], weights[2], k, min_similarity,
                             rows=[sample])
            t_loop += time.perf_counter() - t0
            sample += 1
        row_loop = {"value": sample * n / t_loop, "unit": "show-pairs/s", "rows": sample,
                    "ms_per_show": t_loop / sample * 1e3,
                    "note": "literal per-show loop of the reference's populate_database.py; whole-catalogue "
                            "time = ms_per_show x n (extrapolated)"}
    return {
        "value": rows * n / t,
        "unit": "show-pairs/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"{rows} of {n} query rows x all {n} columns, float64 chunked GEMM + per-row "
                  f"partition/sort top-{k} (oracle/similarity_oracle.py), {t:.1f} s of CPU work",
        "row_loop": row_loop,
    }


def end_to_end(x_host, offsets, weights, k, min_similarity, dev):
    """One pass of the drop-in from host feature arrays to host neighbour table, timed with the wall
    clock: what a caller pays on top of the device step (reported beside `value`, never inside it).
    The packed matrix is first split into the reference's five arrays in their stored form (int64
    genres, CSR float64 text, float64 / bool one-hots; untimed); then
    scripts.populate_database.compute_top_neighbours -- upload, assembly, norms, sweep, float64
    re-rank, download -- the columnar table, and the reference's dict-of-records form."""
    import numpy as np
    import scipy.sparse as sp
    import torch

    from tvbingefriend_recommendation_service_amd.scripts.populate_database import (
        compute_top_neighbours, neighbour_records, neighbour_table)

    n = x_host.shape[0]
    g, tx, m = (x_host[:, offsets[b] : offsets[b + 1]] for b in range(3))
    feats = {"genre_features": g.astype(np.int64), "text_features": sp.csr_matrix(tx.astype(np.float64)),
             "platform_features": m[:, :21].astype(np.float64), "type_features": m[:, 21:26].astype(bool),
             "language_features": m[:, 26:].astype(np.float64)}
    show_ids = list(range(1, n + 1))
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    result = compute_top_neighbours(feats, weights[0], weights[1], weights[2], k, min_similarity, device=dev)
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    table = neighbour_table(result, show_ids)
    t.append(time.perf_counter())
    records = neighbour_records(result, show_ids, 0, n)
    t.append(time.perf_counter())
    d = [b - a for a, b in zip(t, t[1:])]
    return {
        "host_features_to_host_table_s": d[0] + d[1],
        "compute_top_neighbours_s": d[0], "columnar_table_s": d[1],
        "pairs_kept": int(len(table["show_id"])),
        "record_dicts_s": d[2], "shows_with_records": len(records),
        "note": "single pass of the drop-in (upload from pageable memory, assembly from CSR / dense sources, fp32 sweep, "
                "float64 re-rank, download), cold allocations included; record_dicts_s = the reference's "
                "dict[int, list[dict]] form built from the same arrays (csrc/_tvbf_records.so when it is built, else a "
                "Python loop), not part of host_features_to_host_table_s",
    }


def launch_ranks(args) -> int:
    """`--gpus N` without a launcher: start N fresh rank processes of this script (rank r on GPU r,
    rendezvous on 127.0.0.1) and wait for them.  Runs before anything here has touched the GPU --
    the children initialise it, this process never does.  Rank 0 prints the JSON line on the
    inherited stdout.  When a rank fails the others are ended (by PID) and its code is returned."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env))
    code = 0
    while any(p.poll() is None for p in procs):
        failed = [p for p in procs if p.poll() not in (None, 0)]
        if failed:
            code = failed[0].returncode
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.1)
    for p in procs:
        rc = p.wait()
        if rc and not code:
            code = rc
    return code


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    import tvbf_amd  # noqa: F401
    from tvbingefriend_recommendation_service_amd import engine, native, sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start --gpus ranks")
    if args.backend == "nccl" and world > torch.cuda.device_count():
        raise SystemExit(f"--gpus {world} with the nccl backend needs {world} GPUs, {torch.cuda.device_count()} "
                         "visible (--backend gloo rehearses several ranks on one GPU)")
    dev = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    min_sim = args.min_similarity
    if min_sim is None:
        min_sim = 0.1 if args.data == "hybrid" else -1.0
    shard = sharding.shard_bounds(args.n, world, rank)
    x, offsets, weights = make_shard(args, shard, dev)
    scales = [float(np.sqrt(w)) for w in weights]
    ld = engine.padded_ld(args.d)

    # persistent buffers: the large ones are allocated once, outside the timed region
    exact = args.path == "exact"
    kc = engine.candidate_count(args.k) if exact else args.k       # candidates per show the sweep collects
    sweep_min = min_sim - engine.error_margin(ld, weights) if exact else min_sim
    # Several ranks: where the stored features of all shows are resident on every rank -- the exact
    # path needs them for the re-rank -- each rank builds the whole normalised matrix itself (K1 over
    # all rows, 0.15 ms at the headline size) and no all-gather of Z is needed; otherwise (or with
    # --z-exchange gather) each rank builds its shard and one RCCL all-gather distributes them.
    z_local = world > 1 and (args.z_exchange == "local" or (args.z_exchange == "auto" and exact))
    if z_local:
        z_all = torch.zeros((shard.padded_total, ld), dtype=torch.float32, device=dev)
        z_shard = z_all[rank * shard.per_rank:(rank + 1) * shard.per_rank]
    else:
        z_shard = torch.empty((shard.per_rank, ld), dtype=torch.float32, device=dev)
        z_all = z_shard if world == 1 else torch.empty((shard.padded_total, ld), dtype=torch.float32, device=dev)
    nq = shard.rows
    groups = [0, 1, 2][: len(weights)]
    gw = (list(weights) + [0.0, 0.0])[:3]
    # The stored features of ALL shows on every rank (what every rank of the reference's pipeline loads
    # from the feature files; the float64 re-rank reads the candidates' stored rows): every rank
    # generates all shards itself, before the timed region.  Hybrid rows are kept in the reference's
    # stored form (int64 genres, CSR float64 text, float64 / bool one-hots), dense rows as one packed
    # fp32 matrix.
    if world == 1:
        x_all = x
    else:
        x_all = torch.cat([make_shard(args, sharding.shard_bounds(args.n, world, r), dev)[0] for r in range(world)])
    if args.data == "hybrid":
        src_all = stored_sources(engine, x_all[: args.n], offsets)
    else:
        src_all = [engine.DenseSource(x_all[: args.n], 0)]
    src_shard = engine.slice_sources(src_all, shard.row0, nq)
    x_host = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        x_host = x[:nq].cpu().numpy()
    del x, x_all  # only the stored form stays on the device (for dense rows it is the packed matrix itself)
    z_shard.zero_()  # padding rows of the last shard stay zero
    # the gathered variant beside a resident-Z run (a few steps after the timed region): its own shard buffer
    z_shard_g = None
    if z_local and exact and not args.no_gather_compare:
        z_shard_g = torch.zeros((shard.per_rank, ld), dtype=torch.float32, device=dev)

    def new_out(kk):
        return engine.TopK(torch.empty((nq, kk), dtype=torch.int32, device=dev),
                           torch.empty((nq, kk), dtype=torch.float32, device=dev),
                           torch.empty((nq,), dtype=torch.int32, device=dev))

    out = new_out(kc)
    slots = 16 if world == 1 else 32
    ws_bytes = int(native.load().tvbf_topk_workspace_bytes_for_slots(max(nq, 1), kc, slots))
    workspace = torch.empty((ws_bytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
    sym_bytes = engine.symmetric_workspace_bytes(args.n, ld, kc) if args.sweep != "plain" else 0
    if world > 1 and not sharding.symmetric_route_fits(world, kc, ld):
        sym_bytes = 0
    if world == 1 and sym_bytes and args.sweep == "auto" and engine.choose_catalogue_route(
            sym_bytes, ws_bytes, engine.available_bytes(dev)) != "symmetric":
        sym_bytes = 0  # not enough free memory for its score matrices and column buffers: same lists from the plain sweep
    if args.sweep == "symmetric" and not sym_bytes:
        raise SystemExit("--sweep symmetric: row length a multiple of 64, more than 1920 shows, and for several "
                         "GPUs a K small enough that one merge takes all lists")
    symmetric = sym_bytes > 0

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # The route is decided here, before any collective of a step: the shape test above is the same
    # on every rank, the buffers of the symmetric sharded route are allocated now, and one flag is
    # reduced over the ranks.  After this point an exception inside a step ends the job.
    side = None
    if symmetric and world == 1:
        workspace = torch.empty((sym_bytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
    elif world > 1:
        failed = 0.0
        if symmetric:  # lists of the own, half and (gated) plain sweeps; column buffers of all shows
            try:
                sym_ws_bytes = int(native.load().tvbf_topk_workspace_bytes_for_slots(max(nq, 1), kc, 96))
                sym_workspace = torch.empty((sym_ws_bytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
                side = engine.ColumnSide(shard.padded_total, dev, cap=1024 if kc <= 64 else 2048)
                torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                failed = 1.0
                print(f"[rank {rank}] buffers of the symmetric sharded route: {e!r}", file=sys.stderr, flush=True)
        if max_over_ranks(failed) > 0.0:
            symmetric, side = False, None
            if rank == 0:
                print("symmetric sharded route not available on every rank: plain sharded route", file=sys.stderr,
                      flush=True)
        elif symmetric:
            slots, workspace = 96, sym_workspace
    # a rank's own block (its rows against their own columns) goes through the stored-matrix route
    # when it is small enough for it and the memory is there: the buffer is allocated once
    own_matrix = None
    if world > 1 and args.sweep != "plain":
        mbytes = engine.own_matrix_bytes(nq, ld)
        if mbytes and mbytes <= engine.MEMORY_HEADROOM * engine.available_bytes(dev):
            own_matrix = torch.empty((mbytes + 7) // 8, dtype=torch.int64, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    kernel_ms = []
    phase_marks = []  # per timed step: [(name, event), ...] on the compute stream
    rerank_stats = {}
    tiles_step = [0]  # 128 x 128 score tiles the sweeps of one step compute on this rank
    own, after, before = sharding.column_pieces(shard)

    class HostGather:
        """gloo rehearsal: CPU tensors only, copied synchronously to and from the device."""

        def __init__(self, zs):
            host = zs.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host)
            self.full = torch.cat(parts, dim=0)

        def wait(self):
            z_all.copy_(self.full)

    class Resident:
        """Z of all shows was built on this rank: nothing to wait for."""

        def wait(self):
            pass

    def start_gather(zs, local):
        if world == 1:
            return None
        if local:
            return Resident()
        if args.backend == "nccl":
            return dist.all_gather_into_tensor(z_all, zs, async_op=True)
        return HostGather(zs)

    def step(timed, path_exact=exact, gathered=False):
        """``gathered``: the all-gather variant in a run whose Z is resident (own shard built into
        z_shard_g, the shards all-gathered into z_all)."""
        marks = []
        local = z_local and not gathered
        zs = z_shard_g if gathered else z_shard

        def mark(name):
            # phase boundary on the compute stream; an event after work.wait() completes when the
            # gathered shards are there, so "gather_wait" is the part of the all-gather not hidden
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                marks.append((name, e))

        kk = kc if path_exact else args.k
        ms_sweep = sweep_min if path_exact else min_sim
        res = out if kk == kc else out_k
        mark("start")
        if local:  # the drop-in's K1 over the stored arrays of all shows
            engine.assemble_normalize(src_all, args.n, offsets, scales, out=z_all[: args.n])
        elif nq:  # this rank's rows of Z; the others arrive with the all-gather
            engine.assemble_normalize(src_shard, nq, offsets, scales, out=zs[:nq])
        mark("k1")
        work = start_gather(zs, local)
        if symmetric and world == 1:
            if timed:
                ev[0].record()
            engine.cosine_topk_symmetric(zs, kk, ms_sweep, n=nq, workspace=workspace, out=res)
            tiles_step[0] = engine.symmetric_tiles(nq, ld, kk, dev)
        elif symmetric:
            job = sharding.SymmetricShardJob(zs, shard, kk, ms_sweep, slots=slots, workspace=workspace,
                                             colside=side, own_matrix=own_matrix)
            if timed:
                ev[0].record()
            job.phase_own()  # overlaps the all-gather of Z
            mark("own")
            bounds_all = sharding.gather_words(job.own_bounds(), world)
            mark("bounds_gather")
            work.wait()
            if not local:
                mark("gather_wait")
            job.phase_half(z_all, bounds_all)
            mark("half")
            packed = sharding.pack_lists_and_flags(job.lists_for_peers(bounds_all), job.flag, shard)
            mark("col_reduce")
            lists, flag = sharding.unpack_lists_and_flags(sharding.exchange_lists(packed), shard, kk)
            mark("exchange")
            job.finish(z_all, lists, flag, res)
            mark("finish")
            tiles_step[0] = job.tiles
        else:
            session = engine.TopkSession(zs, nq, shard.row0, kk, ms_sweep, slots=slots, workspace=workspace)
            if timed:
                ev[0].record()
            if own_matrix is not None and world > 1:  # own columns: overlaps the all-gather
                session.accumulate_own_matrix(own_matrix)
            else:
                session.accumulate((zs, own[1], own[0]))
            mark("own")
            if work is not None:
                work.wait()
                if not local:
                    mark("gather_wait")
                session.accumulate((z_all[after[0]:], after[1], after[0]), (z_all, before[1], before[0]))
                mark("rest")
            session.finish(res)
            mark("finish")
            tiles_step[0] = session.tiles
        if timed:
            ev[1].record()
        mark("sweeps")
        if path_exact:
            top = engine.exact_topk(z_all, src_all, offsets, groups, gw, args.k, min_sim, n=args.n, q_row0=shard.row0,
                                    nq=nq, cand=res)
            rerank_stats.update(top.stats)
            result = top
        else:
            result = engine.component_scores(z_all, offsets, scales, res.idx, q_row0=shard.row0, n=args.n)
        mark("rerank" if path_exact else "k5")
        if timed:
            ev[1].synchronize()
            kernel_ms.append(ev[0].elapsed_time(ev[1]))
            phase_marks.append(marks)
        return result

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out_k = out if kc == args.k else new_out(args.k)
    if symmetric and world > 1:
        step(False)  # untimed: the first use of the int64 all-gather and the all-to-all
    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    tiles_main = tiles_step[0]  # of the timed path (the comparison steps below run another K)
    if world > 1:
        elapsed = max_over_ranks(elapsed)
        kernel_avg_ms = max_over_ranks(sum(kernel_ms) / max(len(kernel_ms), 1))
    else:
        kernel_avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)

    # per-phase device time (HIP events on the compute stream), averaged over the timed steps, and
    # for several ranks the maximum over the ranks of every phase: where a step's time goes
    def phases_of(marks_list):
        acc = {}
        for marks in marks_list:
            for (_, e0), (name, e1) in zip(marks, marks[1:]):
                acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1) / len(marks_list)
        if world > 1:
            acc = {name: max_over_ranks(v) for name, v in acc.items()}
        return {f"{name}_ms": round(v, 4) for name, v in acc.items()}

    phases = phases_of(phase_marks)

    # several ranks with Z built on every rank: the all-gather variant (own shard, one all-gather of the
    # shards behind the own block) timed the same way on a few steps, with its phases, beside the main figure
    gather_step = None
    if z_shard_g is not None:
        n_cmp = max(1, min(args.steps, 8))
        kept_kernel_ms = list(kernel_ms)
        try:  # a failure here (every rank runs the same calls) must not cost the main figure
            step(False, gathered=True)
            phase_marks.clear()
            barrier()
            t1 = time.perf_counter()
            for _ in range(n_cmp):
                step(True, gathered=True)
            barrier()
            el = max_over_ranks(time.perf_counter() - t1)
            gather_step = {"ms_per_step": el / n_cmp * 1e3, "value": float(args.n) * float(args.n) * n_cmp / el,
                           "steps": n_cmp, "phases": phases_of(phase_marks),
                           "note": "the same exact step with each rank building only its shard of Z and one "
                                   "all-gather of the shards (--z-exchange gather); gather_wait_ms = the part of "
                                   "the all-gather the own block did not hide"}
        except Exception as e:  # noqa: BLE001
            gather_step = {"error": repr(e)}
        phase_marks.clear()
        kernel_ms[:] = kept_kernel_ms

    # the fp32-only step (lists of the fp32 scores + K5), timed the same way on a few steps, beside the exact one
    fp32_step = None
    if exact and not args.no_fp32_compare:
        n_cmp = max(1, min(args.steps, 8))
        if world > 1 and symmetric and kc != args.k and not sharding.symmetric_route_fits(world, args.k, ld):
            n_cmp = 0
        try:  # an auxiliary figure: its failure is reported, the main figure stands
            if n_cmp:
                step(False, path_exact=False)
                barrier()
                t1 = time.perf_counter()
                for _ in range(n_cmp):
                    step(False, path_exact=False)
                barrier()
                el = time.perf_counter() - t1
                if world > 1:
                    el = max_over_ranks(el)
                fp32_step = {"ms_per_step": el / n_cmp * 1e3, "value": float(args.n) * float(args.n) * n_cmp / el,
                             "steps": n_cmp, "note": "K1 -> fp32 lists of K -> K5 (fp32 per-block scores): no float64 "
                                                      "re-rank, indices exact only where fp32 separates the scores"}
        except Exception as e:  # noqa: BLE001
            fp32_step = {"error": repr(e)}

    if args.check and nq:
        step(False)  # the candidates of the timed path once more: `out` must hold them
        ref = engine.TopkSession(z_shard, nq, shard.row0, kc, sweep_min)
        ref.accumulate((z_shard, own[1], own[0]))
        if world > 1:
            ref.accumulate((z_all[after[0]:], after[1], after[0]), (z_all, before[1], before[0]))
        want = ref.finish()
        torch.cuda.synchronize()
        same = all(torch.equal(getattr(out, f), getattr(want, f)) for f in ("idx", "score", "count"))
        print(f"[rank {rank}] lists equal to the plain sweep: {same}", file=sys.stderr, flush=True)
        if not same:
            raise SystemExit(3)

    pairs = float(args.n) * float(args.n) * args.steps
    # dominant kernel: 2*D flops per ordered pair, nq*n pairs per launch on this rank.  `achieved` is
    # what the matrix pipe really executed: the library's count of 128 x 128 score tiles of the
    # step's sweeps x 2 * 128 * 128 * ld flops (zero padding columns of Z included, they are
    # multiplied like any other), over the HIP-event time around them.
    per_rank_rows = shard.per_rank if world > 1 else nq
    algorithmic = 2.0 * per_rank_rows * args.n * args.d
    executed = 2.0 * tiles_main * 128.0 * 128.0 * ld
    achieved = executed / (kernel_avg_ms * 1e-3) / 1e12
    effective = algorithmic / (kernel_avg_ms * 1e-3) / 1e12
    plain_tiles = ((nq + 127) // 128) * ((args.n + 127) // 128)
    workload = (f"N={args.n} shows x D={args.d} fp32 ({'genre32|text%d|meta32 hybrid-like' % (args.d - 64) if args.data == 'hybrid' else 'dense gaussian'}), "
                f"K={args.k}, min_similarity={min_sim}")
    if exact:
        workload += f" (sweep collects {kc} candidates per show for the float64 re-rank)"
    traffic = None
    tf = ROOT / "profiles" / "traffic.json"
    if tf.exists():
        try:
            key = f"n{args.n}_d{args.d}_k{kc}_{args.data}_g{world}" + ("_sym" if symmetric else "")
            traffic = json.loads(tf.read_text()).get(key)
        except Exception:
            traffic = None
    share = tiles_main / plain_tiles if plain_tiles else None  # of the ordered pairs, really computed
    line = {
        "metric": "show-pairs/sec (all-pairs cosine + per-row top-K)",
        "value": pairs / elapsed,
        "unit": "show-pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": workload, "rows_per_gpu": per_rank_rows,
                   "parallelism": (f"row-shard x{world}, "
                                   + ("Z built on every rank from the resident stored features (no all-gather of Z)"
                                      if z_local else "all-gather of the normalised shards")
                                   + (" + all-gather of bound keys + all-to-all of column-side lists" if symmetric else "")
                                   + f" ({args.backend})") if world > 1 else "single GPU",
                   "z_exchange": ("local" if z_local else "gather") if world > 1 else None,
                   "sweep": "symmetric: every unordered pair of shows computed once, both shows' lists fed from it; "
                            "roofline.achieved counts the flops executed" if symmetric else "plain: every ordered pair computed"},
        "topk_build_time_s": elapsed / args.steps,
        "path": ("exact: fp32 sweep for %d candidates per show, float64 re-rank to K=%d from the stored features "
                 "(reference's float64 order, scores, threshold cut and per-block scores)" % (kc, args.k)) if exact
                else "fp32: fp32 lists of K + K5",
        "phases": phases,
        "rerank": dict(rerank_stats) if exact else None,
        "fp32_step": fp32_step,
        "gather_step": gather_step,
        "rccl_world": (dist.get_world_size() if world > 1 else 1),
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                     "kernel": ("cosine_topk_dma_kernel<false, 0, false> + <false, 0, true> (own block and half sweep of the "
                                "symmetric sharded route; HIP events span the collectives between them as well)"
                                if symmetric and world > 1 else
                                "cosine_topk_dma_kernel<false, 2> (half sweep of the symmetric route) + cosine_tri_dma_kernel<0> "
                                "(the shards' own blocks); HIP events span the whole call: fills, row selection of the own "
                                "blocks, column-buffer reduction, list merge"
                                if symmetric else
                                "cosine_topk_dma_kernel<false, 0> (HIP events also span the 4 us bound fill and the "
                                "0.3 ms list merge of the same call)"), "kernel_ms": kernel_avg_ms,
                     "pairs_computed_share": share,
                     "algorithmic_flops": algorithmic, "executed_flops": executed,
                     "effective_vs_2N2D": effective, "effective_frac_vs_2N2D": effective / FP32_MFMA_PEAK_TFLOPS,
                     "note": "achieved and frac = flops the MFMA pipe executed (tile count from the library) over "
                             "the HIP-event time, a physical pipe fraction; effective_vs_2N2D credits 2*D flops "
                             "for every ordered pair, computed or not (the work the path is specified by)"},
    }
    if x_host is not None and not args.no_end_to_end and len(weights) == 3:
        del workspace  # the measurement allocates its own
        try:
            line["end_to_end"] = end_to_end(x_host, offsets, weights, args.k, min_sim, dev)
        except Exception as e:  # noqa: BLE001
            line["end_to_end"] = {"error": repr(e)}
    if x_host is not None and not args.no_cpu_baseline:
        try:
            line["cpu_baseline"] = cpu_baseline(x_host, offsets, weights, args.k, min_sim, args.cpu_seconds)
        except Exception as e:  # noqa: BLE001
            line["cpu_baseline"] = {"error": repr(e), "value": None, "unit": "show-pairs/s", "cores": 0, "kind": "port",
                                    "sample": "failed"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
