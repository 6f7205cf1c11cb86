#!/usr/bin/env python3
"""Phase breakdown of the LDS-DMA kernels' k-block pair loop (needs a timing-only build made with
-DTVBF_DIAG_TIMELINE; such a build is never shipped):

    TVBF_LIB=build/libtvbf_timeline.so python tools/timeline_dma.py [n] [d] [k] [plain|symmetric] [hybrid|dense]

Every wavefront sums the cycle-counter time between the marks of a pair of k-blocks:
  issueA  LDS-DMA transfers and query loads of the next k-block issued
  mfmaA   fragment reads + 64 MFMAs issued (first k-block of the pair)
  barA    wait for the own transfers + workgroup barrier
  issueB / mfmaB / barB   the same for the second k-block
  epi     epilogue (last pair of a tile only)
  top     per-tile work before the pair loop
With the symmetric sweep the half sweep is the launch that is recorded last in the buffer.

A build with -DTVBF_DIAG_TOPSPLIT on top splits `top` of a sweep with a column side: the column
bounds of the tile are booked as issueA, the drain of the column queue as barA, the barrier behind
it as barB and a change of row block as issueB (compare with a run of the plain timeline build);
the three row-side counters then read: cycles inside bound refreshes, drains, refreshes."""
import os
os.environ.setdefault("TVBF_TUNING", "1")  # the library looks at its TVBF_* switches only then
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import tvbf_amd  # noqa: F401
from tvbingefriend_recommendation_service_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
sweep = sys.argv[4] if len(sys.argv) > 4 else "plain"
data = sys.argv[5] if len(sys.argv) > 5 else "hybrid"  # or "dense": gaussian rows, threshold off
dev = torch.device("cuda:0")
if data == "dense":
    z = engine.normalize_blocks(torch.randn((n, d), device=dev), [0, d], [1.0])
    ms = -1.0
else:
    x = torch.zeros((n, d), device=dev)
    x[:, :32] = (torch.rand((n, 32), device=dev) < 0.08).float()
    t_ = torch.randn((n, d - 64), device=dev).abs_()
    t_ *= torch.rand((n, d - 64), device=dev) < 0.03
    x[:, 32:d - 32] = t_
    del t_
    x[torch.arange(n, device=dev), d - 32 + torch.randint(0, 32, (n,), device=dev)] = 1.0
    z = engine.normalize_blocks(x, [0, 32, d - 32, d], [0.4 ** 0.5, 0.5 ** 0.5, 0.1 ** 0.5])
    del x
    ms = 0.1
run = (lambda: engine.cosine_topk(z, k, ms)) if sweep == "plain" else (lambda: engine.cosine_topk_symmetric(z, k, ms))
run()
torch.cuda.synchronize()
buf = torch.zeros((1 << 12, 4, 16), dtype=torch.int64, device=dev)
os.environ["TVBF_DIAG_BUF"] = hex(buf.data_ptr())
t0 = time.perf_counter()
run()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
raw = buf.cpu().numpy().astype(np.float64)
raw = raw[raw[:, 0, 8] > 0]
names = ["issueA", "mfmaA", "barA", "issueB", "mfmaB", "epi", "barB", "top"]
tot = raw[..., :8].sum(axis=(0, 1))
pairs, tiles = raw[..., 8].sum(), raw[..., 9].sum()
print(f"{len(raw)} workgroups, {int(pairs)} wave-pairs, {int(tiles)} wave-tiles, call wall {wall * 1e3:.1f} ms")
print(f"cycles per pair of k-blocks (128 MFMAs = 8192 pipe cycles of this wave): {tot.sum() / pairs:.0f}")
for nme, v in zip(names, tot):
    print(f"{nme:7s} {v / pairs:8.1f} per pair  {100 * v / tot.sum():5.1f} %")
print(f"epilogue per tile and wave: {tot[5] / tiles:.0f} cycles; per-tile top: {tot[7] / tiles:.0f}")
cand = raw[..., 14].sum()
pops = (raw[..., 15].astype(np.int64) & 0xFFFFFFFF).sum()
compact = (raw[..., 15].astype(np.int64) >> 32).sum()
print(f"row side per tile and wave: {cand / tiles:.2f} scores offered to the lists, {pops / tiles:.2f} trips of the pop loop, "
      f"{compact / tiles:.3f} compactions; per row and sweep: {cand / n:.0f} offered, {compact / n:.1f} compactions")
w0, w1 = raw[..., 11], raw[..., 12]  # constant-rate clock (100 MHz) at a wave's start and end
t0_, span = w0.min(), w1.max() - w0.min()
print(f"waves start {100 * (w0.mean() - t0_) / span:.2f} % (latest {100 * (w0.max() - t0_) / span:.2f} %) into the kernel and end at "
      f"{100 * (w1.mean() - t0_) / span:.2f} % on average (earliest {100 * (w1.min() - t0_) / span:.2f} %): "
      f"{100 * (w1 - w0).sum() / (span * w0.size):.2f} % of wave slots x time is occupied")
nwg = len(raw)
end_wg = (w1.max(axis=1) - t0_) / span
print(f"workgroups end at (share of the kernel's duration): first half of the grid {end_wg[:nwg // 2].mean():.3f}, second half "
      f"{end_wg[nwg // 2:].mean():.3f}; by blockIdx % 8: " + " ".join(f"{end_wg[x::8].mean():.3f}" for x in range(8)))
hw = raw[:, 0, 13].astype(np.int64)
cu_key = ((hw >> 32) & 0xF) * 4096 + (hw & 0xFFFFFFFF & 0x3F00) + ((hw >> 13) & 0x7) * 16 * 0  # XCC, SE/SH/CU bits of HW_ID
cu_key = ((hw >> 32) & 0xF) * 65536 + ((hw >> 8) & 0xF) + (((hw >> 12) & 0x1) << 4) + (((hw >> 13) & 0x7) << 5)
ends = {}
for key, e in zip(cu_key, end_wg):
    ends.setdefault(int(key), []).append(float(e))
pairs_ = [sorted(v) for v in ends.values() if len(v) == 2]
if pairs_:
    first = np.mean([v[0] for v in pairs_]); second = np.mean([v[1] for v in pairs_])
    print(f"{len(ends)} compute units seen, {len(pairs_)} with two workgroups: the first to finish ends at {first:.3f}, the other at {second:.3f}")
print(f"in the loop: {100 * tot.sum() / raw[..., 10].sum():.1f} % of the waves' lifetime")
