#!/usr/bin/env python3
"""Phase breakdown of the fused kernel's k-block loop (needs a timing-only build made with
-DTVBF_DIAG_TIMELINE; such a build is never shipped):

    TVBF_LIB=build/libtvbf_timeline.so python tools/timeline.py [n] [d] [k] [hybrid|dense]

Every wavefront sums the cycle-counter time between six marks of the loop; printed are the mean
ticks per k-block (the counter's own rate: see "ticks per second") and their share of the period:
  loop    end of a k-block -> top of the next
  wait1   arrival at the first barrier -> released
  write   wait for the global loads + register -> LDS stores issued
  wait2   LDS stores complete + second barrier
  mfma    address arithmetic, global load issue, fragment reads, 64 MFMAs issued
  epi     epilogue
"""
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
kind = sys.argv[4] if len(sys.argv) > 4 else "hybrid"
dev = torch.device("cuda:0")
if kind == "dense":
    x = torch.randn((n, d), device=dev)
    offsets, scales, ms = [0, d], [1.0], -1.0
else:
    x = torch.zeros((n, d), device=dev)
    x[:, :32] = (torch.rand((n, 32), device=dev) < 0.08).float()
    t_ = torch.randn((n, d - 64), device=dev).abs_()
    t_ *= torch.rand((n, d - 64), device=dev) < 0.03
    x[:, 32:d - 32] = t_
    del t_
    x[torch.arange(n, device=dev), d - 32 + torch.randint(0, 32, (n,), device=dev)] = 1.0
    offsets, scales, ms = [0, 32, d - 32, d], [0.4 ** 0.5, 0.5 ** 0.5, 0.1 ** 0.5], 0.1
z = engine.normalize_blocks(x, offsets, scales)
del x
engine.cosine_topk(z, k, ms)
torch.cuda.synchronize()
buf = torch.zeros((1 << 16, 4, 16), dtype=torch.int64, device=dev)
os.environ["TVBF_DIAG_BUF"] = hex(buf.data_ptr())
t0 = time.perf_counter()
engine.cosine_topk(z, k, ms)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
raw = buf.cpu().numpy()
used = raw[:, 0, 6] > 0
raw = raw[used]
b = raw.astype(np.float64)
kblocks = b[..., 6].sum()
tot = b[..., :6].sum(axis=(0, 1))
names = ["loop", "wait1", "write", "wait2", "mfma", "epi"]
period = tot.sum() / kblocks
print(f"{used.sum()} workgroups, {int(kblocks)} wave-k-blocks, kernel+merge wall {wall * 1e3:.1f} ms")
# 2 workgroups per CU run concurrently: wall ~= sum of per-workgroup time / (2 * CUs)
ticks_per_s = tot.sum() / 4 / (512 * wall)
print(f"ticks per second ~ {ticks_per_s / 1e9:.2f} G  (period {period:.0f} ticks = {period / ticks_per_s * 1e6:.2f} us)")
for nme, v in zip(names, tot):
    print(f"{nme:6s} {v / kblocks:8.1f} ticks  {100 * v / tot.sum():5.1f} %")
tiles = kblocks / (d / 32)
ev = [(raw[..., 7] & 0xFFFFFFFF).sum(), (raw[..., 7] >> 32).sum(), (raw[..., 8] & 0xFFFFFFFF).sum(), (raw[..., 8] >> 32).sum()]
print(f"per wave and tile (4 sub-tiles): slow paths {ev[0] / tiles:.3f}  pop rounds {ev[1] / tiles:.3f}  "
      f"accepted keys {ev[2] / tiles:.3f}  compactions {ev[3] / tiles:.3f}")
print(f"epilogue cycles per tile and wave: {tot[5] / tiles:.0f}; in slow paths {b[..., 9].sum() / tiles:.0f} "
      f"(dump + mask {b[..., 10].sum() / tiles:.0f} double-counted in the pop-loop figure? no: before it; compactions {b[..., 11].sum() / tiles:.0f})")
# by launch order (workgroups are dispatched in blockIdx order: 512 at a time)
order = np.nonzero(used)[0]
for lo in range(0, len(order), 1024):
    sel = slice(lo, lo + 1024)
    kb_ = b[sel, :, 6].sum()
    if kb_ == 0:
        continue
    t5 = b[sel, :, 5].sum() / kb_
    sp = (raw[sel][..., 7] & 0xFFFFFFFF).sum() / (kb_ / (d / 32))
    print(f"  workgroups {lo:6d}+: epi {t5:7.1f} ticks per k-block, slow paths per tile {sp:.3f}, period {b[sel, :, :6].sum() / kb_:.0f}")

# workgroup lifetimes on the 100 MHz wall clock: how busy are the 2 x CU slots, how long are start and end
start = raw[:, :, 12].min(axis=1).astype(np.float64)
end = raw[:, :, 15].max(axis=1).astype(np.float64)
life = (end - start) / 100.0  # us
span = (end.max() - start.min()) / 100.0
loop_cycles = b[:, :, :6].sum(axis=2).max(axis=1)
wave_cycles = raw[:, :, 13].astype(np.float64).max(axis=1)
rate = (wave_cycles / (life * 1e-6)).mean()
print(f"kernel span {span / 1e3:.2f} ms; workgroup lifetime mean {life.mean() / 1e3:.3f} ms; sum of lifetimes / (512 slots x span) = "
      f"{life.sum() / (512 * span):.3f}")
print(f"cycle counter rate {rate / 1e9:.3f} GHz; outside the k-block loop: {100 * (1 - loop_cycles.sum() / wave_cycles.sum()):.2f} % of a workgroup's cycles")
hw = raw[:, 0, 14]
cu = ((hw >> 32) << 20) | (hw & 0xFFFFFFFF) >> 8 & 0xFFF  # xcc, then cu_id / sh / se bits 8..19? keep 12 bits
slots = {}
for i in np.argsort(start):
    slots.setdefault(int(cu[i]), []).append((start[i], end[i]))
gaps = []
for v in slots.values():
    # two slots per CU: greedy assignment to measure idle time between successive workgroups
    ends = []
    for s_, e_ in v:
        ends.sort()
        if len(ends) >= 2:
            gaps.append(s_ - ends[0])
            ends[0] = e_
        else:
            ends.append(e_)
if gaps:
    g = np.array(gaps) / 100.0
    print(f"{len(slots)} CUs seen; gap between a workgroup's end and its successor's start: mean {g.mean():.1f} us, p90 {np.percentile(g, 90):.1f} us")
