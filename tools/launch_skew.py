#!/usr/bin/env python3
"""When the workgroups of one sweep start and end (needs a timing-only build made with
-DTVBF_DIAG_TIMELINE; such a build is never shipped):

    TVBF_LIB=build/libtvbf_timeline.so python tools/launch_skew.py [rows] [cols] [d] [k]

One plain sweep of `rows` query rows against `cols` columns of a hybrid-like matrix, warm bounds
(a first sweep has run).  Every wave records the device-wide 100 MHz counter when it starts and
when it ends; printed: the launch's span, how late workgroups start, how early they end, and the
share of the span in which a compute unit's two workgroups are both alive."""
import os
os.environ.setdefault("TVBF_TUNING", "1")  # the library looks at its TVBF_* switches only then
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import tvbf_amd  # noqa: F401
from tvbingefriend_recommendation_service_amd import engine

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 43_750
d = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
k = int(sys.argv[4]) if len(sys.argv) > 4 else 20
n = rows + cols
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(30001)
x = torch.zeros((n, d), device=dev)
x[:, :32] = (torch.rand((n, 32), generator=g, device=dev) < 0.08).float()
t_ = torch.randn((n, d - 64), generator=g, device=dev).abs_()
t_ *= torch.rand((n, d - 64), generator=g, device=dev) < 0.03
x[:, 32:d - 32] = t_
del t_
x[torch.arange(n, device=dev), d - 32 + torch.randint(0, 32, (n,), generator=g, device=dev)] = 1.0
z = engine.normalize_blocks(x, [0, 32, d - 32, d], [0.4 ** 0.5, 0.5 ** 0.5, 0.1 ** 0.5])
del x

session = engine.TopkSession(z[:rows], rows, 0, k, 0.1, slots=16)
session.accumulate((z[:rows], rows, 0))            # own block: warm bounds, as a rank has them
torch.cuda.synchronize()
buf = torch.zeros((1 << 12, 4, 16), dtype=torch.int64, device=dev)
os.environ["TVBF_DIAG_BUF"] = hex(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
session.accumulate((z[rows:], cols, rows))
e1.record()
torch.cuda.synchronize()
del os.environ["TVBF_DIAG_BUF"]
raw = buf.cpu().numpy()
live = raw[:, 0, 12] > 0
start = raw[live][:, :, 11].min(axis=1).astype(np.float64) / 100.0   # microseconds
end = raw[live][:, :, 12].max(axis=1).astype(np.float64) / 100.0
tiles = raw[live][:, 0, 9]
t0, t1 = start.min(), end.max()
span = t1 - t0
print(f"{live.sum()} workgroups, {int(tiles.sum())} tiles ({tiles.min()}..{tiles.max()} per workgroup); "
      f"events {e0.elapsed_time(e1) * 1e3:.0f} us, first start to last end {span:.0f} us")
life = end - start
print(f"workgroup lifetime: mean {life.mean():.0f} us, min {life.min():.0f}, max {life.max():.0f}; "
      f"per tile {life.sum() / tiles.sum():.1f} us")
q = [50, 90, 99, 100]
print("start after the first start (us), percentiles", q, ":", np.percentile(start - t0, q).round(0))
print("end before the last end (us), percentiles", [0, 1, 10, 50], ":", np.percentile(t1 - end, [100, 99, 90, 50]).round(0))
hw = raw[live][:, 0, 13]
cu = ((hw >> 32) & 0xF) * 256 + ((hw >> 8) & 0xFF)   # xcc | se, sh, cu
ids, inv, per_cu = np.unique(cu, return_inverse=True, return_counts=True)
print(f"{len(ids)} compute units used; workgroups per unit: " + ", ".join(f"{c}: {int((per_cu == c).sum())} units" for c in np.unique(per_cu)))
for c in np.unique(per_cu):
    sel = per_cu[inv] == c
    print(f"  units with {c}: lifetime mean {life[sel].mean():.0f} us ({life[sel].min():.0f}..{life[sel].max():.0f})")
xcc = (hw >> 32) & 0xF
if os.environ.get("SKEW_DUMP"):
    np.savez(os.environ["SKEW_DUMP"], start=start - t0, end=end - t0, tiles=tiles, hw=hw, raw=raw[live])
print("  mean lifetime per XCD:", " ".join(f"{life[xcc == i].mean():.0f}" for i in np.unique(xcc)))
print(f"idle workgroup-time inside the span: {100 * (1 - life.sum() / (span * live.sum())):.1f} %")
