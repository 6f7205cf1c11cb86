#!/usr/bin/env python3
"""Entries per show that the column side of the symmetric sweep buffered (hybrid benchmark data):

    python tools/colside_hist.py [n] [d] [k]

Reads the per-show counters behind the lists in the workspace of engine.cosine_topk_symmetric
(layout: [bounds and list slots][16 bytes of summary][n flags][n counters][n x cap keys], cap = 1024, or 2048 for k > 64)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import tvbf_amd  # noqa: F401
from tvbingefriend_recommendation_service_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(30001)
x = torch.zeros((n, d), device=dev)
x[:, :32] = (torch.rand((n, 32), generator=g, device=dev) < 0.08).float()
t_ = torch.randn((n, d - 64), generator=g, device=dev).abs_()
t_ *= torch.rand((n, d - 64), generator=g, device=dev) < 0.03
t_ /= t_.norm(dim=1, keepdim=True).clamp_min(1e-30)
x[:, 32:d - 32] = t_
del t_
base = d - 32
r = torch.arange(n, device=dev)
for width in (21, 5, 6):
    x[r, base + torch.randint(0, width, (n,), generator=g, device=dev)] = 1.0
    base += width
z = engine.normalize_blocks(x, [0, 32, d - 32, d], [0.4 ** 0.5, 0.5 ** 0.5, 0.1 ** 0.5])
del x
nbytes = engine.symmetric_workspace_bytes(n, z.shape[1], k)
ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
engine.cosine_topk_symmetric(z, k, 0.1, workspace=ws)
torch.cuda.synchronize()
cnt_bytes = (n * 4 + 15) // 16 * 16
cap = 1024 if k <= 64 else 2048
state_bytes = 16 + 2 * cnt_bytes + n * cap * 8
off = nbytes - state_bytes + 16 + cnt_bytes
cnt = ws.view(torch.uint8)[off:off + 4 * n].view(torch.int32).cpu().numpy()
print(f"n={n} d={d} k={k}: {cnt.sum()} entries, mean {cnt.mean():.1f}, max {cnt.max()}")
edges = [0, 32, 64, 128, 192, 256, 384, 512, 640, 768, 896, 1024, 2048, 1 << 30]
h, _ = np.histogram(cnt, bins=edges)
for a, b, c in zip(edges, edges[1:], h):
    print(f"  [{a:5d}, {b if b < 1 << 30 else 'inf':>5}) {c:7d} shows  {100 * c / n:5.1f} %   {100 * cnt[(cnt >= a) & (cnt < b)].sum() / max(cnt.sum(), 1):5.1f} % of entries")
print(engine.symmetric_inspect(ws, n, z.shape[1], k))
