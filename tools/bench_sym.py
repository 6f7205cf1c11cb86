"""Time the plain and the symmetric whole-catalogue sweeps on the same matrix.
usage: bench_sym.py [N] [D] [K] [ITERS] [dense|hybrid]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import tvbf_amd  # noqa
from tvbingefriend_recommendation_service_amd import engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
kind = sys.argv[5] if len(sys.argv) > 5 else "hybrid"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(30001)
if kind == "dense":
    x = torch.randn((n, d), generator=g, device=dev)
    offsets, scales, ms = [0, d], [1.0], -1.0
else:
    x = torch.zeros((n, d), device=dev)
    x[:, :32] = (torch.rand((n, 32), generator=g, device=dev) < 0.08).float()
    t = torch.randn((n, d - 64), generator=g, device=dev).abs_()
    t *= torch.rand((n, d - 64), generator=g, device=dev) < 0.03
    x[:, 32:d - 32] = t
    del t
    base = d - 32
    r = torch.arange(n, device=dev)
    for width in (21, 5, 6):
        x[r, base + torch.randint(0, width, (n,), generator=g, device=dev)] = 1.0
        base += width
    offsets, scales, ms = [0, 32, d - 32, d], [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)], 0.1
z = engine.normalize_blocks(x, offsets, scales)
del x
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
res = {}
nbytes = engine.symmetric_workspace_bytes(n, z.shape[1], k)
ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
for name, fn in (("plain", lambda: engine.cosine_topk(z, k, ms)), ("symmetric", lambda: engine.cosine_topk_symmetric(z, k, ms, workspace=ws))):
    ts = []
    for _ in range(iters + 1):
        ev[0].record(); out = fn(); ev[1].record(); torch.cuda.synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    res[name] = out
    best = min(ts[1:])
    print(f"{name:10s} {best:8.2f} ms  (runs: {' '.join('%.1f' % v for v in ts)})  {2.0 * n * n * d / best / 1e9:.1f} TFLOP/s of 2 N^2 D", flush=True)
same = all(torch.equal(getattr(res["plain"], f), getattr(res["symmetric"], f)) for f in ("idx", "score", "count"))
print("identical lists:", same, engine.symmetric_inspect(ws, n, z.shape[1], k))
