"""Symmetric vs plain sweep on a catalogue with one large group of identical shows.
usage: clique_probe.py N CLIQUE [D] [K]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import tvbf_amd  # noqa
from tvbingefriend_recommendation_service_amd import engine

n, m = int(sys.argv[1]), int(sys.argv[2])
d = int(sys.argv[3]) if len(sys.argv) > 3 else 256
k = int(sys.argv[4]) if len(sys.argv) > 4 else 20
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(3)
x = torch.randn((n, d), generator=g, device=dev)
members = torch.randperm(n, device=dev)[:m]
x[members] = x[members[0]].clone()
z = engine.normalize_blocks(x, [0, d], [1.0])
nbytes = engine.symmetric_workspace_bytes(n, z.shape[1], k)
ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for name, fn in (("plain", lambda: engine.cosine_topk(z, k, -1.0)), ("symmetric", lambda: engine.cosine_topk_symmetric(z, k, -1.0, workspace=ws))):
    for _ in range(2):
        ev[0].record(); out = fn(); ev[1].record(); torch.cuda.synchronize()
    print(f"{name:10s} {ev[0].elapsed_time(ev[1]):8.2f} ms")
    if name == "plain":
        ref = out
print("identical:", all(torch.equal(getattr(ref, f), getattr(out, f)) for f in ("idx", "score", "count")),
      engine.symmetric_inspect(ws, n, z.shape[1], k))
