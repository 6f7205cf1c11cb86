"""Per-rank device time of the symmetric sharded sweep, all ranks run one after the other on ONE
GPU in lockstep (collectives replaced by indexing, so communication is not in the numbers).
usage: simulate_sym_ranks.py N D K WORLD [matrix]   ("matrix": own blocks through the stored-matrix route)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import tvbf_amd  # noqa
from tvbingefriend_recommendation_service_amd import engine, sharding

n, d, k, world = (int(v) for v in sys.argv[1:5])
use_matrix = len(sys.argv) > 5 and sys.argv[5] == "matrix"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(30001)
x = torch.zeros((n, d), device=dev)
x[:, :32] = (torch.rand((n, 32), generator=g, device=dev) < 0.08).float()
t = torch.randn((n, d - 64), generator=g, device=dev).abs_()
t *= torch.rand((n, d - 64), generator=g, device=dev) < 0.03
x[:, 32:d - 32] = t
del t
base = d - 32
r_ = torch.arange(n, device=dev)
for width in (21, 5, 6):
    x[r_, base + torch.randint(0, width, (n,), generator=g, device=dev)] = 1.0
    base += width
ms = 0.1
z = engine.normalize_blocks(x, [0, 32, d - 32, d], [float(np.sqrt(w)) for w in (0.4, 0.5, 0.1)])
del x
shards = [sharding.shard_bounds(n, world, r) for r in range(world)]
per = shards[0].per_rank
z_all = torch.zeros((per * world, z.shape[1]), device=dev)
z_all[:n] = z
del z

def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = fn(); e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1)

for rep in range(2):
    def buf(sh):
        nb = engine.own_matrix_bytes(sh.rows, z_all.shape[1]) if use_matrix else 0
        return torch.empty(nb // 8 + 1, dtype=torch.int64, device=dev) if nb else None
    jobs = [sharding.SymmetricShardJob(z_all[r * per:(r + 1) * per], sh, k, ms, own_matrix=buf(sh)) for r, sh in enumerate(shards)]
    t_own = [timed(j.phase_own)[1] for j in jobs]
    bounds_all = torch.cat([j.own_bounds() for j in jobs])
    t_half = [timed(lambda j=j: j.phase_half(z_all, bounds_all))[1] for j in jobs]
    sends, t_red = zip(*[timed(lambda j=j: j.lists_for_peers(bounds_all)) for j in jobs])
    flag = torch.stack([j.flag for j in jobs]).max(dim=0).values.contiguous()
    t_fin = []
    for r, j in enumerate(jobs):
        recv = torch.stack([sends[p][r] for p in range(world)]).contiguous()
        t_fin.append(timed(lambda: j.finish(z_all, recv, flag))[1])
tot = [a + b + c + e for a, b, c, e in zip(t_own, t_half, t_red, t_fin)]
for r in range(world):
    print(f"rank {r}: own {t_own[r]:.2f}  half {t_half[r]:.2f}  reduce {t_red[r]:.2f}  finish {t_fin[r]:.2f}  total {tot[r]:.2f} ms")
print(f"world {world}: slowest rank {max(tot):.2f} ms, mean {sum(tot) / world:.2f} ms, overflow flag {int(flag[0].item())}, flagged shows {int((flag[4:] != 0).sum().item())}")
