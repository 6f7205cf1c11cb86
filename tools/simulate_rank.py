"""Time the per-rank work of a row-sharded job on ONE GPU (no communication): the rank's own
columns first, then the rest, as sharding.sharded_topk does.  usage: simulate_rank.py N D K WORLD RANK [ITERS] [CHECK_ROWS]
With CHECK_ROWS > 0 that many of the rank's rows are compared with a float64 product on the GPU
(scores within 1e-5, indices equal wherever neighbouring scores differ by more than 2e-5)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import tvbf_amd  # noqa
from tvbingefriend_recommendation_service_amd import engine, sharding

n, d, k, world, rank = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 4
check_rows = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((n, d), generator=g, device=dev)
z = engine.normalize_blocks(x, [0, d], [1.0])
del x
sh = sharding.shard_bounds(n, world, rank)
own, after, before = sharding.column_pieces(sh)
zq = z[sh.row0:sh.row0 + sh.rows].contiguous()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for it in range(iters):
    ev[0].record()
    ses = engine.TopkSession(zq, sh.rows, sh.row0, k, -1.0)
    s1 = ses.accumulate((zq, own[1], own[0]))
    ev[1].record()
    s2 = ses.accumulate((z[after[0]:], after[1], after[0]), (z, before[1], before[0]))
    ev[2].record()
    top = ses.finish()
    ev[3].record()
    torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
flops = 2.0 * sh.rows * n * d
print(f"rank {rank}/{world}: rows {sh.rows}  own {t[0]:.2f} ms ({s1} splits)  rest {t[1]:.2f} ms ({s2} splits)  merge {t[2]:.2f} ms"
      f"  total {sum(t):.2f} ms  {flops / sum(t) / 1e9:.1f} TFLOP/s")

if check_rows:
    pick = torch.linspace(0, sh.rows - 1, check_rows, device=dev).long()
    bad = 0
    for r in pick.tolist():
        h = (z.double() @ zq[r].double()) if n * d <= 2 ** 28 else torch.cat(
            [z[c:c + 65536].double() @ zq[r].double() for c in range(0, n, 65536)])
        h[sh.row0 + r] = -float("inf")
        ref_s, ref_i = torch.topk(h, k + 1)
        got_i, got_s = top.idx[r].long(), top.score[r].double()
        assert int(top.count[r]) == k
        assert float((got_s - h[got_i]).abs().max()) <= 1e-5, "score error"
        assert float((got_s - ref_s[:k]).abs().max()) <= 1e-5, "rank-wise score error"
        gaps = (ref_s[:-1] - ref_s[1:]) > 2e-5
        clear = gaps[:k].clone()
        clear[1:] &= gaps[: k - 1]
        bad += int((got_i[clear] != ref_i[:k][clear]).sum())
    assert bad == 0, f"{bad} index mismatches away from ties"
    print(f"checked {check_rows} rows against float64: scores <= 1e-5, indices exact away from ties")
