import os, sys, time
os.environ.setdefault("TVBF_TUNING", "1")  # the library looks at its TVBF_* switches only then
sys.path.insert(0, "/root/repo")
import torch
import tvbf_amd
from tvbingefriend_recommendation_service_amd import engine
dev = torch.device("cuda:0")
n, d, k = 100_000, 1024, 20
x = torch.zeros((n, d), device=dev)
x[:, :32] = (torch.rand((n, 32), device=dev) < 0.08).float()
t_ = torch.randn((n, d - 64), device=dev).abs_()
t_ *= torch.rand((n, d - 64), device=dev) < 0.03
x[:, 32:d - 32] = t_
x[torch.arange(n, device=dev), d - 32 + torch.randint(0, 32, (n,), device=dev)] = 1.0
offsets, scales = [0, 32, d - 32, d], [0.4 ** 0.5, 0.5 ** 0.5, 0.1 ** 0.5]
z = engine.normalize_blocks(x, offsets, scales)
top = engine.catalogue_topk(z, k, 0.1)
def run():
    return engine.component_scores(z, offsets, scales, top.idx)
def timed():
    run(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): c = run()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e3, c
ms1, c1 = timed()
os.environ["TVBF_K5_NO_MUTUAL"] = "1"
ms0, c0 = timed()
print("mutual once %.3f ms, every pair %.3f ms, equal %s" % (ms1, ms0, torch.equal(c0, c1)))
# a shard's rows inside the catalogue
del os.environ["TVBF_K5_NO_MUTUAL"]
r0, nq = 37_500, 12_500
a = engine.component_scores(z, offsets, scales, top.idx[r0:r0 + nq].contiguous(), q_row0=r0)
print("shard equal", torch.equal(a, c0[r0:r0 + nq]))
