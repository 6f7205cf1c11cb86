import sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import tvbf_amd
from tvbingefriend_recommendation_service_amd import engine
dev = torch.device("cuda:0")
n = 100_000
x = (torch.rand((n, 32), device=dev) < 0.08).float()
x[torch.arange(n), torch.randint(0, 32, (n,), device=dev)] = 1.0
z32 = engine.normalize_blocks(x, [0, 32], [1.0])
z64 = torch.zeros((n, 64), device=dev); z64[:, :32] = z32
for name, z in (("ld=32 (staged, radix)", z32), ("ld=64 (DMA, band)", z64)):
    out = engine.gram_stats(z); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): out = engine.gram_stats(z)
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 2 * 1e3, 2), "ms", out.cpu().numpy().tolist())
