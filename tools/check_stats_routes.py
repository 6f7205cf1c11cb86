import os, sys, json
os.environ.setdefault("TVBF_TUNING", "1")  # the library looks at its TVBF_* switches only then
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import tvbf_amd
from tvbingefriend_recommendation_service_amd import engine
dev = torch.device("cuda:0")
n, d = 100_000, 1024
x = torch.zeros((n, d), device=dev)
x[:, :32] = (torch.rand((n, 32), device=dev) < 0.08).float()
t_ = torch.randn((n, d - 64), device=dev).abs_()
t_ *= torch.rand((n, d - 64), device=dev) < 0.03
x[:, 32:d - 32] = t_
x[torch.arange(n, device=dev), d - 32 + torch.randint(0, 32, (n,), device=dev)] = 1.0
z = engine.normalize_blocks(x, [0, 32, d - 32, d], [0.4 ** 0.5, 0.5 ** 0.5, 0.1 ** 0.5])
info = {}
band = engine.gram_stats(z, info=info).cpu().numpy()
os.environ["TVBF_STATS_ROUTE"] = "radix"
radix = engine.gram_stats(z).cpu().numpy()
print("band ", band.tolist()); print("radix", radix.tolist()); print(info)
print("EQUAL" if np.array_equal(band, radix) else "DIFFERENT")
for kind in ("dense",):
    del os.environ["TVBF_STATS_ROUTE"]
    z = engine.normalize_blocks(torch.randn((n, d), device=dev), [0, d], [1.0])
    info = {}
    band = engine.gram_stats(z, info=info).cpu().numpy()
    os.environ["TVBF_STATS_ROUTE"] = "radix"
    radix = engine.gram_stats(z).cpu().numpy()
    print(kind, "EQUAL" if np.array_equal(band, radix) else "DIFFERENT", info, band.tolist())
