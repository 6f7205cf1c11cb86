import sys
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
z = engine.normalize_blocks(x, [0, 32, d - 32, d], [0.4 ** 0.5, 0.5 ** 0.5, 0.1 ** 0.5])
top = engine.catalogue_topk(z, k, 0.1)
idx = top.idx.long()
valid = idx >= 0
nb = idx.clamp(min=0)
lists_of_nb = idx[nb]                      # (n, k, k): the lists of my neighbours
me = torch.arange(n, device=dev).view(n, 1, 1)
mutual = ((lists_of_nb == me).any(dim=2) & valid)
print("pairs kept", int(valid.sum()), "mutual", int(mutual.sum()), "share", float(mutual.sum()) / float(valid.sum()))
