"""Random shapes and parameters: symmetric sweep (single catalogue and sharded job in lockstep)
against the plain sweep, bitwise.  usage: fuzz_sym.py [CASES] [SEED] [MAX_N]"""
import os, sys
os.environ.setdefault("TVBF_TUNING", "1")  # the library looks at its TVBF_* switches only then
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
import numpy as np, torch
import tvbf_amd  # noqa
from tvbingefriend_recommendation_service_amd import engine, sharding

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 7000
dev = torch.device("cuda:0")


def same(a, b):
    return all(torch.equal(getattr(a, f), getattr(b, f)) for f in ("idx", "score", "count"))


def lockstep(z, n, world, k, ms):
    shards = [sharding.shard_bounds(n, world, r) for r in range(world)]
    per = shards[0].per_rank
    z_all = torch.zeros((per * world, z.shape[1]), device=dev)
    z_all[:n] = z[:n]
    jobs = [sharding.SymmetricShardJob(z_all[r * per:(r + 1) * per], sh, k, ms) for r, sh in enumerate(shards)]
    for j in jobs:
        j.phase_own()
    bounds_all = torch.cat([j.own_bounds() for j in jobs])
    for j in jobs:
        j.phase_half(z_all, bounds_all)
    sends = [j.lists_for_peers(bounds_all) for j in jobs]
    flag = torch.stack([j.flag for j in jobs]).max(dim=0).values.contiguous()
    tops = [j.finish(z_all, torch.stack([sends[p][r] for p in range(world)]).contiguous(), flag)
            for r, j in enumerate(jobs)]
    cat = lambda f: torch.cat([getattr(t, f)[: sh.rows] for t, sh in zip(tops, shards)])  # noqa: E731
    return engine.TopK(cat("idx"), cat("score"), cat("count"))


bad = 0
for c in range(cases):
    n = int(rng.integers(300, max_n))
    d = 64 * int(rng.integers(1, 6))
    k = int(rng.choice([1, 3, 7, 20, 24, 30, 31, 50, 100]))
    ms = float(rng.choice([-1.0, 0.0, 0.05, 0.3]))
    kind = rng.choice(["dense", "lumpy", "clique"])
    x = torch.from_numpy(rng.standard_normal((n, d)).astype(np.float32))
    if kind == "lumpy":      # few distinct values: masses of exactly tied scores
        x = torch.from_numpy((rng.random((n, d)) < 0.1).astype(np.float32) * rng.integers(1, 3, (n, d)).astype(np.float32))
    elif kind == "clique":
        who = rng.permutation(n)[: int(rng.integers(2, max(3, n // 3)))]
        x[who] = x[who[0]].clone()
    z = engine.normalize_blocks(x.to(dev), [0, d], [1.0])
    shards = int(rng.integers(2, 12))
    os.environ["TVBF_SYM_SHARDS"] = str(shards)
    plain = engine.cosine_topk(z, k, ms)
    ok = True
    if engine.symmetric_workspace_bytes(n, d, k):
        ok &= same(plain, engine.cosine_topk_symmetric(z, k, ms))
    world = int(rng.integers(2, 9))
    os.environ.pop("TVBF_SYM_SHARDS")
    if sharding.symmetric_route_fits(world, k, d):
        ok &= same(plain, lockstep(z, n, world, k, ms))
    torch.cuda.synchronize()
    print(f"case {c}: n={n} d={d} k={k} ms={ms} {kind} shards={shards} world={world}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
