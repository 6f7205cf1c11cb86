"""Time of the float64 re-rank beside the fp32 call at one catalogue size (HIP events).
Usage: python tools/time_rerank.py [n] [d] [k] [kind]"""

from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]


def timed(torch, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


def main():
    import torch

    import tvbf_amd  # noqa: F401
    from test_gpu_configs import _hybrid_chunk
    from tvbingefriend_recommendation_service_amd import engine

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    kind = sys.argv[4] if len(sys.argv) > 4 else "hybrid"
    dev = torch.device("cuda:0")
    if kind == "hybrid":
        offsets, weights, groups, ms = [0, 32, d - 32, d], [0.4, 0.5, 0.1], [0, 1, 2], 0.1
        x = _hybrid_chunk(torch, dev, n, d, 30003)
    else:
        offsets, weights, groups, ms = [0, d], [1.0, 0.0, 0.0], [0], -1.0
        x = torch.randn((n, d), device=dev)
    nb = len(offsets) - 1
    scales = [float(np.sqrt(w)) for w in weights[:nb]]
    src = [engine.DenseSource(x, 0)]
    z = engine.normalize_blocks(x, offsets, scales)
    kc = engine.candidate_count(k)
    t_k = timed(torch, lambda: engine.catalogue_topk(z, k, ms))
    t_kc = timed(torch, lambda: engine.catalogue_topk(z, kc, ms - engine.THRESHOLD_BAND))
    top = engine.catalogue_topk(z, k, ms)
    t_k5 = timed(torch, lambda: engine.component_scores(z, offsets, scales, top.idx))
    t_norms = timed(torch, lambda: engine.block_norms(src, n, offsets, check_finite=False))
    norms = engine.block_norms(src, n, offsets)
    cand = engine.catalogue_topk(z, kc, ms - engine.THRESHOLD_BAND)
    out = engine.TopK64(torch.empty((n, k), dtype=torch.int32, device=dev), torch.empty((n, k), dtype=torch.float64, device=dev),
                        torch.empty((n,), dtype=torch.int32, device=dev), torch.empty((n, k, 3), dtype=torch.float64, device=dev), {})
    flagged = torch.zeros(1 + n, dtype=torch.int32, device=dev)

    def rescore():
        flagged.zero_()
        engine.rescore_topk(src, n, offsets, groups, weights, norms, cand, k, ms, out, flagged=flagged)

    t_rescore = timed(torch, rescore)
    stats = {}

    def whole():
        stats.update(engine.exact_topk(z, src, offsets, groups, weights, k, ms, norms=norms).stats)

    t_exact = timed(torch, whole, reps=3)
    for kc_try in (24, 26, 28, 30):
        st3 = {}

        def run():
            c = engine.catalogue_topk(z, kc_try, ms - engine.THRESHOLD_BAND)
            st3.update(engine.exact_topk(z, src, offsets, groups, weights, k, ms, norms=norms, cand=c).stats)

        t = timed(torch, run, reps=3)
        print(f"  sweep for {kc_try} candidates + re-rank: {t:.2f} ms, stages {st3}")
    share = n // 8  # one rank's rows of an 8-rank job: the re-rank stages after the sweep
    cand_share = engine.cosine_topk(z, kc, ms - engine.THRESHOLD_BAND, q_row0=0, nq=share)
    st4 = {}
    t = timed(torch, lambda: st4.update(engine.exact_topk(z, src, offsets, groups, weights, k, ms, norms=norms, q_row0=0,
                                                            nq=share, cand=cand_share).stats), reps=5)
    print(f"  re-rank of {share} rows (one rank of eight): {t:.2f} ms, stages {st4}")
    for rk in (100,):  # second stage as a plain sweep of the flagged rows for 100 candidates instead of the band kernel
        st2 = {}
        t = timed(torch, lambda: st2.update(engine.exact_topk(z, src, offsets, groups, weights, k, ms, norms=norms, resweep_k=rk).stats), reps=3)
        print(f"  second stage with {rk} candidates: {t:.2f} ms, stages {st2}")
    print(f"n={n} d={d} k={k} {kind}: fp32 call K={k} {t_k:.2f} ms, K'={kc} {t_kc:.2f} ms, K5 {t_k5:.2f} ms, "
          f"norms {t_norms:.3f} ms, rescore launch {t_rescore:.2f} ms, exact_topk (all stages) {t_exact:.2f} ms, stages {stats}")


if __name__ == "__main__":
    main()
