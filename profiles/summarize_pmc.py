#!/usr/bin/env python3
"""Fold the raw counter CSVs of profiles/collect_pmc.sh (gpurun_out/pmc_<tag>/) and the kernel
time of profiles/collect_stats.sh (gpurun_out/stats_<tag>/) into the JSON summaries kept under
profiles/:   python profiles/summarize_pmc.py <tag> plain|symmetric|matrix [--round r02 --n .. --d .. --k .. --data ..]

plain:     the general sweep, cosine_topk_dma_kernel<false, 0, false> (bench.py --sweep plain)
symmetric: the half sweep <false, 2, true> plus the shards' own blocks (cosine_tri_dma_kernel<0> with a table of
           diagonal blocks + matrix_rows_topk_kernel<true>) of one call (bench.py default on one GPU); their
           counters and times are added up per call.
matrix:    the stored-matrix route of small catalogues: cosine_tri_dma_kernel<0> + matrix_rows_topk_kernel."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

import argparse

ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("mode", choices=["plain", "symmetric", "matrix"])
ap.add_argument("--round", default="r03")
ap.add_argument("--n", type=int, default=100_000)
ap.add_argument("--d", type=int, default=1024)
ap.add_argument("--k", type=int, default=20)
ap.add_argument("--data", default="hybrid")
args = ap.parse_args()
tag, mode = args.tag, args.mode
root = Path(__file__).resolve().parent.parent
pmc = root / "gpurun_out" / f"pmc_{tag}"
stats = root / "gpurun_out" / f"stats_{tag}"
lg = "true" if args.k > 30 else "false"  # lists in global memory above K = 30
KERNELS = {"plain": [f"cosine_topk_dma_kernel<{lg}, 0, false>"],
           "symmetric": [f"cosine_topk_dma_kernel<{lg}, 2, true>", "cosine_tri_dma_kernel<0>", "matrix_rows_topk_kernel<true>"],
           "matrix": ["cosine_tri_dma_kernel<0>", "matrix_rows_topk_kernel"]}[mode]
suffix = {"plain": "", "symmetric": "_symmetric", "matrix": "_matrix"}[mode]
wl = f"n{args.n // 1000}k_d{args.d}_k{args.k}" + ("" if args.data == "hybrid" else f"_{args.data}")
res = {}
for name in ("sq1", "sq2", "tcc1", "tcc2"):
    # the newest file: a directory merged from several collections keeps the older ones
    f = sorted(glob.glob(f"{pmc}/{name}/**/*counter_collection.csv", recursive=True), key=lambda x: Path(x).stat().st_mtime, reverse=True)
    agg, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        for kern in KERNELS:
            if kern in r["Kernel_Name"]:
                agg[(kern, r["Counter_Name"])] += float(r["Counter_Value"])
                calls[(kern, r["Counter_Name"])] += 1
    tot = collections.defaultdict(float)
    for (kern, c), v in agg.items():
        # per call of the sweep: the first kernel is launched once per call, the half sweep of the
        # symmetric route once or twice (inside the shards, then over them)
        tot[c] += v / calls[(KERNELS[0], c)]
    res[name] = dict(tot)
ms, per_kernel, sweep_calls = 0.0, {}, None
rows = list(csv.DictReader(open(stats / "kernel_stats.csv")))
for r in rows:
    if KERNELS[0] in r["Name"]:
        sweep_calls = int(r["Calls"])
for r in rows:
    for kern in KERNELS:
        if kern in r["Name"]:
            per_call = float(r["TotalDurationNs"]) / sweep_calls / 1e6
            per_kernel[kern] = {"ms_per_call": round(per_call, 3), "launches_per_call": int(r["Calls"]) // sweep_calls}
            ms += per_call
sq1 = res["sq1"]
cyc = sq1["GRBM_GUI_ACTIVE"] / 8
fetch = res["tcc1"]["FETCH_SIZE"] * 1024 * 2
mfma = res["sq2"]["SQ_INSTS_MFMA"]
out = {
    "kernels": KERNELS, "workload": f"N={args.n} D={args.d} K={args.k} {args.data}, 1 GPU, sweep = {mode}",
    "source": f"rocprofv3 --pmc, one pass per group (profiles/collect_pmc.sh {tag}); kernel time from profiles/collect_stats.sh {tag}",
    **res,
    "derived": {
        "kernel_ms_rocprof_avg": round(ms, 3), "kernel_ms_each": per_kernel,
        "TFLOPs_of_2N2D": round(2.0 * args.n * args.n * args.d / (ms * 1e-3) / 1e12, 2),
        "TFLOPs_executed": round(mfma * 4096 / (ms * 1e-3) / 1e12, 2),
        "share_of_all_ordered_pairs_computed": round(mfma / (((args.n + 127) // 128) ** 2 * 4 * (args.d // 32) * 64), 4),
        "effective_clock_GHz": round(cyc / (ms * 1e-3) / 1e9, 3),
        "mfma_pipe_busy_frac": round(sq1["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4),
        "mfma_wave_instructions": mfma,
        "non_mfma_valu_per_wave_tile": round((sq1["SQ_INSTS_VALU"] - mfma) / (mfma / ((args.d // 32) * 64)), 1),
        "l2_hit_rate": round(res["tcc1"]["TCC_HIT_sum"] / (res["tcc1"]["TCC_HIT_sum"] + res["tcc2"]["TCC_MISS_sum"]), 3),
        "fabric_read_bytes_corrected": int(fetch), "fabric_write_bytes": int(res["tcc2"]["WRITE_SIZE"] * 1024),
        "note": "FETCH_SIZE (KB) doubled as the gfx950 guide prescribes for 16-byte-per-lane streaming reads; Infinity-Cache hits are included in it; "
                "one MFMA wave instruction = 32x32x2 multiply-adds = 4096 flops",
        "lds_bank_conflict_frac": round(sq1["SQ_LDS_BANK_CONFLICT"] / res["sq2"]["SQ_LDS_IDX_ACTIVE"], 5),
    },
}
prof = root / "profiles"
json.dump(out, open(prof / f"{args.round}_pmc_{wl}{suffix}.json", "w"), indent=1)
shutil.copy(stats / "kernel_stats.csv", prof / f"{args.round}_kernel_stats_{wl}{suffix}.csv")
(prof / f"{args.round}_bench_{wl}{suffix}_under_rocprof.json").write_text(
    (stats / "bench.json").read_text().strip().splitlines()[-1] + "\n")
tr = json.load(open(prof / "traffic.json"))
tr[f"n{args.n}_d{args.d}_k{args.k}_{args.data}_g1" + ("" if mode == "plain" else "_sym")] = int(fetch + res["tcc2"]["WRITE_SIZE"] * 1024)
tr["_source"] = "profiles/r0x_pmc_*.json (fabric bytes of the sweep kernels of one call; the newest round's value per workload)"
json.dump(tr, open(prof / "traffic.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
