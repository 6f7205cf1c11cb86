#!/usr/bin/env python3
"""Fold the raw counter CSVs of profiles/collect_pmc.sh (gpurun_out/pmc_<tag>/) and the kernel
time of profiles/collect_stats.sh (gpurun_out/stats_<tag>/) into the JSON summaries kept under
profiles/:   python profiles/summarize_pmc.py <tag>"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

tag = sys.argv[1]
root = Path(__file__).resolve().parent.parent
pmc = root / "gpurun_out" / f"pmc_{tag}"
stats = root / "gpurun_out" / f"stats_{tag}"
KERNEL = "cosine_topk_dma_kernel<false>"
res = {}
for name in ("sq1", "sq2", "tcc1", "tcc2"):
    f = glob.glob(f"{pmc}/{name}/**/*counter_collection.csv", recursive=True)
    agg, calls = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            calls[r["Counter_Name"]] += 1
    res[name] = {c: v / calls[c] for c, v in agg.items()}
ms = None
for r in csv.DictReader(open(stats / "kernel_stats.csv")):
    if KERNEL in r["Name"]:
        ms = float(r["AverageNs"]) / 1e6
sq1 = res["sq1"]
cyc = sq1["GRBM_GUI_ACTIVE"] / 8
fetch = res["tcc1"]["FETCH_SIZE"] * 1024 * 2
out = {
    "kernel": KERNEL, "workload": "N=100000 D=1024 K=20 hybrid-like, 1 GPU",
    "source": f"rocprofv3 --pmc, one pass per group (profiles/collect_pmc.sh {tag}); kernel time from profiles/collect_stats.sh {tag}",
    **res,
    "derived": {
        "kernel_ms_rocprof_avg": round(ms, 3),
        "achieved_TFLOPs": round(2 * 1e5 * 1e5 * 1024 / (ms * 1e-3) / 1e12, 2),
        "effective_clock_GHz": round(cyc / (ms * 1e-3) / 1e9, 3),
        "mfma_pipe_busy_frac": round(sq1["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 4),
        "mfma_wave_instructions": res["sq2"]["SQ_INSTS_MFMA"],
        "expected_mfma_wave_instructions": 782 * 782 * 4 * 2048,
        "non_mfma_valu_per_wave_tile": round((sq1["SQ_INSTS_VALU"] - res["sq2"]["SQ_INSTS_MFMA"]) / (782 * 782 * 4), 1),
        "l2_hit_rate": round(res["tcc1"]["TCC_HIT_sum"] / (res["tcc1"]["TCC_HIT_sum"] + res["tcc2"]["TCC_MISS_sum"]), 3),
        "fabric_read_bytes_corrected": int(fetch), "fabric_write_bytes": int(res["tcc2"]["WRITE_SIZE"] * 1024),
        "note": "FETCH_SIZE (KB) doubled as the gfx950 guide prescribes for 16-byte-per-lane streaming reads; Infinity-Cache hits are included in it",
        "lds_bank_conflict_frac": round(sq1["SQ_LDS_BANK_CONFLICT"] / res["sq2"]["SQ_LDS_IDX_ACTIVE"], 5),
    },
}
prof = root / "profiles"
json.dump(out, open(prof / "r01_pmc_n100k_d1024_k20.json", "w"), indent=1)
shutil.copy(stats / "kernel_stats.csv", prof / "r01_kernel_stats_n100k_d1024_k20.csv")
(prof / "r01_bench_n100k_d1024_k20_under_rocprof.json").write_text(
    (stats / "bench.json").read_text().strip().splitlines()[-1] + "\n")
tr = json.load(open(prof / "traffic.json"))
tr["n100000_d1024_k20_hybrid_g1"] = int(fetch + res["tcc2"]["WRITE_SIZE"] * 1024)
json.dump(tr, open(prof / "traffic.json", "w"), indent=1)
print(json.dumps(out["derived"], indent=1))
