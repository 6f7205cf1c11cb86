#!/bin/bash
# Effective clock and MFMA-pipe occupancy of the fused kernel: one rocprofv3 --pmc pass.
# usage (on the GPU box): bash profiles/pmc_clock.sh <tag> [bench args]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}; shift || true
OUT=gpurun_out/pmc_clock_$TAG
mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if ("cosine_tile" in r["Kernel_Name"] or "cosine_topk_dma" in r["Kernel_Name"]):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
vals = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if ("cosine_tile" in r["Kernel_Name"] or "cosine_topk_dma" in r["Kernel_Name"]):
        vals[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
for k, v in vals.items():
    t = dur.get(k)
    if not t: continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    print(f"dispatch {k}: {t*1e3:.2f} ms  clock {cyc/t/1e9:.3f} GHz  mfma_pipe_busy {v['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*cyc):.3f}  "
          f"wait_any {v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']:.3f} wait_inst {v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']:.3f} active {v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES']:.3f}")
PY
