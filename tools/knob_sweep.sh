#!/bin/bash
# Step time of the default workload under a few tuning switches (run on the GPU box through gpurun):
#   bash tools/knob_sweep.sh [bench args]
export TVBF_TUNING=1 TMPDIR=/tmp
run() { printf "%-28s " "$1"; env $1 timeout -k 10 200 python bench.py --steps 24 --warmup 3 --no-cpu-baseline --no-fp32-compare --no-end-to-end "${@:2}" 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(l['ms_per_step'],3), l['phases']['sweeps_ms'], l['phases']['rerank_ms'])"; }
run "X=0" "$@"
for v in 16 24 48 64; do run "TVBF_SYM_SHARDS=$v" "$@"; done
for v in 64 256; do run "TVBF_COL_DRAIN=$v" "$@"; done
for v in 34 36; do run "TVBF_CAP=$v" "$@"; done
for v in 64 256; do run "TVBF_COL_WINDOW=$v" "$@"; done
run "TVBF_SYM_OWN=fused" "$@"
run "X=1" "$@"
