#!/bin/bash
# Per-kernel times of the memory-bound helpers and the statistics mode (run on the GPU box through gpurun):
#   bash profiles/collect_aux.sh <tag> [bench_aux args]
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}; shift || true
OUT=gpurun_out/aux_$TAG
mkdir -p $OUT
timeout -k 10 300 python3 tools/bench_aux.py "$@" > $OUT/bench_aux.jsonl 2> $OUT/log0.txt || { tail -5 $OUT/log0.txt; exit 1; }
cat $OUT/bench_aux.jsonl
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 tools/bench_aux.py --reps 4 "$@" > /dev/null 2> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
head -1 "$f" > $OUT/kernel_stats.csv
grep -E "cosine_tile|cosine_tri|normalize_blocks|component_scores|assemble|query_|gram_stats|topk_finalize" "$f" >> $OUT/kernel_stats.csv
cut -c1-160 $OUT/kernel_stats.csv
rm -rf $OUT/raw
