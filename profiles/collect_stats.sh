#!/bin/bash
# Per-kernel time summary of the default bench workload (run on the GPU box through gpurun):
#   bash profiles/collect_stats.sh <tag> [bench args]
# writes gpurun_out/stats_<tag>/kernel_stats.csv (own kernels only) and bench.json.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-x}; shift || true
OUT=gpurun_out/stats_$TAG
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-compare "$@" > $OUT/bench.json 2> $OUT/log.txt || { tail -5 $OUT/log.txt; exit 1; }
f=$(find $OUT/raw -name "*kernel_stats.csv" | head -1)
head -1 "$f" > $OUT/kernel_stats.csv
grep -E "cosine_tile|cosine_topk_dma|cosine_tri_dma|matrix_rows_topk|refine_min|topk_finalize|normalize_blocks|component_scores|bound_init|assemble|query_|gram_stats|col_reduce|bound_merge|fill_u32|zero16|rescore_topk|block_norms|score_rows|dense_rows_topk" "$f" >> $OUT/kernel_stats.csv
cat $OUT/kernel_stats.csv
tail -1 $OUT/bench.json | cut -c1-400
rm -rf $OUT/raw
