#!/bin/bash
# Kernel-time summaries (rocprofv3 --kernel-trace --stats) of the workloads quoted in the design notes,
# one directory per workload under gpurun_out/stats_<round>_<name>/ (run on the GPU box through gpurun):
#   bash profiles/collect_round.sh r02
set -u
R=${1:-rXX}
run() { name=$1; shift; bash profiles/collect_stats.sh ${R}_$name "$@" > gpurun_out/${R}_$name.txt 2>&1 || { echo "$name failed"; tail -3 gpurun_out/${R}_$name.txt; }; }
run n100k_d1024_k20_symmetric
run n100k_d1024_k20_plain --sweep plain
run n100k_d1024_k50_symmetric --k 50
run n100k_d1024_k100_symmetric --k 100
run n100k_d1024_k20_dense_symmetric --data dense
run n10k_d512_k20_dense --shows 10000 --dims 512 --data dense --steps 20
run n30k_d576_k20 --shows 30000 --dims 576
run n80k_d576_k20 --shows 80000 --dims 576
for f in gpurun_out/${R}_*.txt; do echo "== $f"; cut -c1-180 $f | grep -E "cosine|matrix_rows|metric" | cut -c1-180; done
