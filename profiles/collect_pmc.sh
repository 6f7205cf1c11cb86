#!/bin/bash
# Counter passes for the fused kernel (run on the GPU box through gpurun).
# Each pass is its own rocprofv3 run with --pmc only; the profiled program is python3 itself.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_${1:-r01}
shift || true
ARGS="${*:---steps 1 --warmup 0 --no-cpu-baseline --no-fp32-compare}"
mkdir -p $OUT
pass() { name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 bench.py $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$name" <<'PY'
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-60:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if "at::native" in k or "rocclr" in k: continue
    print(name, k, {c: v / calls[(k, c)] for c, v in d.items()})
PY
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE &&
pass sq2 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 &&
pass tcc1 FETCH_SIZE TCC_HIT_sum &&
pass tcc2 WRITE_SIZE TCC_MISS_sum
