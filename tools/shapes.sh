#!/bin/bash
export TVBF_TUNING=1  # the library looks at its TVBF_* switches only in a process started with this set
# Kernel time / MFMA fraction of the fused kernel for the shapes quoted in the design notes.
set -e
for args in "" "--k 50" "--k 100" "--shows 10000 --dims 512 --data dense --steps 20" "--shows 30000 --dims 576" "--data dense"; do
  echo -n "[$args] "; timeout -k 10 180 python bench.py --no-cpu-baseline --steps 5 $args | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['kernel_ms'],3), 'ms', round(d['roofline']['achieved'],1), 'TF', round(d['roofline']['frac'],4), 'step', round(d['ms_per_step'],3))"
done
