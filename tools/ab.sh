#!/bin/bash
export TVBF_TUNING=1  # the library looks at its TVBF_* switches only in a process started with this set
# A/B of library builds on one box: bash tools/ab.sh "<bench args>" lib1.so lib2.so ...  ("-" = the in-tree library)
set -e
ARGS=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset TVBF_LIB; else export TVBF_LIB=$lib; fi
    echo -n "$lib: "
    timeout -k 10 180 python bench.py --no-cpu-baseline $ARGS | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['kernel_ms'],2), round(d['roofline']['frac'],4))"
  done
done
