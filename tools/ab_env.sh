#!/bin/bash
export TVBF_TUNING=1  # the library looks at its TVBF_* switches only in a process started with this set
# A/B of environment settings on one box: bash tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...  ("-" = no setting)
set -e
ARGS=$1; shift
for rep in 1 2; do
  for setting in "$@"; do
    echo -n "$setting: "
    if [ "$setting" = "-" ]; then env python bench.py --no-cpu-baseline $ARGS; else env $setting python bench.py --no-cpu-baseline $ARGS; fi | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['roofline']['kernel_ms'],2), round(d['roofline']['frac'],4))"
  done
done
