#!/bin/bash
export TVBF_TUNING=1  # the library looks at its TVBF_* switches only in a process started with this set
# A/B of library variants at K = 50 and K = 100 (lists in global memory): bash tools/ab_k.sh <variant> ...
pick() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3), round(j['roofline']['frac'],4))"; }
for v in "$@"; do
  if [ "$v" = main ]; then unset TVBF_LIB; else export TVBF_LIB=build/libtvbf_$v.so; fi
  echo "== $v"
  for k in 50 100; do
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --k $k --no-cpu-baseline 2>/dev/null | pick "sym k$k" || exit 1
    timeout -k 10 200 python bench.py --steps 5 --warmup 2 --k $k --sweep plain --no-cpu-baseline 2>/dev/null | pick "plain k$k" || exit 1
  done
done
