#!/bin/bash
export TVBF_TUNING=1  # the library looks at its TVBF_* switches only in a process started with this set
# A/B timing of library variants built by tools/build_variant.sh, in one process environment:
#   bash tools/ab_variants.sh <variant> [<variant> ...]     ("main" = the shipped library)
# prints the step time and the pipe fraction of the plain sweep, the symmetric call and the
# 10k x 512 dense case for each.  Boxes differ by ~2 %: compare only numbers of one run.
pick() { python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3), round(j['roofline']['frac'],4))"; }
for v in "$@"; do
  if [ "$v" = main ]; then unset TVBF_LIB; else export TVBF_LIB=build/libtvbf_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --sweep plain --no-cpu-baseline 2>/dev/null | pick plain || exit 1
  if [ -z "$AB_PLAIN_ONLY" ]; then
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | pick sym || exit 1
  timeout -k 10 200 python bench.py --steps 50 --warmup 5 --shows 10000 --dims 512 --data dense --no-cpu-baseline 2>/dev/null | pick c2 || exit 1
  fi
done
