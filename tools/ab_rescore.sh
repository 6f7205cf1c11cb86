#!/bin/bash
# Timing-only variants of the re-rank kernel side by side (tools/build_variant.sh rs1 -DTVBF_DIAG_RESCORE=1: the
# show's row alone; rs2 -DTVBF_DIAG_RESCORE=2: no gather of the candidates' rows) -- results are wrong by construction.
for v in main "$@"; do
  if [ "$v" = main ]; then unset TVBF_LIB; else export TVBF_LIB=build/libtvbf_$v.so; fi
  echo "== $v"; timeout -k 10 200 python -u tools/time_rerank.py 100000 1024 20 hybrid 2>&1 | grep -o "rescore launch [0-9.]* ms"
done
