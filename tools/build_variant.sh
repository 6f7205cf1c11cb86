#!/bin/bash
# Build a variant of the library for A/B runs: bash tools/build_variant.sh <name> [-DFLAG ...]
# writes build/libtvbf_<name>.so (same flags as native.build(); select it with TVBF_LIB=...).
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p build
SRC=tvbingefriend-recommendation-service_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude \
  -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -o build/libtvbf_$NAME.so \
  $SRC/tvbf_kernels.hip $SRC/tvbf_stats.hip $SRC/tvbf_assemble.hip $SRC/tvbf_rescore.hip 2>&1 | grep -v "packed-fp32-ops" || true
ls -la build/libtvbf_$NAME.so
