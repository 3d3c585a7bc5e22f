#!/bin/bash
# ab_libs/lib_<name>.so = the library built with extra -D flags (A/B and timing-only experiments; tools/ab.sh runs them)
# usage: bash tools/build_variant.sh <name> [-DMACRO[=v] ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/ab_libs
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-strict-aliasing -Wall -Wno-unused-function \
  "$@" -x hip -shared -o $R/ab_libs/lib_$NAME.so $R/h263-rs_amd/csrc/kernels.hip $R/h263-rs_amd/csrc/batch.cpp $R/h263-rs_amd/csrc/batch_staging.cpp $R/h263-rs_amd/csrc/mixed_set.cpp $R/h263-rs_amd/csrc/state.cpp $R/h263-rs_amd/csrc/device_util.cpp $R/h263-rs_amd/csrc/worker_pool.cpp $R/h263-rs_amd/host/bitstream.cpp -lpthread
