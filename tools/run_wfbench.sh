#!/bin/bash
# build wfbench variants HERE (cross-compiled): run_wfbench.sh build <name> [kernel-header-dir] [-D flags...]
#   `base` is usually built once from a saved copy of thaler-study_amd/csrc (tools/build/base_csrc, made by `save-base`)
# on the GPU box: run_wfbench.sh run <logs...>   alternates every built variant three times
set -e
cd "$(dirname "$0")"
mkdir -p build
case "$1" in
  save-base) rm -rf build/base_csrc; cp -r ../thaler-study_amd/csrc build/base_csrc; rm -rf build/base_csrc/build; echo saved;;
  build) name=$2; dir=${3:-../thaler-study_amd/csrc}; shift; shift; shift || true
         /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -DKERNELS_HPP="\"$(realpath $dir)/kernels.hpp\"" "$@" -o build/wfbench_$name wfbench.hip
         ls -la build/wfbench_$name;;
  run) shift
       for rep in 1 2 3; do for b in build/wfbench_*; do $b $(basename $b | sed s/wfbench_//) "$@"; done; done;;
esac
