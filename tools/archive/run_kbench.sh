#!/bin/bash
# builds kbench variants (here, on CPU) - run the binaries on the GPU box
set -e
cd "$(dirname "$0")"
mkdir -p build
build() { name=$1; shift; /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 "$@" -o build/kbench_$name kbench.hip; }
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  build $name $flags &
done
wait
ls -la build/
