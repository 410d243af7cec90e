#!/bin/bash
# Build experimental variants of libnbody_hip.so into tools/exp/<name>/ (kernel tuning only).
set -e
cd "$(dirname "$0")/.."
build() { # name, extra flags
  name=$1; shift
  mkdir -p tools/exp/$name
  for f in kernels pipeline step_chain device_ctx rccl_bind shard_plan; do
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Inbody_amd/csrc "$@" -c nbody_amd/csrc/$f.hip -o tools/exp/$name/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o tools/exp/$name/libnbody_hip.so tools/exp/$name/*.o -ldl -lpthread
  rm tools/exp/$name/*.o
}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  build $name $flags
done
