#!/bin/bash
# Builds variants of libidto_hip.so that differ only in how csrc/idto_hip.hip (everything but fd_kernel) was compiled -
# extra hipcc flags per variant, e.g. a measurement macro of penta_pipe.h - into build/variants/<name>/libidto_hip.so;
# `bash tools/gpu.sh variants <name> ...` benches them on the GPU box (IDTO_HIP_LIB picks one).
# usage: main_variants.sh name1 "flags1" name2 "flags2" ...   (build/fd_launch.o must exist: ./build.sh); variants build in parallel
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Iidto_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  mkdir -p build/variants/$name
  ( $HIPCC $FLAGS $extra -c idto_amd/csrc/idto_hip.hip -o build/variants/$name/idto_hip.o &&
    $HIPCC --offload-arch=gfx950 -fPIC -shared build/fd_launch.o build/variants/$name/idto_hip.o -o build/variants/$name/libidto_hip.so -ldl &&
    echo built $name ) &
done
wait
