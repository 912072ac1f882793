#!/bin/bash
# Rebuilds only idto_hip.o (everything but fd_kernel's translation unit) and relinks: the solver kernels' edit loop.
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Iidto_amd/csrc"
$HIPCC $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 $MAIN_FLAGS -c idto_amd/csrc/idto_hip.hip -o build/idto_hip.o "$@"
$HIPCC --offload-arch=gfx950 -fPIC -shared build/fd_launch.o build/idto_hip.o -o idto_amd/libidto_hip.so -ldl
