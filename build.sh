#!/bin/bash
# Builds libidto_hip.so (gfx950 kernels + C-ABI).  hipcc cross-compiles without a GPU.
# -ffp-contract=off: host/device bit-exactness of the finite-difference path.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
  -Iinclude -Iidto_amd/csrc idto_amd/csrc/idto_hip.hip -o idto_amd/libidto_hip.so "$@"
