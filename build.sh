#!/bin/bash
# Builds libidto_hip.so (gfx950 kernels + C-ABI; links RCCL for the multi-GPU slab exchange).
# hipcc cross-compiles without a GPU.
# -ffp-contract=off: host/device bit-exactness of the finite-difference path.
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs, not AGPRs (saves the v_accvgpr_read moves on
# the solver's products phase: -150 cycles per block row).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Iidto_amd/csrc"
mkdir -p build
# two translation units, compiled side by side: the finite-difference kernel and everything else
$HIPCC $FLAGS $FD_FLAGS -c idto_amd/csrc/fd_launch.hip -o build/fd_launch.o "$@" &
$HIPCC $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 $MAIN_FLAGS -c idto_amd/csrc/idto_hip.hip -o build/idto_hip.o "$@"
wait %1
$HIPCC --offload-arch=gfx950 -fPIC -shared build/fd_launch.o build/idto_hip.o -o idto_amd/libidto_hip.so -L/opt/rocm/lib -lrccl
# libidto_opt.so: the host-side TrajectoryOptimizer (C++) + its C-ABI, on top of libidto_hip.so
g++ -O3 -std=c++17 -fPIC -shared -Wall -Iinclude idto_amd/csrc/host/trajectory_optimizer.cc \
  idto_amd/csrc/host/mpc_controller.cc idto_amd/csrc/host/idto_opt_c.cc -o idto_amd/libidto_opt.so -Lidto_amd -lidto_hip -Wl,-rpath,'$ORIGIN'
