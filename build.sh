#!/bin/bash
# Builds libidto_hip.so (gfx950 kernels + C-ABI; RCCL for the multi-GPU slab exchange is resolved at run time: dlopen).
# hipcc cross-compiles without a GPU.
# -ffp-contract=off: host/device bit-exactness of the finite-difference path.
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs, not AGPRs (saves the v_accvgpr_read moves on
# the solver's products phase: -150 cycles per block row).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Iidto_amd/csrc"
mkdir -p build
# two translation units, compiled side by side: the finite-difference kernel and everything else.
# -Rpass-analysis=kernel-resource-usage: the compiler's register / scratch / spill figures per kernel go to
# build/*.remarks, and tools/check_resources.py FAILS THE BUILD when a production kernel spills vector registers or
# exceeds its scratch / SGPR-spill limit (fd_kernel once computed wrong partials after an edit
# pushed its SGPR spills from 105 to 146 - that must not wait for a GPU to be noticed).
REM="-Rpass-analysis=kernel-resource-usage"
( $HIPCC $FLAGS $FD_FLAGS $REM -c idto_amd/csrc/fd_launch.hip -o build/fd_launch.o "$@" 2> build/fd_launch.remarks || { grep -v "remark:" build/fd_launch.remarks >&2; exit 1; } ) &
$HIPCC $FLAGS -mllvm -amdgpu-mfma-vgpr-form=1 $MAIN_FLAGS $REM -c idto_amd/csrc/idto_hip.hip -o build/idto_hip.o "$@" 2> build/idto_hip.remarks || { grep -v "remark:" build/idto_hip.remarks >&2; exit 1; }
wait %1
grep -h -A3 "warning:" build/fd_launch.remarks build/idto_hip.remarks >&2 || true
if [ -z "$IDTO_SKIP_RESOURCE_CHECK" ]; then python3 tools/check_resources.py build/fd_launch.remarks build/idto_hip.remarks > build/resource_check.txt || { cat build/resource_check.txt >&2; exit 1; }; fi
$HIPCC --offload-arch=gfx950 -fPIC -shared build/fd_launch.o build/idto_hip.o -o idto_amd/libidto_hip.so -ldl
# libidto_opt.so: the host-side TrajectoryOptimizer (C++) + its C-ABI, on top of libidto_hip.so
g++ -O3 -std=c++17 -fPIC -shared -Wall -Iinclude idto_amd/csrc/host/trajectory_optimizer.cc \
  idto_amd/csrc/host/mpc_controller.cc idto_amd/csrc/host/idto_opt_c.cc -o idto_amd/libidto_opt.so -Lidto_amd -lidto_hip -Wl,-rpath,'$ORIGIN'

# a C++ consumer of the boundary with no Python in the process (tests/test_gpu_cpp_consumer.py runs it on the GPU box)
g++ -O2 -std=c++17 -Wall -Iinclude tests/cpp/solve_acrobot.cc -o build/solve_acrobot -Lidto_amd -lidto_opt -lidto_hip -Wl,-rpath,'$ORIGIN/../idto_amd'
