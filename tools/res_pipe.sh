#!/bin/bash
# register / scratch usage of selected kernels of the main translation unit (compiler remarks; CPU only)
cd "$(dirname "$0")/.."
mkdir -p build/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Iidto_amd/csrc -S --cuda-device-only -Rpass-analysis=kernel-resource-usage -mllvm -amdgpu-mfma-vgpr-form=1 $MAIN_FLAGS idto_amd/csrc/idto_hip.hip -o build/isa/main.s 2> build/isa/main.txt
for k in ${@:-penta_pipe_kernelILi19E}; do
  grep -A12 "Function Name: _ZN8idto_dev[0-9]*$k" build/isa/main.txt | grep -E "Name|VGPRs:|AGPRs|Scratch|Spill|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ' '; echo
done
