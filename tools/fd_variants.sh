#!/bin/bash
# Builds variants of libidto_hip.so that differ only in how csrc/fd_launch.hip was compiled (extra hipcc flags per
# variant) into build/variants/<name>/libidto_hip.so; tools/fd_stops.py picks one with IDTO_HIP_LIB.
# usage: fd_variants.sh name1 "flags1" name2 "flags2" ...   (build/idto_hip.o must exist: ./build.sh)
set -e
cd "$(dirname "$0")/.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Iidto_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  mkdir -p build/variants/$name
  ( $HIPCC $FLAGS $extra -c idto_amd/csrc/fd_launch.hip -o build/variants/$name/fd_launch.o &&
    $HIPCC --offload-arch=gfx950 -fPIC -shared build/variants/$name/fd_launch.o build/idto_hip.o -o build/variants/$name/libidto_hip.so -ldl &&
    echo built $name ) &
done
wait
