#!/bin/bash
export TMPDIR=/tmp
for v in "$@"; do
  L=build/variants/$v/libidto_hip.so; [ "$v" = base ] && L=idto_amd/libidto_hip.so
  echo "== $v"; IDTO_HIP_LIB=$L timeout 120 python tools/nd_timeline.py 2>&1 | grep "elimination of row\|^separator \|joiner \|producer "
done
