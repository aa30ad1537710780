#!/bin/bash
# usage: tools/variants.sh name1="-DFLAG ..." name2="..."   -- build experimental variants of the library in parallel
# into metta_amd/variants/lib_<name>.so (they travel to the GPU box; select one with METTA_AMD_LIB=<path>).
cd "$(dirname "$0")/.."; mkdir -p metta_amd/variants
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-function $flags \
      metta_amd/csrc/mettagrid_hip.hip metta_amd/csrc/advantage.hip -o metta_amd/variants/lib_$name.so \
      > /tmp/variant_$name.log 2>&1; echo "variant $name rc=$?" ) &
done
wait
