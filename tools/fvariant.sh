#!/bin/bash
# usage: tools/fvariant.sh <stem> name1="-DFLAG ..." name2="..."   -- variants of ONE kernel translation unit (e.g. k_fused),
# linked with the other objects of the current build into metta_amd/variants/lib_<name>.so (METTA_AMD_LIB selects one).
cd "$(dirname "$0")/.."; mkdir -p metta_amd/variants
stem=$1; shift
others=$(ls metta_amd/csrc/build/*.o | grep -v "/$stem.o")
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $flags \
      -c metta_amd/csrc/$stem.hip -o /tmp/fv_$name.o > /tmp/fv_$name.log 2>&1 && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others /tmp/fv_$name.o -o metta_amd/variants/lib_$name.so >> /tmp/fv_$name.log 2>&1; echo "variant $name rc=$?" ) &
done
wait
