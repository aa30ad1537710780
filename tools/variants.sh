#!/bin/bash
# usage: tools/variants.sh name1="-DFLAG ..." name2="..."   -- build experimental variants of the library
# into metta_amd/variants/lib_<name>.so (they travel to the GPU box; select one with METTA_AMD_LIB=<path>).
cd "$(dirname "$0")/.."; mkdir -p metta_amd/variants
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( make -s -j8 -C metta_amd/csrc OBJD=/tmp/variant_obj_$name OUT=$PWD/metta_amd/variants/lib_$name.so EXTRA="$flags" \
      > /tmp/variant_$name.log 2>&1; echo "variant $name rc=$?" )
done
