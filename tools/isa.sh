#!/bin/bash
# usage: tools/isa.sh [out_dir]  -- device assembly of the observation kernel only (the other kernels' bodies are
# stubbed by MG_ISA_LITE so that the compile takes seconds); prints register use and instruction counts.
out=${1:-/tmp/isa}; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMG_ISA_LITE $EXTRA -S --cuda-device-only \
  -o $out/lite.s "$(dirname "$0")/../metta_amd/csrc/mettagrid_hip.hip" 2>/dev/null
awk '/^_Z13mg_obs_kernelILb1ELb0EEv5KArgs:/{p=1} p{print} /s_endpgm/{if(p){exit}}' $out/lite.s > $out/obs_lite.s
grep -A12 "\.name: *_Z13mg_obs_kernelILb1ELb0" $out/lite.s | grep "sgpr_count\|sgpr_spill\|vgpr_count\|vgpr_spill"
echo "lines: $(grep -vc '^\s*;\|^\.\|^\s*$' $out/obs_lite.s)  valu: $(grep -c '^\sv_' $out/obs_lite.s)  salu: $(grep -c '^\ss_' $out/obs_lite.s)  lds: $(grep -c '^\sds_' $out/obs_lite.s)"
