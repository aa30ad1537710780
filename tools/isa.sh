#!/bin/bash
# usage: tools/isa.sh <kernel file stem, e.g. k_obs> [out_dir]  -- device assembly of one kernel translation unit;
# prints register use, spills, occupancy and instruction counts of each kernel in it.
stem=${1:-k_obs}; out=${2:-/tmp/isa}; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function $EXTRA -S --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -o $out/$stem.s "$(dirname "$0")/../metta_amd/csrc/$stem.hip" 2>&1 \
  | grep -E "Function Name|TotalSGPRs|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed 's/.*remark: [^ ]* *//; s/ \[-Rpass.*//' | paste - - - - - - - - - 
echo "valu: $(grep -c '^\sv_' $out/$stem.s)  salu: $(grep -c '^\ss_' $out/$stem.s)  lds: $(grep -c '^\sds_' $out/$stem.s)  scratch: $(grep -c 'scratch_' $out/$stem.s)"
