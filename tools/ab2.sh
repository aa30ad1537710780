#!/bin/bash
# usage (GPU box): tools/ab2.sh "<groups list>" name...  -- bench.py per variant library and group count ("main" = in-tree)
gl=$1; shift
mkdir -p gpurun_out/ab
for v in "$@"; do
  lib=""; [ "$v" != main ] && lib=$PWD/metta_amd/variants/lib_$v.so
  for g in $gl; do
    METTA_AMD_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-obs-pass --no-single --groups $g > gpurun_out/ab/$v.$g.json 2> gpurun_out/ab/$v.$g.err || { tail -5 gpurun_out/ab/$v.$g.err; exit 1; }
    python3 -c "
import json; d=json.loads(open('gpurun_out/ab/$v.$g.json').read().strip().splitlines()[-1])
print('$v groups $g', 'ms/step %.5f' % d['ms_per_step'], 'kernel ms %.5f' % d['roofline']['per_launch']['kernel_ms_mean'], d['roofline']['kernel'][:32])"
  done
done
