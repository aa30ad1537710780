#!/bin/bash
# usage (GPU box): tools/ab.sh name...  -- bench.py (4 groups, 300 steps) with each variant library; "main" = in-tree
for v in "$@"; do
  lib=""; [ "$v" != main ] && lib=$PWD/metta_amd/variants/lib_$v.so
  mkdir -p gpurun_out/ab
  METTA_AMD_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-obs-pass --no-single > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || { tail -5 gpurun_out/ab/$v.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/ab/$v.json').read().strip().splitlines()[-1])
print('$v', 'ms/step %.5f' % d['ms_per_step'], 'kernel ms %.5f' % d['roofline']['per_launch']['kernel_ms_mean'], d['roofline']['kernel'][:40])"
done
