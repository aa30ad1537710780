#!/bin/bash
# usage (GPU box): tools/ab3.sh "<variant> <groups> [ENV=VAL ...]" ...   -- one bench run per quoted spec
mkdir -p gpurun_out/ab
for spec in "$@"; do
  set -- $spec; v=$1; g=$2; shift 2
  lib=""; [ "$v" != main ] && lib=$PWD/metta_amd/variants/lib_$v.so
  env METTA_AMD_LIB=$lib "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-obs-pass --no-single --groups $g > gpurun_out/ab/x.json 2> gpurun_out/ab/x.err || { tail -5 gpurun_out/ab/x.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/ab/x.json').read().strip().splitlines()[-1])
print('$spec:', 'ms/step %.5f' % d['ms_per_step'], 'kernel ms %.5f' % d['roofline']['per_launch']['kernel_ms_mean'], d['roofline']['kernel'][:32])"
done
