#!/bin/bash
# usage (GPU box): tools/ab_env.sh "<bench args>" "NAME=VAL ..." ...  -- bench.py with each environment setting ("-" = none)
args=$1; shift
mkdir -p gpurun_out/abe
i=0
for v in "$@"; do
  i=$((i+1)); e=""; [ "$v" != "-" ] && e="$v"
  env $e timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-obs-pass --no-single $args > gpurun_out/abe/$i.json 2> gpurun_out/abe/$i.err || { tail -5 gpurun_out/abe/$i.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/abe/$i.json').read().strip().splitlines()[-1])
print('[$v]', 'ms/step %.5f' % d['ms_per_step'], 'kernel ms %.5f' % d['roofline']['per_launch']['kernel_ms_mean'], d['roofline']['kernel'][:44])"
done
