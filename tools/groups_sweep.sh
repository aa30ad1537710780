#!/bin/bash
# usage (GPU box): tools/groups_sweep.sh "<groups list>" [env assignments...]  -- bench.py wall-clock ms/step per group count
mkdir -p gpurun_out/gs
for g in $1; do
  env "${@:2}" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-obs-pass --no-single --groups $g > gpurun_out/gs/g$g.json 2> gpurun_out/gs/g$g.err || { tail -3 gpurun_out/gs/g$g.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/gs/g$g.json').read().strip().splitlines()[-1])
print('groups $g ${@:2}', 'ms/step %.5f' % d['ms_per_step'], 'kernel ms %.5f' % d['roofline']['per_launch']['kernel_ms_mean'], d['roofline']['kernel'][:32])"
done
