#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_check.sh <tag> "<pytest targets>" [bench args...]
# Runs the named GPU tests, then (only if they pass) one bench line; everything lands under gpurun_out/<tag>/.
tag=$1; targets=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
if [ -n "$targets" ]; then
  timeout -k 10 900 python3 -u -m pytest $targets -x -q -m gpu --timeout 180 --timeout-method thread > $out/tests.log 2>&1
  rc=$?; tail -4 $out/tests.log; echo "tests rc=$rc"
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
rc=$?; echo "bench rc=$rc"; tail -c 400 $out/bench.err
python3 - <<PY
import json
try:
    d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
    s=d.get('single_stream') or {}
    o=(d['roofline'].get('obs_kernel') or {})
    print('ms/step %.5f  value %.4g  frac %.4f | per-launch ms %.5f | single ms %s | obs ms %s' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['per_launch']['kernel_ms_mean'] or -1, s.get('ms_per_step'), o.get('ms_mean')))
except Exception as e:
    print('no bench line:', e)
PY
exit $rc
