#!/bin/bash
# usage (GPU box): tools/short_run.sh -- how warm-up length and step count move the short-run figure of bench.py
mkdir -p gpurun_out/short
show() { python3 -c "
import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], 'ms/step %.5f' % d['ms_per_step'], 'region ms %.4f' % (d['ms_per_step']*d['steps']))" "$1" "$2"; }
for w in 5 50 300 1000; do
  timeout -k 10 200 python3 bench.py --steps 20 --warmup $w --no-cpu-baseline --no-single --no-obs-pass > gpurun_out/short/w$w.json 2>/dev/null || exit 1
  show gpurun_out/short/w$w.json "steps 20 warmup $w:"
done
for k in 40 80 160; do
  timeout -k 10 200 python3 bench.py --steps $k --warmup 5 --no-cpu-baseline --no-single --no-obs-pass > gpurun_out/short/k$k.json 2>/dev/null || exit 1
  show gpurun_out/short/k$k.json "steps $k warmup 5:"
done
