#!/bin/bash
# usage: tools/prof.sh <tag> [extra bench args]   (runs on the GPU box under gpurun)
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" > gpurun_out/prof_$tag/bench.json 2> gpurun_out/prof_$tag/err.log
echo rc=$?
cut -c1-120 $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) | head -4
python3 -c "
import json;d=json.load(open('gpurun_out/prof_$tag/bench.json'));print('value',d['value'],'ms/step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms_mean'],'frac',d['roofline']['frac'])"
