#!/bin/bash
# usage: tools/ablate.sh "<diag values>"  -- obs-kernel time under MG_DIAG ablations (results are wrong on purpose)
export TMPDIR=/tmp
for d in $1; do
  mkdir -p gpurun_out/abl_$d
  MG_DIAG=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/abl_$d -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/abl_$d/bench.json 2> gpurun_out/abl_$d/err.log || exit 1
  echo "diag=$d $(grep "mg_"  $(find gpurun_out/abl_$d -name '*kernel_stats.csv' | head -1) | cut -d, -f1,4 | tr "\n" " ")"
done
