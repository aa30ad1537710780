#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>"   -- one PMC pass over a short bench run
tag=$1; ctrs=$2
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/pmc_$tag
timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pmc_$tag/bench.json 2> gpurun_out/pmc_$tag/err.log
echo rc=$?
python3 - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_$tag/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for row in csv.DictReader(open(fn)):
        k=row['Kernel_Name'][:40]
        agg[k][row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in agg.items():
    if 'mg_' not in k: continue
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()}, 'n=',len(next(iter(v.values()))))
PY
