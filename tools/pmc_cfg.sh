#!/bin/bash
# usage (GPU box): tools/pmc_cfg.sh <workload of tools/bench_configs.py> <tag> "<counters>"  -- one PMC pass, per-wave figures
w=$1; tag=$2; ctrs=$3
export TMPDIR=/tmp
R=$PWD
out=gpurun_out/pmc_$tag; mkdir -p $out
timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/$out -- python3 tools/bench_configs.py $w 20 > $out/bench.json 2> $out/err.log
echo "rc=$? ($w)"
python3 - <<PY
import csv,glob,collections
f=glob.glob('$out/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for row in csv.DictReader(open(fn)):
        agg[row['Kernel_Name'][:44]][row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in agg.items():
    if 'mg_step' not in k and 'mg_obs' not in k and 'mg_act' not in k and 'mg_uses' not in k: continue
    m={c: sum(x)/len(x) for c,x in v.items()}
    w=m.get('SQ_WAVES',0)
    print(k, {c: round(x) for c,x in m.items()}, 'n=',len(next(iter(v.values()))))
    if w: print('   per wave:', {c: round(x/w,1) for c,x in m.items() if c!='SQ_WAVES'})
PY
