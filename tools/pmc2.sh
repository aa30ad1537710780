#!/bin/bash
# usage (GPU box): tools/pmc2.sh <variant|main> <tag> "<counters>" [MG_DIAG]  -- one PMC pass over a short bench run
# of one library build; prints the per-kernel averages and the per-wave figures.
v=$1; tag=$2; ctrs=$3; diag=${4:-0}
export TMPDIR=/tmp
R=$PWD
lib=""; [ "$v" != main ] && lib=$PWD/metta_amd/variants/lib_$v.so
out=gpurun_out/pmc_$tag; mkdir -p $out
METTA_AMD_LIB=$lib MG_DIAG=$diag timeout -s KILL 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/$out -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single --no-obs-pass ${PMC_BENCH_ARGS} > $out/bench.json 2> $out/err.log
echo "rc=$? ($v diag=$diag)"
python3 - <<PY
import csv,glob,collections
f=glob.glob('$out/**/*counter_collection.csv',recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for row in csv.DictReader(open(fn)):
        agg[row['Kernel_Name'][:44]][row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in agg.items():
    if 'mg_step' not in k and 'mg_obs' not in k and 'mg_act' not in k: continue
    m={c: sum(x)/len(x) for c,x in v.items()}
    w=m.get('SQ_WAVES',0)
    print(k, {c: round(x) for c,x in m.items()}, 'n=',len(next(iter(v.values()))))
    if w: print('   per wave:', {c: round(x/w,1) for c,x in m.items() if c!='SQ_WAVES'})
PY
