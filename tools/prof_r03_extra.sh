#!/bin/bash
# usage (GPU box): tools/prof_r03_extra.sh -- the one-handle form of the headline (kernel stats, HBM traffic passes), the
# other configurations' lines, and kernel stats of the navigation and combat workloads; results under gpurun_out/r03x/
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03x; mkdir -p $O
B="--groups 1 --no-cpu-baseline --no-obs-pass"
timeout -k 10 300 python3 bench.py $B > $O/single_bench.json 2> $O/err.log || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/single_stats -- python3 bench.py $B --steps 100 --warmup 10 > $O/single_bench_under_rocprof.json 2>> $O/err.log || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/single_pmc_$c -- python3 bench.py $B --steps 20 --warmup 5 > /dev/null 2>> $O/err.log || exit 1
done
: > $O/other.jsonl
timeout -k 10 120 python3 tools/bench_configs.py navigation 300 >> $O/other.jsonl 2>> $O/err.log || exit 1
MG_GROUPS=4 timeout -k 10 120 python3 tools/bench_configs.py navigation 300 >> $O/other.jsonl 2>> $O/err.log || exit 1
timeout -k 10 120 python3 tools/bench_configs.py combat 40 256 >> $O/other.jsonl 2>> $O/err.log || exit 1
MG_GROUPS=4 timeout -k 10 120 python3 tools/bench_configs.py combat 40 256 >> $O/other.jsonl 2>> $O/err.log || exit 1
timeout -k 10 120 python3 tools/bench_configs.py curriculum >> $O/other.jsonl 2>> $O/err.log || exit 1
timeout -k 10 120 python3 tools/bench_configs.py rollout 100 >> $O/other.jsonl 2>> $O/err.log || exit 1
timeout -k 10 120 python3 tools/bench_configs.py mapgen >> $O/other.jsonl 2>> $O/err.log || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nav_stats -- python3 tools/bench_configs.py navigation 100 > /dev/null 2>> $O/err.log || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/combat_stats -- python3 tools/bench_configs.py combat 40 256 > /dev/null 2>> $O/err.log || exit 1
for d in single_stats nav_stats combat_stats; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/$d.csv; done
python3 - <<'PY'
import csv, glob, collections
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    out = collections.defaultdict(list)
    for fn in glob.glob(f'gpurun_out/r03x/single_pmc_{c}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(fn)):
            out[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
    with open(f'gpurun_out/r03x/single_pmc_{c}.csv', 'w') as f:
        f.write('kernel,counter,mean_per_launch,launches\n')
        for (k, cn), v in sorted(out.items()):
            if 'mg_' in k:
                f.write(f'"{k}",{cn},{sum(v)/len(v):.1f},{len(v)}\n')
                print(k, cn, sum(v) / len(v), len(v))
PY
cat $O/other.jsonl | cut -c1-260; head -3 $O/single_stats.csv | cut -d, -f1-4; head -3 $O/nav_stats.csv | cut -d, -f1-4; head -4 $O/combat_stats.csv | cut -d, -f1-4
