#!/bin/bash
# usage: tools/prof_r02.sh   (on the GPU box under gpurun) -- bench line, rocprofv3 kernel stats and the two HBM traffic
# PMC passes for the round-2 step path; results under gpurun_out/r02/
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r02; mkdir -p $O
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single > /dev/null 2> $O/pmc_fetch.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single > /dev/null 2> $O/pmc_write.err || exit 1
python3 - <<'PY'
import csv, glob, collections, json
def agg(d):
    out = collections.defaultdict(list)
    for fn in glob.glob(f'gpurun_out/r02/{d}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(fn)):
            out[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in out.items()}
print(json.dumps({f"{k[0]}|{k[1]}": v for k, v in {**agg('pmc_fetch'), **agg('pmc_write')}.items() if 'mg_' in k[0]}, indent=1))
PY
cat $O/bench.json
cut -d, -f1-4 $(find $O/stats -name "*kernel_stats.csv" | head -1) | head -5
