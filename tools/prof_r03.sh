#!/bin/bash
# usage: tools/prof_r03.sh [extra bench args]  (on the GPU box under gpurun) -- the bench line at both step counts,
# rocprofv3 kernel stats of the same command, the two HBM traffic PMC passes and the two SQ passes (instructions,
# cycles); results under gpurun_out/r03/  (copy the summaries into profiles/)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r03; mkdir -p $O
X="$@"
timeout -k 10 300 python3 bench.py $X > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $X > $O/bench_20.json 2>> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-single --no-obs-pass $X > $O/bench_under_rocprof.json 2> $O/stats.err || exit 1
pmc() {  # tag, counters
  timeout -s KILL 120 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/pmc_$1 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single --no-obs-pass $X > /dev/null 2> $O/pmc_$1.err || exit 1
}
pmc fetch "FETCH_SIZE"
pmc write "WRITE_SIZE"
pmc sq_insts "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH"
pmc sq_cycles "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"
python3 - <<'PY'
import csv, glob, collections, json
def agg(d):
    out = collections.defaultdict(list)
    for fn in glob.glob(f'gpurun_out/r03/{d}/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(fn)):
            out[(row['Kernel_Name'].split('(')[0], row['Counter_Name'])].append(float(row['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}
allc = {}
for d in ('pmc_fetch', 'pmc_write', 'pmc_sq_insts', 'pmc_sq_cycles'):
    rows = agg(d)
    with open(f'gpurun_out/r03/{d}_summary.csv', 'w') as f:
        f.write('kernel,counter,mean_per_launch,launches\n')
        for (k, c), (v, n) in sorted(rows.items()):
            if 'mg_' in k:
                f.write(f'"{k}",{c},{v:.1f},{n}\n')
                allc[f'{k}|{c}'] = v
print(json.dumps(allc, indent=1))
PY
cat $O/bench.json; cat $O/bench_20.json
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; cut -d, -f1-4 $f | head -6
