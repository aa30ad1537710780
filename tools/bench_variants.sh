#!/bin/bash
# usage: tools/bench_variants.sh name...   -- kernel times of experimental builds (tools/variants.sh), "main" = the in-tree library
export TMPDIR=/tmp
for v in "$@"; do
  lib=""; [ "$v" != main ] && lib=$PWD/metta_amd/variants/lib_$v.so
  mkdir -p gpurun_out/var_$v
  METTA_AMD_LIB=$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $PWD/gpurun_out/var_$v -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/var_$v/bench.json 2> gpurun_out/var_$v/err.log || exit 1
  python3 - "$v" <<'PY'
import csv, glob, json, sys
v = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(f"gpurun_out/var_{v}/**/*kernel_stats.csv", recursive=True)[0])))
t = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) / 1e3 for r in rows if "mg_" in r["Name"] and "reset" not in r["Name"]}
d = json.load(open(f"gpurun_out/var_{v}/bench.json"))
print(v, {k: round(x, 1) for k, x in t.items()}, "ms/step", round(d["ms_per_step"], 4), "value", f'{d["value"]:.3e}')
PY
done
