#!/bin/bash
# One rocprofv3 counter pass over a short bench run; prints the fused kernel's means.
# usage: tools/pmc_pass.sh <tag> COUNTER [COUNTER ...]   (at most 8 SQ counters per pass)
# PMC_KERNEL (substring of the kernel name, default k_simulate_fused) and PMC_SCRIPT
# (python script + arguments, default the short bench run) select another target.
tag=$1; shift
cd "$(dirname "$0")/.."
out=gpurun_out/pmc_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --pmc "$@" -d "$out" -o p --output-format csv -- python3 ${PMC_SCRIPT:-bench.py --no-cpu-baseline --steps 4 --warmup 1} > "$out/log.txt" 2>&1 || { tail -5 "$out/log.txt"; exit 1; }
python3 - "$out" "${PMC_KERNEL:-k_simulate_fused}" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    m = sum(v) / len(v)
    print(f"{c:32s} {m:16.1f}  per search {m / 819200:10.3f}  dispatches {len(v)}")
PY
