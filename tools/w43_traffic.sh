#!/bin/bash
# Launch time and L2 <-> memory traffic of the F(4x4,3x3) network kernel for one build of the library.
# usage: tools/w43_traffic.sh <lib file under alphazero_amd/> [shape]     (two counter passes + one timing run)
lib=$1; shape=${2:-15}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
out=gpurun_out/w43t_${lib%.so}
rm -rf "$out"; mkdir -p "$out"
B="python3 tools/net_forms.py --shapes $shape --forms 5 --reps 5 --lib $lib"
python3 tools/net_forms.py --shapes $shape --forms 5 --reps 30 --lib $lib 2>/dev/null | tail -1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 100 rocprofv3 --pmc $c -d "$out/$c" -o p --output-format csv -- $B > "$out/$c.log" 2>&1 || { echo "$c pass failed"; tail -3 "$out/$c.log"; }
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_net_w43" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
rd = sum(agg["FETCH_SIZE"]) / max(1, len(agg["FETCH_SIZE"])) * 1024 * 2 / 1e6   # KiB units, doubled on gfx950
wr = sum(agg["WRITE_SIZE"]) / max(1, len(agg["WRITE_SIZE"])) * 1024 / 1e6
print(f"  traffic per launch: read {rd:.1f} MB + written {wr:.1f} MB = {rd + wr:.1f} MB  ({len(agg['FETCH_SIZE'])} launches)")
PY
