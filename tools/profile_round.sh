#!/bin/bash
# Profiles a bench.py run on the GPU box: one kernel-trace pass and separate
# counter passes (never combined with a trace domain), then condenses them into
# profiles/<tag>_*.
# Usage: tools/profile_round.sh <tag> [<workload>]
#   no workload: the headline run (traffic figures go to the top level of profiles/traffic.json)
#   workload:    a name `bench.py --only` takes, e.g. hash_8192_15 (figures stored under that key)
# PASSES=traffic: kernel trace + the two HBM counter passes only
set -e
tag=${1:-r02}
wl=${2:-}
cd "$(dirname "$0")/.."
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ -n "$wl" ]; then
  B="python3 bench.py --only $wl"
  export TRAFFIC_KEY="$wl"
else
  B="python3 bench.py --no-cpu-baseline --no-secondary --steps 6 --warmup 2"
fi
timeout -k 10 170 rocprofv3 --kernel-trace --stats -d "$out/stats" -o s --output-format csv -- $B > "$out/stats.log" 2>&1
echo stats done
if [ "${PASSES:-all}" = "traffic" ]; then  # only the two HBM counters (a secondary workload's roofline.traffic)
  timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o p --output-format csv -- $B > "$out/pmc_fetch.log" 2>&1
  echo fetch done
  timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o p --output-format csv -- $B > "$out/pmc_write.log" 2>&1
  echo write done
  python3 tools/summarize_prof.py "$tag" "$out/stats" "$out/pmc_fetch" "$out/pmc_write"
  cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc.json profiles/traffic.json "$out/" 2>/dev/null || true
  exit 0
fi
timeout -s KILL 170 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$out/pmc_sq" -o p --output-format csv -- $B > "$out/pmc_sq.log" 2>&1
echo sq done
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d "$out/pmc_sq2" -o p --output-format csv -- $B > "$out/pmc_sq2.log" 2>&1
echo sq2 done
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d "$out/pmc_fetch" -o p --output-format csv -- $B > "$out/pmc_fetch.log" 2>&1
echo fetch done
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE -d "$out/pmc_write" -o p --output-format csv -- $B > "$out/pmc_write.log" 2>&1
echo write done
# matrix-core and LDS counters (the network kernel's bound)
timeout -s KILL 170 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE -d "$out/pmc_mfma" -o p --output-format csv -- $B > "$out/pmc_mfma.log" 2>&1
echo mfma done
python3 tools/summarize_prof.py "$tag" "$out/stats" "$out/pmc_sq" "$out/pmc_sq2" "$out/pmc_fetch" "$out/pmc_write" "$out/pmc_mfma"
cp profiles/${tag}_kernel_stats.csv profiles/${tag}_pmc.json profiles/traffic.json "$out/" 2>/dev/null || true
