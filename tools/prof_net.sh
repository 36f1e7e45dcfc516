#!/bin/bash
# Counter passes over tools/net_forms.py for one board size and a list of kernel forms.
# Usage: tools/prof_net.sh <tag> <shape> <forms>      e.g. tools/prof_net.sh pair20 20 4,5
set -e
tag=$1; shape=$2; forms=$3
cd "$(dirname "$0")/.."
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 tools/net_forms.py --shapes $shape --forms $forms --reps 5"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d "$out/stats" -o s --output-format csv -- $B > "$out/stats.log" 2>&1
echo stats done
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d "$out/pmc_sq" -o p --output-format csv -- $B > "$out/pmc_sq.log" 2>&1
echo sq done
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -d "$out/pmc_sq2" -o p --output-format csv -- $B > "$out/pmc_sq2.log" 2>&1
echo sq2 done
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_IDX_ACTIVE -d "$out/pmc_mfma" -o p --output-format csv -- $B > "$out/pmc_mfma.log" 2>&1
echo mfma done
timeout -s KILL 120 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM -d "$out/pmc_if" -o p --output-format csv -- $B > "$out/pmc_if.log" 2>&1 || echo "ifetch pass failed"
echo ifetch done
python3 tools/summarize_prof.py "$tag" "$out/stats" "$out/pmc_sq" "$out/pmc_sq2" "$out/pmc_mfma" "$out/pmc_if"
