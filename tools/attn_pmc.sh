#!/bin/bash
# SQ instruction / stall counters of the attention kernels at a given shape (default the C5 shape), separate --pmc passes (kernel trace only):
#   gpurun -- bash tools/attn_pmc.sh [B H L]     -> gpurun_out/attn_pmc.txt
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1 TMPDIR=/tmp
mkdir -p gpurun_out
B=${1:-64}; H=${2:-16}; L=${3:-512}
OUT=gpurun_out/attn_pmc.txt; : > $OUT
rocprofv3 --list-avail 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u | tr '\n' ' ' > gpurun_out/sq_counters.txt
PMCG=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
        "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16"
        "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH")
i=0
for g in "${PMCG[@]}"; do
  D=gpurun_out/attn_pmc_$i; rm -rf $D
  rocprofv3 --pmc $g --kernel-trace -d $D -o a -- python tools/attn_bench.py $B $H $L > $D.log 2>&1 || { echo "group $i failed: $g" >> $OUT; tail -3 $D.log >> $OUT; }
  DB=$(find $D -name "*_results.db" | head -1)
  [ -n "$DB" ] && python profiles/pmc_dump.py "$DB" "p5_attn" >> $OUT
  rm -rf $D
  i=$((i+1))
done
cat $OUT
