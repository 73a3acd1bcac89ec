#!/bin/bash
# Round-end evidence run on the GPU box (gpurun call A): full GPU suite, the default bench line, PMC traffic + SQ activity of the roofline
# kernels.  Outputs under gpurun_out/; tools/collect_profiles.sh copies what the docs cite into profiles/.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 --durations=15 > gpurun_out/final_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/final_pytest.log
tail -4 gpurun_out/final_pytest.log
bash profiles/collect_pmc.sh > gpurun_out/final_pmc.log 2>&1; tail -3 gpurun_out/final_pmc.log
# SQ activity of the two roofline kernels alone (MFMA busy, waits, LDS conflicts): one pass each, kernel trace only
for c in "fwd|python tools/gemm_one.py 8192 2048 512" "wgrad2|python tools/gemm_group_one.py"; do
  IFS='|' read -r N CMD <<< "$c"
  D=gpurun_out/pmc/sq_$N
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace -d $D -o g -- $CMD > $D.log 2>&1 \
    && python profiles/pmc_dump.py "$(find $D -name '*_results.db' | head -1)" p5_gemm5 > gpurun_out/pmc_sq_$N.txt 2>&1
done
cat gpurun_out/pmc_sq_fwd.txt gpurun_out/pmc_sq_wgrad2.txt 2>/dev/null | head -20
timeout 900 python bench.py > gpurun_out/final_bench.log 2>&1; grep '^{' gpurun_out/final_bench.log | tail -1 > gpurun_out/final_bench.json; tail -c 300 gpurun_out/final_bench.json
rm -rf gpurun_out/pmc/*/ 2>/dev/null
