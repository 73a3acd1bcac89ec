#!/bin/bash
# After `gpurun -- bash tools/final_run.sh`: copy the summaries the docs cite from gpurun_out/ (scratch) into profiles/ (tracked).
cd "$(dirname "$0")/.."
R=${1:-r03}
cp gpurun_out/final_train.md profiles/${R}_train_t5small_b64_kernel_stats.md
cp gpurun_out/final_train_by_shape.md profiles/${R}_train_t5small_b64_by_shape.md
cp gpurun_out/final_train_in_step.json profiles/in_step.json
cp gpurun_out/final_train_in_step.json profiles/${R}_train_in_step.json
cp gpurun_out/final_gen.md profiles/${R}_generate_t5small_b20_k10_kernel_stats.md
cp gpurun_out/final_gen_by_shape.md profiles/${R}_generate_t5small_b20_k10_by_shape.md
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/pmc_raw.txt profiles/${R}_pmc_gemm_raw.txt
cat gpurun_out/pmc_sq_fwd.txt gpurun_out/pmc_sq_wgrad2.txt > profiles/${R}_pmc_sq_roofline_kernels.txt 2>/dev/null || true
cp gpurun_out/final_bench.json profiles/${R}_bench_line.json
grep -E "passed|failed" gpurun_out/final_pytest.log | tail -1 > profiles/${R}_gpu_pytest_summary.txt
grep -A18 "slowest" gpurun_out/final_pytest.log > profiles/${R}_gpu_pytest_durations.txt 2>/dev/null || true
ls -la profiles/
