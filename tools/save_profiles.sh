#!/bin/bash
# copy the judged summaries of the last tools/gpu_round.sh run from gpurun_out/ (scratch) into profiles/<round>/ (tracked)
r=${1:-r01}; mkdir -p profiles/$r
tail -1 gpurun_out/bench.log > profiles/$r/bench.json
cp "$(ls gpurun_out/prof_bench/*/*kernel_stats.csv | tail -1)" profiles/$r/bench_kernel_stats.csv
cp gpurun_out/pmc_bench_summary.json profiles/$r/pmc_bench_summary.json
cp gpurun_out/pmc_bench_summary.json profiles/pmc_bench_summary.json
cp gpurun_out/pmc_bench_summary.txt profiles/$r/pmc_bench_summary.txt
for m in bench batch fused gemm; do [ -s gpurun_out/check_$m.log ] && cp gpurun_out/check_$m.log profiles/$r/check_$m.log; done
cp gpurun_out/pytest_gpu.log profiles/$r/pytest_gpu.log
ls -la profiles/$r | head -40
