#!/bin/bash
# copy the judged summaries of the last tools/gpu_round.sh run from gpurun_out/round/ (scratch) into profiles/<round>/ (tracked)
r=${1:-r02}; O=gpurun_out/round; mkdir -p profiles/$r
tail -1 $O/bench.log > profiles/$r/bench.json
cp $O/pytest_gpu.log profiles/$r/pytest_gpu.log
cp $O/fuzz.log profiles/$r/fuzz.log 2>/dev/null; tail -1 $O/bench_steps20.log > profiles/$r/bench_steps20.json 2>/dev/null
cp $O/*_kernel_stats.csv $O/pmc_*_summary.json $O/pmc_*_summary.txt $O/config4_mfma_busy.txt $O/headline_sq_counters.txt $O/exp_*.log profiles/$r/ 2>/dev/null
cp $O/multi_rank/bench_2rank_gloo.json $O/multi_rank/check_2rank.json profiles/$r/ 2>/dev/null
cp $O/pmc_config4_mfma_busy.json profiles/pmc_config4_mfma_busy.json 2>/dev/null; rm -f profiles/$r/pmc_config4_mfma_busy.json
mv profiles/$r/pmc_bench_summary.json profiles/pmc_bench_summary.json     # ONE copy: bench.py reads roofline.traffic from it
ls -la profiles/$r | head -60
