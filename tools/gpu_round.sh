#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace + PMC traffic.  Output -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py --steps 200 --warmup 20 ) > gpurun_out/bench.log 2>&1
B="python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-poly"
rm -rf gpurun_out/prof_bench gpurun_out/pmc_bench_fetch gpurun_out/pmc_bench_write
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- $B ) > gpurun_out/rocprof_bench.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_bench_fetch -- $B ) > gpurun_out/rocprof_pmc_fetch.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_bench_write -- $B ) > gpurun_out/rocprof_pmc_write.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_bench_fetch gpurun_out/pmc_bench_write gpurun_out/pmc_bench_summary.json > gpurun_out/pmc_bench_summary.txt 2>&1
for m in bench batch fused gemm; do ( timeout 300 $R/tools/pm_gpu_check $m 2>&1 | grep -E "BENCH|CHECK" ) > gpurun_out/check_$m.log 2>&1; done
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -2; cat gpurun_out/bench.log | tail -1 | cut -c1-600
cat gpurun_out/pmc_bench_summary.txt
