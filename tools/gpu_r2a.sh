#!/bin/bash
# round-2 session A: full GPU suite at HEAD (new tests included), default bench line, A/B of the wide column tiles
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/bench.log 2>&1
( for dt in c64 c128; do timeout 300 $R/tools/pm_gpu_check tune 4096 $dt 3 "" "col_var=2" "col_var=2,log_k=3" "col_var=2,log_k=0"; done
  timeout 300 $R/tools/pm_gpu_check tune 2048 c64 3 "" "col_var=2" "fold=1"
  timeout 300 $R/tools/pm_gpu_check tune 8192 c64 2 "" ) 2>&1 | grep TUNE > gpurun_out/tune_r2a.log
( timeout 300 python tools/exp_host_overhead.py ) > gpurun_out/host_overhead.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/bench.log | cut -c1-3000; cat gpurun_out/tune_r2a.log; tail -12 gpurun_out/host_overhead.log
