#!/bin/bash
# round-2 session B: full GPU suite, start-skew A/B at 2048^2 / 4096^2
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 $R/tools/pm_gpu_check tune 2048 c64 3 "" "row_skew=2" "row_skew=4" "row_skew=6" "col_skew=2" "col_skew=4" "col_skew=6" "row_skew=3,col_skew=3" "row_skew=4,col_skew=4" "row_skew=5,col_skew=5"
  timeout 300 $R/tools/pm_gpu_check tune 4096 c64 3 "" "row_skew=4" "row_skew=8" "col_skew=4" "col_skew=8" "row_skew=6,col_skew=6"
  timeout 300 $R/tools/pm_gpu_check tune 1024 c64 3 "" "row_skew=1,col_skew=1" "row_skew=2,col_skew=2" 
  timeout 300 $R/tools/pm_gpu_check tune 2048 c128 3 "" "row_skew=4,col_skew=4" "row_skew=6,col_skew=6" ) 2>&1 | grep TUNE > gpurun_out/tune_r2b.log
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/tune_r2b.log
