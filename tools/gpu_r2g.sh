#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( timeout 600 $R/tools/pm_gpu_check gemm 2>&1 | grep -E "BENCH|FAIL|rc=|gemm_|GPU CHECK" ) > gpurun_out/check_gemm.log 2>&1
grep -E "FAIL|rc=|GPU CHECK" gpurun_out/check_gemm.log | head; grep -E "BENCH|gemm_tile" gpurun_out/check_gemm.log | head -30
rm -rf gpurun_out/pmc_gemm
( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_gemm -- python $R/tools/exp_gemm_trace.py ) > gpurun_out/rocprof_pmc_gemm.log 2>&1
python tools/pmc_clock.py gpurun_out/pmc_gemm 2>&1 | grep pm:: | tee gpurun_out/gemm_clock.txt
