#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
rm -rf gpurun_out/prof_gemm
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_gemm -- python $R/tools/exp_gemm_trace.py ) > gpurun_out/rocprof_gemm.log 2>&1
python tools/trace_summary.py gpurun_out/prof_gemm > gpurun_out/gemm_timeline.txt 2>&1
cat gpurun_out/gemm_timeline.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/exp_mfma.cpp -o /tmp/exp_mfma && /tmp/exp_mfma | tee gpurun_out/exp_mfma.log
