#!/bin/bash
# interleaved A/B of tuning knobs: tools/gpu_ab.sh N c64|c128 rounds "cfg" "cfg" ...   (one line per configuration)
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
timeout 600 $R/tools/pm_gpu_check tune "$@" 2>&1 | grep TUNE | tee -a gpurun_out/tune_ab.log
