#!/bin/bash
# interleaved A/B of two builds of the library on one box: A = prysm_amd/libprysm_amd.so, B = prysm_amd/alt/libprysm_amd.so
# usage: tools/ab_libs.sh <pm_gpu_check args...>     (rounds alternate A, B, A, B)
R=${GRAFT_REPO_ROOT:-$PWD}
for r in 1 2; do
  echo "== A (default build), round $r"; $R/tools/pm_gpu_check "$@" 2>&1 | grep -E "TUNE|BENCH"
  echo "== B (alt build), round $r"; LD_PRELOAD=$R/prysm_amd/alt/libprysm_amd.so $R/tools/pm_gpu_check "$@" 2>&1 | grep -E "TUNE|BENCH"
done
