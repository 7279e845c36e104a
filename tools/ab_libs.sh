#!/bin/bash
# interleaved A/B of two builds of the library on one box.  A = prysm_amd/libprysm_amd.so (tools/pm_gpu_check),
# B = prysm_amd/alt/libprysm_amd.so through a second check binary linked against it (tools/pm_gpu_check_alt: build with
# `make -C tools alt`).  LD_PRELOAD does NOT work for this: both copies register kernels under the same (interposed) host
# stubs and the library loaded last wins.
R=${GRAFT_REPO_ROOT:-$PWD}
for r in 1 2; do
  echo "== A (default build), round $r"; $R/tools/pm_gpu_check "$@" 2>&1 | grep -E "TUNE|BENCH|FAIL"
  echo "== B (alt build), round $r"; $R/tools/pm_gpu_check_alt "$@" 2>&1 | grep -E "TUNE|BENCH|FAIL"
done
