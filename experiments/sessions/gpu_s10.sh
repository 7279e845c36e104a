#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s10; rm -rf $O; mkdir -p $O
( timeout 300 $R/tools/pm_gpu_check tune 4096 c64 3 "" "nt_out=1" "nt_in=1" "nt_in=1,nt_out=1" "nt_in=0,nt_out=0" 2>&1 | grep TUNE ) > $O/tune_nt.log 2>&1
( timeout 300 $R/tools/pm_gpu_check tune 2048 c64 3 "" "nt_out=1" "nt_in=1,nt_out=1" 2>&1 | grep TUNE ) >> $O/tune_nt.log 2>&1
( timeout 300 $R/tools/pm_gpu_check tune 4096 c128 3 "" "nt_out=0" "nt_out=1" "nt_in=0" 2>&1 | grep TUNE ) >> $O/tune_nt.log 2>&1
( timeout 300 $R/tools/pm_gpu_check tune 8192 c64 2 "" "nt_out=1" "nt_in=0" 2>&1 | grep TUNE ) >> $O/tune_nt.log 2>&1
for v in 0 1; do ( PM_TUNE=nt_out=$v timeout 200 python bench.py --only conv | tail -1 | cut -c1-160 | sed "s/^/nt_out=$v /" ) >> $O/conv_nt.log 2>&1; done
( timeout 200 python bench.py --only conv | tail -1 | cut -c1-160 | sed "s/^/auto /" ) >> $O/conv_nt.log 2>&1
cat $O/tune_nt.log | cut -c1-190; grep -v amdgpu.ids $O/conv_nt.log
