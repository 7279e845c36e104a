#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s8; rm -rf $O; mkdir -p $O
for f in -1 0 1; do for c in config3 c128 conv; do ( PM_TUNE=fold=$f timeout 200 python bench.py --only $c | tail -1 | cut -c1-230 | sed "s/^/fold=$f /" ) >> $O/fold.log 2>&1; done; done
( PM_TUNE=fold=0 timeout 300 $R/tools/pm_gpu_check fused 2>&1 | grep -E "BENCH|FAIL" | sed "s/^/fold=0 /" ) >> $O/fold.log 2>&1
( cd /tmp && PM_TUNE=fold=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python $R/bench.py --only config3 ) > $O/rocprof_c3.log 2>&1
cp "$(ls $O/prof_c3/*/*kernel_stats.csv | tail -1)" $O/config3_fold0_kernel_stats.csv; rm -rf $O/prof_c3
grep -v amdgpu.ids $O/fold.log; head -5 $O/config3_fold0_kernel_stats.csv | cut -c1-70,200-330
