#!/bin/bash
# round-3 session 5: full parity suite on the lean middle passes (windows, crop, 4096 / 8192-point tiles, Hermitian chain without the
# partner register array), timings of the chains they serve
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s5; rm -rf $O; mkdir -p $O
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $O/pytest_gpu.log 2>&1
( timeout 300 $R/tools/pm_gpu_check fused 2>&1 | grep -E "BENCH|FAIL|OK" ) > $O/fused.log 2>&1
for c in conv mtf config3 padded; do ( timeout 300 python bench.py --only $c | tail -1 | cut -c1-900 ) >> $O/only.log 2>&1; done
( timeout 300 python tools/exp_conv.py ) > $O/exp_conv.log 2>&1
( timeout 300 python tools/exp_bluestein.py ) > $O/exp_bluestein.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_conv -- python $R/bench.py --only conv ) > $O/rocprof_conv.log 2>&1
cp "$(ls $O/prof_conv/*/*kernel_stats.csv | tail -1)" $O/conv_kernel_stats.csv; rm -rf $O/prof_conv
tail -4 $O/pytest_gpu.log; cat $O/fused.log | cut -c1-160; cat $O/only.log | cut -c1-700; tail -12 $O/exp_conv.log; tail -12 $O/exp_bluestein.log; head -8 $O/conv_kernel_stats.csv | cut -c1-150
