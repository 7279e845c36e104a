#!/bin/bash
# round-3 session 4: lean two-workgroups-per-CU middle pass (colmul_mode 3), fused FFTDFT axes, full parity suite, bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s4; rm -rf $O; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "middle_pass or fftdft" 2>&1 | tail -15 ) > $O/pytest_r3.log 2>&1
for m in 2 3 0; do ( PM_TUNE=colmul_mode=$m timeout 300 $R/tools/pm_gpu_check fused 2>&1 | grep -E "BENCH|FAIL" | sed "s/^/m=$m /" ) >> $O/fused_modes.log 2>&1; done
( cd /tmp && PM_TUNE=colmul_mode=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3 -- python $R/bench.py --only config3 ) > $O/rocprof_c3.log 2>&1
cp "$(ls $O/prof_c3/*/*kernel_stats.csv | tail -1)" $O/config3_m3_kernel_stats.csv; rm -rf $O/prof_c3
( timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $O/pytest_gpu.log 2>&1
( timeout 900 python bench.py ) > $O/bench.log 2>&1
tail -3 $O/pytest_r3.log; cat $O/fused_modes.log | cut -c1-150; head -5 $O/config3_m3_kernel_stats.csv | cut -c1-60,180-330; tail -3 $O/pytest_gpu.log; tail -1 $O/bench.log | cut -c1-300
