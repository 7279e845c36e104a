#!/bin/bash
# round-3 session 1: parity suite, the three middle-pass forms of the fused chain (check + timing), the default bench line with the new
# config-5 keys, kernel stats of config 3 per form
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s1; rm -rf $O; mkdir -p $O
( timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $O/pytest_gpu.log 2>&1
for m in 0 1 2; do ( PM_TUNE=colmul_mode=$m timeout 300 $R/tools/pm_gpu_check fused ) > $O/fused_mode$m.log 2>&1; done
( timeout 900 python bench.py ) > $O/bench.log 2>&1
for m in 0 1 2; do
  ( cd /tmp && PM_TUNE=colmul_mode=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config3_m$m -- python $R/bench.py --only config3 ) > $O/rocprof_config3_m$m.log 2>&1
  cp "$(ls $O/prof_config3_m$m/*/*kernel_stats.csv | tail -1)" $O/config3_m${m}_kernel_stats.csv
  rm -rf $O/prof_config3_m$m
done
tail -3 $O/pytest_gpu.log; grep -h "BENCH\|FAIL" $O/fused_mode*.log; tail -1 $O/bench.log | cut -c1-600
for m in 0 1 2; do echo "mode $m"; head -5 $O/config3_m${m}_kernel_stats.csv | cut -c1-160; done
