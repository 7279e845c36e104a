#!/bin/bash
# round-3 session 2: the new GEMM forms (parity + timing), intermediate layout x middle-pass form sweep of the fused chain
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s2; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "cgemm or mdft" 2>&1 | tail -15 ) > $O/pytest_gemm.log 2>&1
( timeout 900 python -m pytest tests -x -q -m gpu -k "mdft or gemm or config4 or config5 or executor or coronagraph or fpm" 2>&1 | tail -15 ) > $O/pytest_mdft.log 2>&1
( timeout 300 $R/tools/pm_gpu_check gemm ) > $O/gemm_check.log 2>&1
for wk in 1 0; do ( PM_TUNE=gemm_wk=$wk timeout 300 python bench.py --only config4 ) > $O/config4_wk$wk.log 2>&1; done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config4 -- python $R/bench.py --only config4 ) > $O/rocprof_config4.log 2>&1
cp "$(ls $O/prof_config4/*/*kernel_stats.csv | tail -1)" $O/config4_kernel_stats.csv; rm -rf $O/prof_config4
for lk in 0 1 2 3; do for m in 0 2; do
  ( PM_TUNE=log_k=$lk,colmul_mode=$m timeout 200 $R/tools/pm_gpu_check fused 2>&1 | grep -E "BENCH|FAIL" | sed "s/^/lk=$lk m=$m /" ) >> $O/fused_sweep.log 2>&1
done; done
tail -3 $O/pytest_gemm.log; tail -3 $O/pytest_mdft.log; grep -E "BENCH|FAIL|OK" $O/gemm_check.log | tail -30
for wk in 1 0; do echo wk=$wk; tail -1 $O/config4_wk$wk.log | cut -c1-400; done
head -8 $O/config4_kernel_stats.csv | cut -c1-200
grep "N=4096 in=4096\|FAIL" $O/fused_sweep.log | cut -c1-150
