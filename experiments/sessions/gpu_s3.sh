#!/bin/bash
# round-3 session 3: GEMM forms A/B, new parity tests, plain-transform layout sweeps, per-kernel times of config 3 per layout
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s3; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -15 ) > $O/pytest_r3.log 2>&1
for wk in 0 1 2 4 5 6; do ( PM_TUNE=gemm_wk=$wk timeout 300 python bench.py --only config4 | cut -c1-200 | sed "s/^/gemm_wk=$wk /" ) >> $O/config4_forms.log 2>&1; done
for wk in 0 5; do ( PM_TUNE=gemm_wk=$wk timeout 300 python bench.py --only adjoint | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_wk=$wk', d['adjoints']['mdft_adjoint_512_to_2048_c64'])" ) >> $O/config4_forms.log 2>&1; done
( cd /tmp && PM_TUNE=gemm_wk=5 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_config4 -- python $R/bench.py --only config4 ) > $O/rocprof_config4.log 2>&1
cp "$(ls $O/prof_config4/*/*kernel_stats.csv | tail -1)" $O/config4_kernel_stats.csv; rm -rf $O/prof_config4
for lk in 0 1 3; do
  ( cd /tmp && PM_TUNE=log_k=$lk timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_lk$lk -- python $R/bench.py --only config3 ) > $O/rocprof_c3_lk$lk.log 2>&1
  cp "$(ls $O/prof_c3_lk$lk/*/*kernel_stats.csv | tail -1)" $O/config3_lk${lk}_kernel_stats.csv; rm -rf $O/prof_c3_lk$lk
done
( timeout 300 $R/tools/pm_gpu_check tune 4096 c64 3 "" "log_k=0" "log_k=1" "log_k=2" "log_k=3" "log_k=5" 2>&1 | grep TUNE ) > $O/tune_logk.log 2>&1
( timeout 300 $R/tools/pm_gpu_check tune 4096 c128 3 "" "log_k=0" "log_k=1" "log_k=2" "log_k=3" 2>&1 | grep TUNE ) >> $O/tune_logk.log 2>&1
( timeout 300 $R/tools/pm_gpu_check tune 2048 c64 3 "" "log_k=0" "log_k=1" "log_k=2" "log_k=3" 2>&1 | grep TUNE ) >> $O/tune_logk.log 2>&1
( timeout 300 $R/tools/pm_gpu_check tune 8192 c64 2 "" "log_k=0" "log_k=1" "log_k=2" "log_k=3" "log_k=4" 2>&1 | grep TUNE ) >> $O/tune_logk.log 2>&1
tail -3 $O/pytest_r3.log; cat $O/config4_forms.log | cut -c1-260; head -6 $O/config4_kernel_stats.csv | cut -c1-140
for lk in 0 1 3; do echo lk=$lk; head -4 $O/config3_lk${lk}_kernel_stats.csv | cut -c1-60,180-330; done
cat $O/tune_logk.log | cut -c1-200
