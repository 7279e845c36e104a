#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s9; rm -rf $O; mkdir -p $O
for r in 1 2; do for v in 0 1; do ( PM_TUNE=nt_out=$v timeout 200 python bench.py --only config3 | tail -1 | cut -c1-120 | sed "s/^/nt_out=$v /" ) >> $O/nt.log 2>&1; done; done
( timeout 200 python bench.py --only config3 | tail -1 | cut -c1-120 | sed "s/^/auto /" ) >> $O/nt.log 2>&1
( PM_TUNE=nt_out=1 timeout 300 $R/tools/pm_gpu_check fused 2>&1 | grep -E "BENCH|FAIL" | sed "s/^/nt_out=1 /" ) >> $O/nt.log 2>&1
( PM_TUNE=nt_out=0 timeout 300 $R/tools/pm_gpu_check fused 2>&1 | grep -E "BENCH|FAIL" | sed "s/^/nt_out=0 /" ) >> $O/nt.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu -k "angular or fused or focus_dft_intensity or conv" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
grep -v amdgpu.ids $O/nt.log | cut -c1-170; tail -3 $O/pytest.log
