#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
bash $R/tools/gpu_s15.sh > /dev/null 2>&1
O=$R/gpurun_out/s16; rm -rf $O; mkdir -p $O
( cd $R && PYTHONPATH=$R timeout 900 python tools/fuzz_fft2.py 600 77 2>&1 | tail -12 ) > $O/fuzz.log 2>&1
( cd $R && PYTHONPATH=$R timeout 600 python tools/exp_mix.py 2>&1 | grep -E "MIX|Error|error" ) > $O/exp_mix.log 2>&1
head -8 $R/gpurun_out/s15/mix_kernel_stats.csv | cut -c1-200; cat $R/gpurun_out/s15/pmc_sq.txt | head -60; cat $R/gpurun_out/s15/pmc_inst.txt | head -60; cat $O/fuzz.log
