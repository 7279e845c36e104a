#!/bin/bash
# packed fp32 arithmetic in the complex64 row / column kernels of the engine: parity first, then A (experiment build = unpacked) / B (product)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s47; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
true
true
( timeout 900 python tools/exp_ab_libs.py prysm_amd/alt/libprysm_amd.so prysm_amd/libprysm_amd.so 2 ) > $O/exp_ab_packed_all_but_rows.log 2>&1
cat $O/exp_ab_packed_all_but_rows.log
