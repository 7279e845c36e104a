#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s6; rm -rf $O; mkdir -p $O
( PYTHONPATH=$R timeout 600 python tools/exp_variant_m.py ) > $O/exp_variant_m.log 2>&1
for w in 1 0; do ( PM_TUNE=herm_wide=$w timeout 200 python bench.py --only mtf | tail -1 | cut -c1-200 | sed "s/^/herm_wide=$w /" ) >> $O/mtf.log 2>&1; done
cat $O/exp_variant_m.log | grep -v amdgpu.ids; cat $O/mtf.log | grep -v amdgpu.ids
