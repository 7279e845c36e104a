#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s12; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests -x -q -m gpu -k "mtf or otf or real or conv or r2c or herm or psf or fuzz" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
for c in mtf conv; do ( timeout 200 python bench.py --only $c | tail -1 | cut -c1-200 ) >> $O/only.log 2>&1; done
( PYTHONPATH=$R timeout 300 python tools/exp_r2c.py 2>&1 | tail -14 ) > $O/exp_r2c.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mtf -- python $R/bench.py --only mtf ) > $O/rocprof_mtf.log 2>&1
cp "$(ls $O/prof_mtf/*/*kernel_stats.csv | tail -1)" $O/mtf_kernel_stats.csv; rm -rf $O/prof_mtf
tail -3 $O/pytest.log; grep -v amdgpu.ids $O/only.log; cat $O/exp_r2c.log | grep -v amdgpu.ids; head -4 $O/mtf_kernel_stats.csv | cut -c1-130
