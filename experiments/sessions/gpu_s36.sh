#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s36; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( timeout 900 python bench.py ) > $O/bench.log 2>&1
tail -1 $O/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['row_pass_ms'], d['roofline']['column_pass_ms'], d['roofline']['traffic'])
oc=d['other_configs']
for k,v in oc.items():
    if isinstance(v, dict) and 'ms' in v: print(k, round(v['ms']*1e3,1), round(v.get('frac_of_hbm_peak', v.get('frac_of_f32_mfma_peak', 0)),3))
print('polyF', d['polychromatic']['variant_F_fft_focus']['per_wavelength_ms_per_gpu'])
"
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-poly ) > $O/bench20.log 2>&1
tail -1 $O/bench20.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps20', d['value'], d['ms_per_step'])"
