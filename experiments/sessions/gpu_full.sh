#!/bin/bash
# full GPU suite + smoke + default bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/bench.log 2>&1
tail -4 gpurun_out/pytest_gpu.log; tail -1 gpurun_out/smoke.log; tail -1 gpurun_out/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'frac',d['whole_step_frac_of_hbm_peak'],'roofline',d['roofline']['frac'],d['roofline']['kernel'])
print('n2048',d['n2048']['value'],d['n2048']['whole_step_frac_of_hbm_peak'])
print('poly',{k:(v if not isinstance(v,dict) else {a:round(b,3) for a,b in v.items()}) for k,v in d['polychromatic'].items() if k.startswith('variant')})
for k,v in d['other_configs'].items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})
print('cpu',d['cpu_baseline']['value'], d['cpu_baseline'].get('tuned',{}).get('value'))
"
