#!/bin/bash
# round-2 session C: new 128x128 GEMM (check + bench), remaining round-2 tests, kernel timeline of config 2
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
( timeout 600 $R/tools/pm_gpu_check gemm 2>&1 | grep -E "BENCH|CHECK|FAIL|gemm_" ) > gpurun_out/check_gemm.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu -k "two_rank or mdft or executors or coronagraph or fuzz or config5_variant_m" 2>&1 | tail -30 ) > gpurun_out/pytest_gpu.log 2>&1
rm -rf gpurun_out/prof_c2
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -- python $R/bench.py --only config2 ) > gpurun_out/rocprof_c2.log 2>&1
python - <<'PY' > gpurun_out/c2_timeline.txt 2>&1
import csv, glob
f = glob.glob('gpurun_out/prof_c2/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'fft_kernel' in r['Kernel_Name']]
rows = rows[-60:]
prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    kind = 'row' if 'RowLoad' in r['Kernel_Name'] else 'col'
    print(kind, 'dur %.2f us' % ((e - s) / 1e3), 'gap %.2f us' % ((s - prev) / 1e3) if prev else '')
    prev = e
PY
cat gpurun_out/check_gemm.log; tail -5 gpurun_out/pytest_gpu.log; tail -24 gpurun_out/c2_timeline.txt
