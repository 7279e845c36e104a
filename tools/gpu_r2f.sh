#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
run() { # name, counters...
  n=$1; shift
  rm -rf gpurun_out/pmc_$n
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$n -- python $R/tools/exp_gemm_trace.py ) > gpurun_out/rocprof_pmc_$n.log 2>&1
  python - "$n" <<'PY'
import csv, glob, sys, collections
n = sys.argv[1]
fs = glob.glob(f'gpurun_out/pmc_{n}/**/*counter_collection.csv', recursive=True)
if not fs:
    print(n, 'no output'); print(open(f'gpurun_out/rocprof_pmc_{n}.log').read()[-1500:]); sys.exit()
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if 'pm::' not in r['Kernel_Name']: continue
    name = r['Kernel_Name'].split('(')[0][9:40] + ' g' + r['Grid_Size']
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, c in agg.items():
    print(f'{name:50s}', ' '.join(f'{k}={sorted(v)[len(v)//2]:.4g}' for k, v in c.items()))
PY
}
run a SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
run b SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run c TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
run d TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
