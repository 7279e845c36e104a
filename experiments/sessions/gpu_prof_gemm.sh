#!/bin/bash
# SQ counters of the complex GEMM on the config-4 shapes (rocprofv3 PMC passes, kernel trace separately)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_gemm; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="$R/tools/pm_gpu_check gemmprof"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq2 -- $B > $O/pmc_sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("$O/pmc_sq", "$O/pmc_sq2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        print(k)
        for n, v in sorted(c.items()):
            print(f'    {n:32s} {sum(v) / len(v):16.0f}  x{len(v)}')
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
