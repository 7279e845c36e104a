#!/bin/bash
# per-kernel durations of the fused angular-spectrum chain (rocprofv3 kernel trace of `pm_gpu_check fused`)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_fused; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $R/tools/pm_gpu_check fused > $O/trace.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:24]:
    print(r['Name'][:150], r['Calls'], r['AverageNs'], r['MinNs'])
PY
