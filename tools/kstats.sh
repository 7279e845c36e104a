#!/bin/bash
# rocprofv3 kernel stats of one command, the table cut to name / calls / average: tools/kstats.sh <out dir> <command ...>
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- "$@" ) > $O/rocprof.log 2>&1
f=$(ls $O/prof/*/*kernel_stats.csv | tail -1)
cp $f $O/kernel_stats.csv; rm -rf $O/prof
python - $O/kernel_stats.csv <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%9.1f us x%-5s %s' % (float(r['AverageNs']) / 1e3, r['Calls'], r['Name'][:150]))
P
