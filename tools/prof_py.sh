#!/bin/bash
# rocprofv3 kernel stats of one python command: tools/prof_py.sh <out csv> <script> [args...]; prints the top kernels
export TMPDIR=/tmp
out=$1; shift
D=$(mktemp -d /tmp/prof.XXXX)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python "$@" ) > $D/log 2>&1
f=$(ls $D/*/*kernel_stats.csv 2>/dev/null | tail -1)
if [ -z "$f" ]; then tail -5 $D/log; exit 1; fi
cp $f $out
python - $out <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print(f"{float(r['AverageNs'])/1e3:9.2f} us x{r['Calls']:>5}  {r['Name'][:150]}")
PY
rm -rf $D
