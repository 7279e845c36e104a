#!/usr/bin/env python
"""Per-kernel averages of whatever counters a rocprofv3 --pmc run collected:  python tools/pmc_counters.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
flt = sys.argv[2] if len(sys.argv) > 2 else 'pm::'
for k, cs in sorted(agg.items()):
    if flt not in k:
        continue
    n = max(len(v) for v in cs.values())
    print(f'{k[:150]}  x{n}')
    for c, v in sorted(cs.items()):
        print(f'    {c:28s} {sum(v) / len(v):16.0f}')
