"""per-kernel duration / gap-before table from a rocprofv3 kernel_trace.csv: python tools/trace_summary.py <dir> [last_n]"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
agg = collections.defaultdict(list)
prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][:70] + ' g' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))
    agg[name].append(((e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
for k, v in agg.items():
    d = sorted(x[0] for x in v); g = sorted(x[1] for x in v)
    print(f'{k:100s} n={len(v):4d} dur med {d[len(d)//2]:8.2f} us  gap-before med {g[len(g)//2]:6.2f} us')
