"""kernel durations from a rocprofv3 kernel_trace.csv in launch order, consecutive launches of the same kernel grouped:
python tools/trace_runs.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'pm::' in r['Kernel_Name']]
pairs = {}
order = []
for r in rows:
    k = r['Kernel_Name'].split('(')[0][9:80] + ' g' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))
    if k not in pairs:
        pairs[k] = []
        order.append(k)
    pairs[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in order:
    v = pairs[k]
    step = 20 if len(v) % 20 == 0 else len(v)
    for i in range(0, len(v), step):
        c = sorted(v[i:i + step])
        print(f'{k:90s} n={len(c):3d} med {c[len(c) // 2]:7.2f} us')
