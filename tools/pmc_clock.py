"""effective shader clock and MFMA-pipe utilisation per kernel from a rocprofv3 --pmc run with --kernel-trace:
python tools/pmc_clock.py <dir>   (counters: GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES)"""
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in csv.DictReader(open(cc)):
    name = r['Kernel_Name'].split('(')[0][:60] + ' g' + r.get('Grid_Size', '?')
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    key = r.get('Dispatch_Id')
    if key not in seen and 'Start_Timestamp' in r:
        seen.add(key)
        dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for name, c in agg.items():
    med = {k: sorted(v)[len(v) // 2] for k, v in c.items()}
    t = sorted(dur[name])[len(dur[name]) // 2] if dur[name] else float('nan')
    line = f'{name:80s} dur {t:8.2f} us '
    if 'GRBM_GUI_ACTIVE' in med:
        line += f" GUI_ACTIVE {med['GRBM_GUI_ACTIVE']:.0f} -> {med['GRBM_GUI_ACTIVE'] / t / 1e3:.3f} GHz"
    for k in ('SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY'):
        if k in med:
            line += f' {k} {med[k]:.3g}'
    print(line)
