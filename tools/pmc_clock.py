"""effective shader clock and MFMA-pipe utilisation per kernel from a rocprofv3 --pmc run with --kernel-trace:
python tools/pmc_clock.py <dir> [<out.json>]   (counters: GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES)

mfma_busy of a kernel = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 x 1024):
the fraction of SIMD-cycles of the launch in which the MFMA pipe was busy.  With <out.json> the per-kernel figures are written with
the fingerprint of the kernel sources (bench.py prints `mfma_busy` in the config 4 entry only while the sources still hash to it)."""
import csv, glob, json, os, sys, collections
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
table = {}
for name, c in agg.items():
    med = {k: sorted(v)[len(v) // 2] for k, v in c.items()}
    t = sorted(dur[name])[len(dur[name]) // 2] if dur[name] else float('nan')
    line = f'{name:80s} dur {t:8.2f} us '
    if 'GRBM_GUI_ACTIVE' in med:
        line += f" GUI_ACTIVE {med['GRBM_GUI_ACTIVE']:.0f} -> {med['GRBM_GUI_ACTIVE'] / t / 1e3:.3f} GHz"
    for k in ('SQ_BUSY_CYCLES', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY'):
        if k in med:
            line += f' {k} {med[k]:.3g}'
    if med.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in med:
        busy = med['SQ_VALU_MFMA_BUSY_CYCLES'] / (med['GRBM_GUI_ACTIVE'] * 128.0)
        line += f' mfma_busy {busy:.3f}'
        table[name] = {'us_profiled': t, 'mfma_busy': busy, 'sclk_ghz': med['GRBM_GUI_ACTIVE'] / t / 1e3 / 8}
    print(line)
if len(sys.argv) > 2:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    gemm = {k: v for k, v in table.items() if 'cgemm' in k}
    tot = sum(v['us_profiled'] for v in gemm.values())
    out = {'_meta': {'source_fingerprint': bench.source_fingerprint(), 'command': 'bench.py --only config4'}, 'kernels': table,
           'mfma_busy_time_weighted_over_the_two_products': sum(v['mfma_busy'] * v['us_profiled'] for v in gemm.values()) / tot if tot else None}
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
