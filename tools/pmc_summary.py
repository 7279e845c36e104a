#!/usr/bin/env python
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into per-kernel HBM traffic per launch.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots), so they come from two runs of the same
command.  Units: the counters are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports
exactly 1/2 of the bytes of a wide coalesced streaming read, so reads are doubled; WRITE_SIZE is taken as is.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(n):
    m = re.search(r'cgemm_dma_kernel<(\d+), (\d+)', n)
    if m:
        return f'cgemm_dma_{64 * int(m.group(1))}x{64 * int(m.group(2))}_f32'
    if 'cgemm' in n:
        return 'cgemm_' + ('f32' if '<float' in n else 'f64')
    if 'splitk' in n:
        return 'splitk_reduce'
    m = re.search(r'FftCfg<(\w+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, \d+)?>', n)      # (+ the points-per-thread argument, round 4)
    if m:
        kind = 'row_pass' if 'RowLoad' in n else 'column_pass'
        if 'c2r' in n:
            kind = 'row_pass_c2r'
        elif 'col_mul_herm' in n:
            kind = 'column_pass_mul_herm'
        elif 'spectral' in n:
            kind += '_spectral'
        elif 'r2c' in n:
            kind = 'row_pass_r2c'
        elif 'row_hermt' in n:
            kind = 'row_pass_hermt'        # transposed Hermitian form, pass B (fft_hermt.h)
        elif 'col_hermt' in n:
            kind = 'column_pass_hermt'     # ... pass A (planes of N points when folded)
        elif 'herm' in n:
            kind = 'column_pass_herm'
        elif 'conv1' in n:
            kind = 'czt_axis_' + ('cols' if ', true>' in n else 'rows')
        elif 'col_mul' in n:
            kind = 'column_pass_mul'
        return f'fft_{kind}_{m.group(1)}_N{1 << int(m.group(2))}'
    return re.sub(r'\(.*', '', n)[:60]


def collect(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                agg[short(r['Kernel_Name'])].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch = collect(sys.argv[1], 'FETCH_SIZE')
    write = collect(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        rd = f * 1024 * 2
        wr = w * 1024
        out[k] = {'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w, 'launches': max(nf, nw),
                  'hbm_read_bytes_corrected': rd, 'hbm_write_bytes': wr, 'hbm_traffic_bytes': rd + wr}
    # what these counters describe: the kernel sources of THIS tree (bench.py prints `traffic` only while they still match)
    from bench import source_fingerprint
    meta = {'source_fingerprint': source_fingerprint(), 'command': sys.argv[4] if len(sys.argv) > 4 else 'bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-poly',
            'passes': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs; FETCH_SIZE doubled (gfx950)'}
    json.dump(dict(out, _meta=meta), open(sys.argv[3], 'w'), indent=1)
    for k, v in out.items():
        print(f"{k:48s} read {v['hbm_read_bytes_corrected'] / 1e6:9.1f} MB  write {v['hbm_write_bytes'] / 1e6:9.1f} MB  x{v['launches']}")


if __name__ == '__main__':
    main()
