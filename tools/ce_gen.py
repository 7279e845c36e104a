"""Writes prysm_amd/csrc/fft_ce_{rows,cols}_{f32,f64}.hip: the plans and workgroup shapes of the composite register engine that are built,
with the LDS row pads of tools/ce_banks.py.  Edit TABLE, run, rebuild.

    python tools/ce_gen.py            # the shipped shapes
    python tools/ce_gen.py exp        # ... plus the alternatives under #ifdef PM_CE_EXP (knobs ce_rows_seqs / ce_cols_seqs = 1000 COMP + SEQS)

A shape is (SEQS, COMP, WPE[, LOG_G]): sequences per workgroup, 1 = complex exchange / 2 = real and imaginary planes, waves per SIMD the
kernel is compiled for (0 = the default rule of fft_ce_kernels.h), and for column passes the 2^LOG_G adjacent tiles that run on one XCD.
The shipped shapes are the winners of tools/exp_ce_sweep.py (profiles/r05/exp_ce_sweep.log)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import functools  # noqa: E402

from ce_banks import pads as _pads  # noqa: E402


@functools.lru_cache(maxsize=None)
def pads(plan, seqs, col, es):
    return _pads(list(plan), seqs, col, es)

# length: plan, shipped (rows, cols) shape, alternatives for sweeps
TABLE = {
    'f32': {
        384: ((24, 4, 4), (8, 1, 0), (4, 1, 0, 4), [], []),
        768: ((24, 8, 4), (2, 1, 0), (4, 1, 0, 2), [], []),
        1152: ((24, 8, 6), (1, 1, 0), (8, 1, 0, -1), [], []),
        1280: ((20, 4, 4, 4), (1, 1, 0), (8, 1, 0, -1), [], []),
        1536: ((24, 8, 8), (1, 1, 0), (8, 1, 0, 3), [], []),
        2304: ((24, 12, 8), (2, 2, 0), (4, 1, 0, 3), [], []),
        2560: ((40, 8, 8), (1, 2, 0), (4, 1, 3, 4), [], []),
        3072: ((24, 8, 4, 4), (1, 2, 0), (4, 2, 0, 4), [], []),
        3600: ((30, 30, 2, 2), (1, 2, 0), (4, 2, 4, 3), [], []),
        5120: ((40, 8, 4, 4), (1, 2, 0), (4, 2, 3, 5), [], []),
        6144: ((24, 8, 8, 4), (1, 2, 0), (4, 2, 0, 5), [], []),
        500: ((10, 10, 5), (4, 1, 0), (8, 1, 0, 2), [], []),
        900: ((30, 30), (2, 1, 0), (4, 1, 0, 2), [], []),
        1000: ((10, 10, 10), (2, 1, 0), (4, 1, 0, 5), [], []),
        1500: ((30, 10, 5), (2, 2, 0), (8, 2, 4, 3), [], []),
        1600: ((20, 20, 4), (1, 1, 0), (8, 1, 0, 1), [], []),
        1800: ((30, 10, 6), (1, 1, 0), (4, 1, 4, 3), [], []),
        2000: ((20, 10, 10), (2, 1, 0), (4, 1, 0, 2), [], []),
        2500: ((10, 10, 5, 5), (1, 2, 0), (4, 2, 0, 3), [], []),
        3000: ((30, 10, 10), (1, 2, 0), (4, 2, 0, 5), [], []),
        4000: ((20, 20, 10), (1, 1, 0), (4, 2, 0, 3), [], []),
        4500: ((30, 15, 10), (1, 2, 0), (4, 2, 4, 5), [], []),
        5000: ((20, 10, 5, 5), (2, 2, 0), (4, 2, 0, 5), [], []),
        6000: ((30, 10, 10, 2), (1, 2, 0), (4, 2, 4, 5), [], []),
        8000: ((20, 20, 20), (1, 2, 0), (2, 1, 0, 4), [], []),
    },
    'f64': {
        384: ((24, 4, 4), (4, 1, 0), (4, 1, 0, -1), [], []),
        768: ((24, 8, 4), (1, 1, 0), (4, 1, 0, -1), [], []),
        1152: ((24, 8, 6), (1, 1, 0), (8, 2, 0, -1), [], []),
        1280: ((20, 4, 4, 4), (1, 1, 3), (4, 1, 3, 3), [], []),
        1536: ((24, 8, 8), (1, 2, 3), (4, 2, 3, 3), [], []),
        2304: ((24, 12, 8), (2, 2, 3), (2, 2, 3, 4), [], []),
        2560: ((40, 8, 8), (1, 2, 2), (4, 2, 2, 3), [], []),
        3072: ((24, 8, 4, 4), (1, 2, 3), (4, 2, 3, 5), [], []),
        3600: ((30, 30, 2, 2), (1, 2, 2), (4, 2, 2, 4), [], []),
        5120: ((40, 8, 4, 4), (1, 2, 2), (2, 2, 2, 6), [], []),
        6144: ((24, 8, 8, 4), (1, 2, 3), (2, 2, 3, 5), [], []),
        500: ((10, 10, 5), (2, 1, 0), (4, 1, 0, 4), [], []),
        900: ((30, 30), (2, 2, 2), (4, 2, 2, 2), [], []),
        1000: ((10, 10, 10), (1, 1, 0), (4, 1, 0, 1), [], []),
        1500: ((30, 10, 5), (2, 2, 2), (2, 2, 2, 5), [], []),
        1600: ((20, 20, 4), (1, 2, 3), (2, 2, 3, 4), [], []),
        1800: ((30, 10, 6), (1, 2, 2), (4, 2, 2, 1), [], []),
        2000: ((20, 10, 10), (1, 2, 3), (2, 2, 3, 5), [], []),
        2500: ((10, 10, 5, 5), (1, 1, 0), (2, 2, 0, 5), [], []),
        3000: ((30, 10, 10), (1, 2, 2), (4, 2, 2, 1), [], []),
        4000: ((20, 20, 10), (1, 2, 3), (4, 2, 3, 5), [], []),
        4500: ((30, 15, 10), (1, 2, 2), (2, 2, 2, 5), [], []),
        5000: ((20, 10, 5, 5), (1, 2, 3), (2, 2, 3, 2), [], []),
        6000: ((30, 10, 10, 2), (1, 2, 2), (2, 2, 2, 3), [], []),
        8000: ((20, 20, 20), (1, 2, 3), (2, 2, 3, 5), [], []),
    },
}
CT = {'f32': 'float', 'f64': 'double'}


def cfg(ct, plan, col, shape):
    seqs, comp, wpe = shape[:3]
    n = 1
    for r in plan:
        n *= r
    lds = n * seqs * (4 if ct == 'float' else 8) * (2 if comp == 1 else 1) * 1.06
    assert lds <= 150 * 1024, ('LDS of %s x %d (comp %d): %.0f KiB' % (plan, seqs, comp, lds / 1024))
    assert n // plan[0] * seqs <= 1024, ('threads', plan, seqs)
    assert seqs <= 16, 'fft_ce_kernels.h kCeMaxSeqs'
    es = (4 if ct == 'float' else 8) * (2 if comp == 1 else 1)
    pp = [p[0][1] for p in pads(tuple(plan), seqs, col, min(es, 8))] + [0, 0, 0]
    return 'CeCfg<%s, CePlan<%s>, %d, %s, %d, %d, %d, %d, %d>' % (ct, ', '.join(map(str, plan)), seqs, 'true' if col else 'false', comp, pp[0], pp[1], pp[2], wpe)


def emit(prec, col, exp, mid=False):
    ct = CT[prec]
    what = 'cols' if col else 'rows'
    L = []
    L.append('// GENERATED by tools/ce_gen.py -- composite register engine, complex%s %s passes: the plans and workgroup shapes that are built' % ('64' if prec == 'f32' else '128', 'column' if col else 'row'))
    L.append('// (fft_ce_kernels.h; LDS row pads: tools/ce_banks.py)')
    if prec == 'f32':
        L.append('#ifndef PM_CE_SCALAR')
        L.append('#define PM_PACKED_F32     // complex64 arithmetic on v_pk_*_f32 (pm_common.h)')
        L.append('#endif')
    L.append('#include "fft_ce_kernels.h"')
    L.append('')
    L.append('namespace pm {')
    L.append('')
    lens = sorted(TABLE[prec])
    if mid:
        L[0] = '// GENERATED by tools/ce_gen.py -- composite register engine, complex%s middle passes of fft2 -> x H -> ifft2 (the column shapes)' % ('64' if prec == 'f32' else '128')
    if mid:
        pass
    elif col:
        L.append('template <> bool ce_cols<%s>(const DirectIn<%s>& in, const ColStoreNat<%s>& out, hipStream_t st, int* rc) {' % (ct, ct, ct))
        L.append('    CeIn<%s> ci;' % ct)
        L.append('    CeColOut<%s> co;' % ct)
        L.append('    if (!ce_cols_view(in, out, ci, co)) return false;')
        knob, go, args = 'ce_cols_seqs', 'ce_cols_go', 'ci, co, tw, st, %d, in.nb'
    else:
        L.append('template <> bool ce_rows<%s>(const DirectIn<%s>& in, cx<%s>* out, int64_t out_ld, const RowStoreNat<%s>* o, hipStream_t st, int* rc, int64_t out_bstride) {' % (ct, ct, ct, ct))
        L.append('    CeIn<%s> ci;' % ct)
        L.append('    CeRowOut<%s> ro;' % ct)
        L.append('    CeSynth sy;')
        L.append('    if (!ce_rows_view(in, out, out_ld, out_bstride, o, ci, ro, sy)) return false;')
        knob, go, args = 'ce_rows_seqs', 'ce_rows_go', 'ci, ro, tw, st, sy, in.nb'
    if not mid:
        L.append('    const int n = in.ax.n;')
        L.append('    if (%s) return false;' % ' && '.join('n != %d' % n for n in lens))
    if not mid:
        L.append('    int err = 0;')
        L.append('    const cx<%s>* tw = twiddles<%s>(n, &err);' % (ct, ct))
        L.append('    if (!tw) {')
        L.append('        *rc = err;')
        L.append('        return true;')
        L.append('    }')
        if exp:
            L.append('#ifdef PM_CE_EXP')
            L.append('    switch (n * 10000 + tuning().%s) {     // alternatives: knob = 1000 COMP + SEQS' % knob)
            for n in lens:
                plan, rs, cs, ra, ca = TABLE[prec][n]
                seen = set()
                for sh in [cs if col else rs] + (ca if col else ra):
                    key = 1000 * sh[1] + sh[0]
                    if key in seen:
                        continue
                    seen.add(key)
                    L.append('        case %d: *rc = %s<%s>(%s); return true;' % (n * 10000 + key, go, cfg(ct, plan, col, sh), args % (sh[3] if len(sh) > 3 else -1) if col else args))
            L.append('    }')
            L.append('#endif')
        L.append('    switch (n) {')
        for n in lens:
            plan, rs, cs, ra, ca = TABLE[prec][n]
            L.append('        case %d: *rc = %s<%s>(%s); return true;' % (n, go, cfg(ct, plan, col, cs if col else rs), args % cs[3] if col else args))
        L.append('    }')
        L.append('    return false;')
        L.append('}')
    if mid:      # the middle pass of fft2 -> x H -> ifft2 on the column shapes
        L.append('template <> bool ce_cols_mul<%s>(const DirectIn<%s>& in, const MidMul<%s>& m, cx<%s>* dst, int64_t dst_pitch, hipStream_t st, int* rc) {' % (ct, ct, ct, ct))
        L.append('    CeIn<%s> ci;' % ct)
        L.append('    CeMul<%s> mm;' % ct)
        L.append('    if (!ce_mid_view(in, m, dst_pitch, ci, mm)) return false;')
        L.append('    const int n = in.ax.n;')
        L.append('    if (%s) return false;' % ' && '.join('n != %d' % n for n in lens))
        L.append('    int err = 0;')
        L.append('    const cx<%s>* tw = twiddles<%s>(n, &err);' % (ct, ct))
        L.append('    if (!tw) {')
        L.append('        *rc = err;')
        L.append('        return true;')
        L.append('    }')
        L.append('    switch (n) {')
        for n in lens:
            plan, rs, cs, ra, ca = TABLE[prec][n]
            L.append('        case %d: *rc = ce_cols_mul_go<%s>(ci, mm, dst, dst_pitch, tw, st, %d); return true;' % (n, cfg(ct, plan, True, cs), cs[3]))
        L.append('    }')
        L.append('    return false;')
        L.append('}')
    if not col and not mid:
        L.append('')
        L.append('template <> bool ce_has_plan<%s>(int n) { return %s; }' % (ct, ' || '.join('n == %d' % n for n in lens)))
    L.append('')
    L.append('}  // namespace pm')
    return '\n'.join(L) + '\n'


if __name__ == '__main__':
    exp = len(sys.argv) > 1 and sys.argv[1] == 'exp'
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'prysm_amd', 'csrc')
    for prec in TABLE:
        for col in (False, True):
            path = os.path.join(root, 'fft_ce_%s_%s.hip' % ('cols' if col else 'rows', prec))
            open(path, 'w').write(emit(prec, col, exp))
            print('wrote', os.path.relpath(path))
        path = os.path.join(root, 'fft_ce_mid_%s.hip' % prec)
        open(path, 'w').write(emit(prec, True, False, mid=True))
        print('wrote', os.path.relpath(path))
    # ... and the same shapes for the CPU emulator (tools/emu_ce.cpp)
    lines = ['// GENERATED by tools/ce_gen.py: every shipped shape of the composite register engine, for tools/emu_ce.cpp']
    for prec in sorted(TABLE):
        for n in sorted(TABLE[prec]):
            plan, rs, cs = TABLE[prec][n][:3]
            lines.append('CHECK(%s);' % cfg(CT[prec], plan, False, rs))
            lines.append('CHECK(%s);' % cfg(CT[prec], plan, True, cs))
    open(os.path.join(root, '..', '..', 'tools', 'emu_ce_plans.inc'), 'w').write('\n'.join(lines) + '\n')
