"""Instruction mix of the kernels in a device assembly listing (hipcc --cuda-device-only -S): VALU / LDS / VMEM / barriers per kernel
whose mangled name contains the filter.  Straight-line kernels (the register engines) run each counted instruction once per thread.
    python tools/asm_count.py file.s [filter]"""
import collections
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z[^\n:]*):[^\n]*\n(.*?)\n\s*s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    c = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in ';.' or line.endswith(':'):
            continue
        op = line.split()[0]
        if op.startswith('v_'):
            c['valu'] += 1
            if op.startswith('v_pk_'):
                c['valu_pk'] += 1
            if 'fma' in op or 'mul_f' in op or 'add_f' in op or 'sub_f' in op:
                c['valu_fp'] += 1
            if op.startswith('v_mov') or op.startswith('v_accvgpr'):
                c['valu_mov'] += 1
        elif op.startswith('ds_'):
            c['lds'] += 1
        elif op.startswith('global_') or op.startswith('buffer_') or op.startswith('scratch_'):
            c['vmem'] += 1
            if op.startswith('scratch_'):
                c['scratch'] += 1
        elif op.startswith('s_barrier'):
            c['barrier'] += 1
        elif op.startswith('s_waitcnt'):
            c['waitcnt'] += 1
        elif op.startswith('s_'):
            c['salu'] += 1
    short = subprocess = None
    print('%-150s %s' % (name[:150], dict(c)))
