"""Register / LDS / scratch table of every kernel in a built libprysm_amd.so (or a .o), read from the code object's metadata --
no GPU, no recompilation.  With two libraries: the kernels whose registers, spills or occupancy differ.

    python tools/kernel_table.py prysm_amd/libprysm_amd.so                       # the table
    python tools/kernel_table.py prysm_amd/alt_r04/libprysm_amd.so prysm_amd/libprysm_amd.so   # what changed

Occupancy here = waves per SIMD the register allocation allows (512 registers per lane per SIMD, granule 8), capped by the workgroup
size; a header edit that moves a 512-thread kernel from 128 to 130 registers halves the workgroups a CU holds, and nothing but this
table (or a slower run on the GPU) says so."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'


def code_objects(path):
    """gfx950 code objects in the .hip_fatbin section of a host shared library / object file -> list of temporary ELF paths.  The
    section is a run of uncompressed clang offload bundles (one per translation unit); each bundle's header lists (offset, size,
    target id) triples relative to the bundle's start."""
    import struct
    tmp = tempfile.mkdtemp(prefix='pmkt_')
    fat = os.path.join(tmp, 'fat.bin')
    subprocess.run(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', path, fat], check=True)
    d = open(fat, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    out, pos, k = [], 0, 0
    while True:
        pos = d.find(magic, pos)
        if pos < 0:
            break
        n = struct.unpack_from('<Q', d, pos + len(magic))[0]
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', d, q)
            target = d[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if 'gfx950' in target and size:
                o = os.path.join(tmp, 'co%d.elf' % k)
                k += 1
                open(o, 'wb').write(d[pos + off:pos + off + size])
                out.append(o)
        pos += len(magic)
    return out


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return r.stdout.split('\n')[:len(names)]


def kernels(path):
    """{demangled name: dict(vgpr, agpr, sgpr, spill, scratch, lds, wg)}"""
    rows = {}
    for co in code_objects(path):
        txt = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True).stdout
        cur = None
        recs = []
        for line in txt.splitlines():
            m = re.match(r'\s*-? *\.(\w+):\s*(.*)$', line)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            if k == 'agpr_count' or (k == 'args' and cur is None):
                pass
            if line.lstrip().startswith('- .agpr_count') or (line.lstrip().startswith('- .') and k in ('agpr_count', 'args')):
                cur = {}
                recs.append(cur)
            if cur is not None:
                cur[k] = v
        for r in recs:
            if 'name' not in r or 'vgpr_count' not in r:
                continue
            rows[r['name']] = dict(vgpr=int(r.get('vgpr_count', 0)), agpr=int(r.get('agpr_count', 0)), sgpr=int(r.get('sgpr_count', 0)),
                                   spill=int(r.get('vgpr_spill_count', 0)), scratch=int(r.get('private_segment_fixed_size', 0)),
                                   lds=int(r.get('group_segment_fixed_size', 0)), wg=int(r.get('max_flat_workgroup_size', 0)))
    names = list(rows)
    dn = demangle(names)
    return {d: rows[n] for n, d in zip(names, dn)}


def occupancy(k):
    """waves per SIMD the unified register file (512 per lane, allocation granule 8) leaves room for, at most 8 (on gfx90a and later the
    code object's .vgpr_count is the unified total: it already includes the accumulation registers)"""
    regs = k['vgpr']
    regs = max(8, (regs + 7) // 8 * 8)
    return min(8, 512 // regs)


def wgs_per_cu_by_regs(k):
    waves = max(1, (k['wg'] + 63) // 64)
    return (occupancy(k) * 4) // waves if waves <= occupancy(k) * 4 else 0


def short(name, n=150):
    name = re.sub(r'^void ', '', name)
    name = name.replace('pm::', '')
    return name if len(name) <= n else name[:n - 3] + '...'


def main(argv):
    if len(argv) == 2:
        ks = kernels(argv[1])
        for n in sorted(ks):
            k = ks[n]
            print('%4d vgpr %3d agpr %4d spill %5d scratch %2d occ %2d wg/CU(regs) %5d thr  %s' % (
                k['vgpr'], k['agpr'], k['spill'], k['scratch'], occupancy(k), wgs_per_cu_by_regs(k), k['wg'], short(n)))
        print('%d kernels' % len(ks))
        return 0
    a, b = kernels(argv[1]), kernels(argv[2])
    worse = 0
    for n in sorted(set(a) | set(b)):
        if n not in a:
            k = b[n]
            print('NEW      %4d vgpr %4d spill occ %d wg/CU %d  %s' % (k['vgpr'], k['spill'], occupancy(k), wgs_per_cu_by_regs(k), short(n)))
        elif n not in b:
            print('GONE     %s' % short(n))
        else:
            x, y = a[n], b[n]
            if (x['vgpr'], x['spill'], x['scratch']) != (y['vgpr'], y['spill'], y['scratch']):
                tag = 'same-occ'
                if wgs_per_cu_by_regs(y) < wgs_per_cu_by_regs(x) or y['spill'] > x['spill']:
                    tag = 'WORSE'
                    worse += 1
                elif wgs_per_cu_by_regs(y) > wgs_per_cu_by_regs(x) or y['spill'] < x['spill']:
                    tag = 'better'
                print('%-8s %4d -> %4d vgpr  %4d -> %4d spill  wg/CU %d -> %d  %s' % (
                    tag, x['vgpr'], y['vgpr'], x['spill'], y['spill'], wgs_per_cu_by_regs(x), wgs_per_cu_by_regs(y), short(n)))
    print('%d kernels before, %d after, %d worse' % (len(a), len(b), worse))
    return 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
