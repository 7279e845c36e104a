"""Row strides of the composite register engine's LDS exchanges (prysm_amd/csrc/fft_ce.h): for a plan, a workgroup shape and an element
size, the pad (slots added to the row stride NT of the exchange INTO stage s) under which the gathers of that stage spread best over the
banks -- the template arguments PAD1.. of CeCfg.  Bank model: 4-byte accesses 64 lanes over 64 banks, 8-byte accesses 32 lanes over
32 bank pairs per pass; the figure is passes per wave instruction, 1.00 = conflict-free (the writes are consecutive lanes = always 1).

    python tools/ce_banks.py 30,10,10 5 row 4      # plan, sequences per workgroup, row|col, element bytes (4: fp32 planes, 8: complex64 / fp64 planes)"""
import sys
import numpy as np


def passes(R, seqs, col, es, SB, s):
    S, P, N = len(R), R[0], int(np.prod(R))
    TS = N // P
    M = [N // int(np.prod(R[:i + 1])) for i in range(S)]
    G = [TS // M[i] for i in range(S)]
    Q = [P // R[i] for i in range(S)]
    NT = TS * seqs
    lanes = nb = 64 if es == 4 else 32
    T = np.arange(NT)
    sl, t = (T % seqs, T // seqs) if col else (T // TS, T % TS)
    g, i2 = t // M[s], t % M[s]
    gw, r1 = g % G[s - 1], g // G[s - 1]
    tot = cnt = 0
    for u in range(Q[s]):
        for m in range(R[s]):
            tw = gw * M[s - 1] + i2 + M[s] * m
            a = (r1 + R[s] * u) * SB + (tw * seqs + sl if col else sl * TS + tw)
            for w0 in range(0, NT, 64):
                for h0 in range(w0, min(w0 + 64, NT), lanes):
                    tot += np.bincount(a[h0:h0 + lanes] % nb, minlength=nb).max()
                cnt += 1
    return tot / cnt / (64 // lanes)


def pads(R, seqs, col, es):
    N, P = int(np.prod(R)), R[0]
    NT = N // P * seqs
    out = []
    for s in range(1, len(R)):
        res = [(round(passes(R, seqs, col, es, NT + pad, s), 3), pad) for pad in range(0, 65)]
        out.append((min(res), res[0][0]))
    return out


if __name__ == '__main__':
    R = [int(v) for v in sys.argv[1].split(',')]
    seqs, col, es = int(sys.argv[2]), sys.argv[3] == 'col', int(sys.argv[4])
    for s, ((best, pad), nopad) in enumerate(pads(R, seqs, col, es), 1):
        print('exchange into stage %d: pad %2d -> %.2f passes (unpadded %.2f)' % (s, pad, best, nopad))
