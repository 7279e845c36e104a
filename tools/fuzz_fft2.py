"""Randomised differential test of pm_fft2 / pm_fft2_mul_ifft2 through prysm_amd._ops against numpy on the host:
random sizes (engine and direct-DFT lengths), windows, rotations, crops, real / complex / synthesised input, stacks,
epilogues, multipliers, precisions, with and without the fold; then pm_fft1 and pm_fft2_spectral the same way.  Usage: python tools/fuzz_fft2.py [ncases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from prysm_amd import _ops, _lib as L

def ref_fft2(x, M, N, in_off, in_shift, direction, scale):
    P = np.zeros((M, N), dtype=np.complex128)
    m, n = x.shape
    # logical element i reads mem[(i + shift) mod n - off]
    for ax, (size, ln, off, sh) in enumerate(((M, m, in_off[0], in_shift[0]), (N, n, in_off[1], in_shift[1]))):
        pass
    iy = (np.arange(M) + in_shift[0]) % M - in_off[0]
    ix = (np.arange(N) + in_shift[1]) % N - in_off[1]
    oky = (iy >= 0) & (iy < m)
    okx = (ix >= 0) & (ix < n)
    P[np.ix_(oky, okx)] = x[np.ix_(iy[oky], ix[okx])]
    F = np.fft.fft2(P) if direction < 0 else np.fft.ifft2(P) * (M * N)
    return F * scale

def window(F, out_shape, out_off, out_shift):
    M, N = F.shape
    om, on = out_shape
    out = np.zeros((om, on), dtype=F.dtype)
    ky = (np.arange(M) + out_shift[0]) % M - out_off[0]
    kx = (np.arange(N) + out_shift[1]) % N - out_off[1]
    oky = (ky >= 0) & (ky < om)
    okx = (kx >= 0) & (kx < on)
    out[np.ix_(ky[oky], kx[okx])] = F[np.ix_(np.nonzero(oky)[0], np.nonzero(okx)[0])]
    return out

def fuzz_fused(ncases, rng, lib):
    """window(ifft2(fft2(pad(x)) * H)) * scale: fused 3-pass chain (powers of two) or two-call composition."""
    # (round 4: composite column lengths -- 96, 100, 323 = 17 x 19, 360, 1000, 1020, 1536 -- take the three-pass chain with the mixed-radix
    # middle pass; the knobs mix_fused / mix_pad flip between it and the composition, padded and unpadded LDS slots)
    sizes = [4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 12, 20, 100, 96, 323, 360, 1000, 1020, 1536,
             500, 900, 1500, 1600, 1800, 2000, 2500, 3000, 768, 1152, 1280, 2304, 3072]      # round 5: lengths of the composite register engine (middle pass in registers)
    nfail, worst = 0, 0.0
    for case in range(ncases):
        big = rng.random() < 0.2
        M = int(rng.choice([2048, 4096] if big else sizes))
        N = int(rng.choice([2048, 4096] if big else sizes))
        lib.pm_set_tuning(b'mix_fused', int(rng.random() < 0.8))
        lib.pm_set_tuning(b'mix_engine', int(rng.random() < 0.75))
        lib.pm_set_tuning(b'mix_pad', int(rng.random() < 0.7))
        if M * N > (1 << 23):
            N = 2048
        cdt = np.complex64 if rng.random() < 0.5 else np.complex128
        m = M if rng.random() < 0.6 else int(rng.integers(1, M + 1))
        n = N if rng.random() < 0.6 else int(rng.integers(1, N + 1))
        in_off = (int(rng.integers(0, M - m + 1)), int(rng.integers(0, N - n + 1)))
        sh = lambda s_: int(rng.choice([0, s_ // 2]))
        in_shift = (sh(M), sh(N))
        out_shift = (sh(M), sh(N))
        om = M if rng.random() < 0.6 else int(rng.integers(1, M + 1))
        on = N if rng.random() < 0.6 else int(rng.integers(1, N + 1))
        out_off = (int(rng.integers(0, M - om + 1)), int(rng.integers(0, N - on + 1)))
        B = int(rng.choice([0, 0, 2, 3])) if M * N <= (1 << 18) else 0
        sep = rng.random() < 0.5
        per_field = bool(B) and rng.random() < 0.5
        conj = rng.random() < 0.4
        fold = int(rng.choice([-1, 0, 1]))
        lib.pm_set_tuning(b'fold', fold)
        shp = (B, m, n) if B else (m, n)
        x = (rng.standard_normal(shp) + 1j * rng.standard_normal(shp)).astype(cdt)
        nb = B if per_field else 1
        if sep:
            hy = (rng.standard_normal((nb, M)) + 1j * rng.standard_normal((nb, M))).astype(cdt)
            hx = (rng.standard_normal((nb, N)) + 1j * rng.standard_normal((nb, N))).astype(cdt)
            H = hy[:, :, None] * hx[:, None, :]
            mul = torch.from_numpy(hy if per_field else hy[0]).cuda()
            mul_x = torch.from_numpy(hx if per_field else hx[0]).cuda()
        else:
            H = (rng.standard_normal((nb, M, N)) + 1j * rng.standard_normal((nb, M, N))).astype(cdt)
            mul = torch.from_numpy(H if per_field else H[0]).cuda()
            mul_x = None
        scale = 1.0 / (M * N)
        try:
            got = _ops.fft2_mul_ifft2(torch.from_numpy(x).cuda(), scale=scale, mul=mul, mul_x=mul_x, mul_conj=conj, shape=(M, N),
                                      in_off=in_off, in_shift=in_shift, out_shape=(om, on), out_off=out_off,
                                      out_shift=out_shift).cpu().numpy()
        except Exception as exc:
            print('fused case', case, 'EXC', repr(exc)[:200], (M, N, m, n, cdt.__name__, B, sep, per_field, fold))
            nfail += 1
            continue
        fields = x if B else x[None]
        err = 0.0
        for b in range(fields.shape[0]):
            F = ref_fft2(fields[b].astype(np.complex128), M, N, in_off, in_shift, -1, 1.0)
            h = H[b if per_field else 0].astype(np.complex128)
            G = np.fft.ifft2(F * (np.conj(h) if conj else h)) * (M * N) * scale
            want = window(G, (om, on), out_off, out_shift)
            g = got[b] if B else got
            err = max(err, float(np.abs(g - want).max() / max(np.abs(want).max(), 1e-30)))
        tol = 1e-4 if cdt == np.complex64 else 1e-10
        worst = max(worst, err / tol)
        if not err < tol:
            nfail += 1
            print('fused case', case, 'FAIL err', err, (M, N, m, n, in_off, in_shift, (om, on), out_off, out_shift, cdt.__name__, B, sep, per_field, conj, fold))
    for key, val in ((b'fold', -1), (b'mix_fused', 1), (b'mix_pad', 1), (b'mix_engine', 1)):
        lib.pm_set_tuning(key, val)
    print(f'fuzz_fused: {ncases} cases, {nfail} failures, worst err/tol {worst:.3f}')
    return nfail


def fuzz_gemm(ncases, rng, lib):
    """alpha * op(A) @ op(B) for every op combination, ragged sizes (edge tiles, K tails, split-K and unsplit), strided views."""
    nfail, worst = 0, 0.0
    opf = {0: lambda z: z, 1: np.conj, 2: lambda z: z.T, 3: lambda z: z.conj().T}
    for case in range(ncases):
        big = rng.random() < 0.2
        dma = (not big) and rng.random() < 0.3     # the LDS-DMA kernel's shapes: multiples of 64 x 64 x 16, both tile sizes, split or not
        M, N, K = (int(rng.choice([64, 128, 512, 1000])), int(rng.choice([64, 300, 2048])), int(rng.choice([512, 2048, 3000]))) if big else \
                  ((64 * int(rng.integers(1, 6)), 64 * int(rng.integers(1, 6)), 16 * int(rng.integers(1, 40))) if dma else
                   (int(rng.integers(1, 200)), int(rng.integers(1, 200)), int(rng.integers(1, 300))))
        lib.pm_set_tuning(b'gemm_tile', int(rng.choice([0, 64, 128])))
        lib.pm_set_tuning(b'gemm_dma_wgs', int(rng.choice([1, 64, 512])))
        cdt = np.complex64 if rng.random() < 0.5 else np.complex128
        opA, opB = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        alpha = float(rng.choice([1.0, 0.37]))
        lib.pm_set_tuning(b'gemm_min_wgs', int(rng.choice([1, 1024])))
        lib.pm_set_tuning(b'gemm_3m', int(rng.choice([0, 1])))
        pad = int(rng.choice([0, 0, 3, 4]))    # leading dimension larger than the row (an odd one keeps complex64 off the DMA kernel)
        sa = (K, M) if opA & 2 else (M, K)
        sb = (N, K) if opB & 2 else (K, N)
        A = (rng.standard_normal((sa[0], sa[1] + pad)) + 1j * rng.standard_normal((sa[0], sa[1] + pad))).astype(cdt)
        B = (rng.standard_normal((sb[0], sb[1] + pad)) + 1j * rng.standard_normal((sb[0], sb[1] + pad))).astype(cdt)
        Ad = torch.from_numpy(A).cuda()[:, :sa[1]]
        Bd = torch.from_numpy(B).cuda()[:, :sb[1]]
        try:
            got = _ops.cgemm(Ad, Bd, opA, opB, alpha).cpu().numpy()
        except Exception as exc:
            print('gemm case', case, 'EXC', repr(exc)[:200], (M, N, K, opA, opB, cdt.__name__))
            nfail += 1
            continue
        want = alpha * (opf[opA](A[:, :sa[1]].astype(np.complex128)) @ opf[opB](B[:, :sb[1]].astype(np.complex128)))
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
        tol = 5e-5 if cdt == np.complex64 else 1e-12
        worst = max(worst, err / tol)
        if not err < tol:
            nfail += 1
            print('gemm case', case, 'FAIL err', err, (M, N, K, opA, opB, cdt.__name__, pad))
    lib.pm_set_tuning(b'gemm_min_wgs', 1024)
    lib.pm_set_tuning(b'gemm_3m', 1)
    lib.pm_set_tuning(b'gemm_tile', 0)
    lib.pm_set_tuning(b'gemm_dma_wgs', 512)
    print(f'fuzz_gemm: {ncases} cases, {nfail} failures, worst err/tol {worst:.3f}')
    return nfail


def fuzz_real_conv(ncases, rng, lib):
    """real_out=True of the same chain on a REAL object: real(ifft2(fft2(x) H)) -- the half-spectrum chain (forced for every legal
    shape, folded and not) or the real part of the complex chain, random rotations, general / conjugated multipliers, both precisions."""
    sizes = [2, 8, 32, 64, 128, 256, 1024, 4096, 12, 100]
    nfail, worst = 0, 0.0
    lib.pm_set_tuning(b'r2c', 2)
    for case in range(ncases):
        M, N = int(rng.choice(sizes)), int(rng.choice(sizes[2:]))
        if M * N > (1 << 22):
            M = 64
        rdt = np.float32 if rng.random() < 0.5 else np.float64
        cdt = np.complex64 if rdt == np.float32 else np.complex128
        shy = int(rng.choice([0, M // 2, int(rng.integers(0, M))]))
        shx = int(rng.choice([0, N // 2, N // 2, int(rng.integers(0, N))]))
        conj = rng.random() < 0.3
        fold = int(rng.choice([-1, 0, 1]))
        lib.pm_set_tuning(b'fold', fold)
        x = rng.standard_normal((M, N)).astype(rdt)
        H = (rng.standard_normal((M, N)) + 1j * rng.standard_normal((M, N))).astype(cdt)
        try:
            got = _ops.fft2_mul_ifft2(torch.from_numpy(x).cuda(), scale=1.0 / (M * N), mul=torch.from_numpy(H).cuda(), mul_conj=conj,
                                      in_shift=(shy, shx), out_shift=(shy, shx), real_out=True)
            assert not got.is_complex()
            got = got.cpu().numpy()
        except Exception as exc:
            print('real conv case', case, 'EXC', repr(exc)[:200], (M, N, rdt.__name__, shy, shx, conj, fold))
            nfail += 1
            continue
        xr = np.roll(x.astype(np.float64), (-shy, -shx), axis=(0, 1))
        h = H.astype(np.complex128)
        want = np.roll(np.fft.ifft2(np.fft.fft2(xr) * (np.conj(h) if conj else h)).real, (shy, shx), axis=(0, 1))
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
        tol = 1e-4 if rdt == np.float32 else 1e-10
        worst = max(worst, err / tol)
        if not err < tol:
            nfail += 1
            print('real conv case', case, 'FAIL err', err, (M, N, rdt.__name__, shy, shx, conj, fold))
    lib.pm_set_tuning(b'fold', -1)
    lib.pm_set_tuning(b'r2c', 1)
    print(f'fuzz_real_conv: {ncases} cases, {nfail} failures, worst err/tol {worst:.3f}')
    return nfail


def fuzz_fft1(ncases, rng, lib):
    """pm_fft1_ws through _ops.fft1: engine, direct, Bluestein, mixed-radix (3 / 5 / 7 x 2^k) and -- with the native length lowered --
    radix-2 / radix-4 lengths, both axes and directions, zero-padded inputs at an offset, windows of the bins, a scale."""
    sizes = [2, 8, 64, 256, 1024, 4096, 3, 12, 36, 96, 100, 127, 160, 192, 224, 384, 640, 1000, 1536, 2560, 3584, 6144, 45, 250, 1001, 2592, 3000, 5000, 6000, 8190,
             500, 900, 1500, 1800, 2500, 4000, 4500, 8000]
    nfail, worst = 0, 0.0
    for case in range(ncases):
        n = int(rng.choice(sizes))
        small = rng.random() < 0.3 and n in (64, 256)
        lib.pm_set_tuning(b'big_native_log', 5 if small else 13)      # 64 / 128 then take the radix-2 / radix-4 step; 256 goes direct
        lib.pm_set_tuning(b'mix', int(rng.choice([0, 2, 1, 1, 1, 1])))     # composite lengths: mostly their own kernel, sometimes round 2's routes
        lib.pm_set_tuning(b'mix_engine', int(rng.random() < 0.75))
        if small:
            n = int(rng.choice([64, 128]))
        cdt = np.complex64 if rng.random() < 0.5 else np.complex128
        batch = int(rng.choice([1, 2, 5, 8, 33]))
        ln = n if rng.random() < 0.5 else int(rng.integers(1, n + 1))
        off = int(rng.integers(0, n - ln + 1))
        axis = int(rng.integers(0, 2))
        direction = -1 if rng.random() < 0.5 else +1
        olen = n if rng.random() < 0.5 else int(rng.integers(1, n + 1))
        ooff = int(rng.integers(0, n - olen + 1))
        scale = float(rng.choice([1.0, 1.0 / n]))
        x = (rng.standard_normal((batch, ln)) + 1j * rng.standard_normal((batch, ln))).astype(cdt)
        padded = np.zeros((batch, n), dtype=np.complex128)
        padded[:, off:off + ln] = x
        want = (np.fft.fft(padded, axis=1) if direction < 0 else np.fft.ifft(padded, axis=1) * n)[:, ooff:ooff + olen] * scale
        xa = x if axis == 1 else np.ascontiguousarray(x.T)
        try:
            got = _ops.fft1(torch.from_numpy(xa).cuda(), n, axis=axis, direction=direction, in_off=off, out_off=ooff, out_len=olen,
                            scale=scale).cpu().numpy()
        except Exception as exc:
            print('fft1 case', case, 'EXC', repr(exc)[:200], (n, batch, ln, off, axis, direction, cdt.__name__))
            nfail += 1
            continue
        got = got if axis == 1 else got.T
        err = float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))
        tol = 3e-5 if cdt == np.complex64 else 1e-10
        worst = max(worst, err / tol)
        if not err < tol:
            nfail += 1
            print('fft1 case', case, 'FAIL err', err, (n, batch, ln, off, olen, ooff, axis, direction, cdt.__name__, small))
    lib.pm_set_tuning(b'big_native_log', 13)
    lib.pm_set_tuning(b'mix', 1)
    lib.pm_set_tuning(b'mix_engine', 1)
    print(f'fuzz_fft1: {ncases} cases, {nfail} failures, worst err/tol {worst:.3f}')
    return nfail


def fuzz_spectral(ncases, rng, lib):
    """pm_fft2_spectral (the wavelength loop as launch pairs over groups of wavelengths) against the fp64 sum of its terms: random
    shapes, pads, wavelength counts, group sizes and kernel forms, onto a non-zero accumulator."""
    from prysm_amd.propagation import focus_intensity
    sizes = [32, 64, 128, 256, 512, 1024]
    nfail, worst = 0, 0.0
    for case in range(ncases):
        m, n = int(rng.choice(sizes)), int(rng.choice(sizes))
        Q = int(rng.choice([1, 1, 2]))
        if m <= 128 and Q == 1 and rng.random() < 0.4:
            n = 2048      # rows of 2048 samples: with fold = 1 the folded grouped kernels
        count = int(rng.integers(1, 12))
        group, mode = int(rng.choice([1, 2, 3, 5, 8])), int(rng.integers(0, 4))
        lib.pm_set_tuning(b'spectral', group)
        lib.pm_set_tuning(b'spectral_mode', mode)
        lib.pm_set_tuning(b'fold', int(rng.choice([-1, 0, 1])))
        rdt = np.float32 if rng.random() < 0.6 else np.float64      # float64 maps: the complex128 grouped kernels
        amp = (rng.random((m, n)) > 0.3).astype(rdt)
        opd = (rng.standard_normal((m, n)) * 200).astype(rdt)
        ks = [2 * np.pi / w / 1e3 for w in rng.uniform(0.4, 0.9, count)]
        wts = list(rng.uniform(0.2, 2.0, count))
        base = rng.random((m * Q, n * Q)).astype(rdt)
        P = np.zeros((m * Q, n * Q), dtype=np.complex128)
        oy, ox = (m * Q - m + 1) // 2, (n * Q - n + 1) // 2
        want = base.astype(np.float64)
        for k, w in zip(ks, wts):
            P[:] = 0
            P[oy:oy + m, ox:ox + n] = amp.astype(np.float64) * np.exp(1j * k * opd.astype(np.float64))
            F = np.fft.fftshift(np.fft.fft2(np.fft.ifftshift(P), norm='ortho'))
            want = want + w * (F.real ** 2 + F.imag ** 2)
        try:
            acc = torch.from_numpy(base.copy()).cuda()
            focus_intensity(_ops.pack_amp_opd(torch.from_numpy(amp).cuda(), torch.from_numpy(opd).cuda()), Q, out=acc,
                            synth=('packed', ks[0]), spectral=(ks, wts))
            got = acc.cpu().numpy()
        except Exception as exc:
            print('spectral case', case, 'EXC', repr(exc)[:200], (m, n, Q, count, group, mode))
            nfail += 1
            continue
        err = float(np.abs(got - want).max() / np.abs(want).max())
        tol = 2e-5 if rdt == np.float32 else 1e-10
        worst = max(worst, err / tol)
        if not err < tol:
            nfail += 1
            print('spectral case', case, 'FAIL err', err, (m, n, Q, count, group, mode, rdt.__name__))
    for key, val in ((b'spectral', 8), (b'spectral_mode', 3), (b'fold', -1)):
        lib.pm_set_tuning(key, val)
    print(f'fuzz_spectral: {ncases} cases, {nfail} failures, worst err/tol {worst:.3f}')
    return nfail


def fuzz_real_any(ncases, rng, lib):
    """_ops.fft2_real (round 5): a real array of any height and EVEN width read as complex pairs, a half-size pm_fft2 on whatever route
    its lengths take, pm_r2c_untangle with arbitrary input / output rotations, the DC normalisation and the four epilogues."""
    heights = [1, 2, 3, 8, 30, 64, 100, 127, 250, 323, 512, 1000, 1001, 1536, 2048]
    widths = [2, 4, 6, 30, 64, 100, 126, 250, 360, 512, 1000, 1020, 1538, 1994, 2048, 4096]
    nfail, worst = 0, 0.0
    for case in range(ncases):
        M, N = int(rng.choice(heights)), int(rng.choice(widths))
        rdt = np.float32 if rng.random() < 0.5 else np.float64
        x = (rng.random((M, N)) + 0.1).astype(rdt)
        sh = lambda n: int(rng.choice([0, n // 2, int(rng.integers(0, n))]))
        ins, outs = (sh(M), sh(N)), (sh(M), sh(N))
        epi = int(rng.choice([L.PM_EPI_NONE, L.PM_EPI_NONE, L.PM_EPI_ABS, L.PM_EPI_ABS2, L.PM_EPI_ARG]))
        norm = bool(rng.random() < 0.5)
        scale = float(rng.choice([1.0, 0.5, 1.0 / np.sqrt(M * N)]))
        xt = torch.from_numpy(x).cuda()
        if rng.random() < 0.3:          # a view with a row pitch wider than the row
            wide = torch.zeros((M, N + 4), dtype=xt.dtype, device='cuda')
            wide[:, 2:N + 2] = xt
            xt = wide[:, 2:N + 2]
        got = _ops.fft2_real(xt, scale=scale, in_shift=ins, out_shift=outs, epilogue=epi, norm_dc=norm).cpu().numpy()
        F = np.fft.fft2(np.roll(x.astype(np.float64), (-ins[0], -ins[1]), axis=(0, 1)))
        if norm:
            F = F / F[0, 0]
        F = np.roll(F * scale, outs, axis=(0, 1))
        tol = 1e-10 if rdt == np.float64 else 3e-5
        if epi == L.PM_EPI_ARG:
            strong = np.abs(F) > 1e-3 * np.abs(F).max()
            err = np.max(np.abs(np.angle(np.exp(1j * (got - np.angle(F))))[strong])) / (1e-7 if rdt == np.float64 else 3e-3)
        else:
            ref = {L.PM_EPI_NONE: F, L.PM_EPI_ABS: np.abs(F), L.PM_EPI_ABS2: np.abs(F) ** 2}[epi]
            err = np.max(np.abs(got - ref)) / (np.max(np.abs(ref)) * tol * (2 if epi == L.PM_EPI_ABS2 else 1))
        worst = max(worst, err)
        if err > 1:
            nfail += 1
            print('FAIL real_any', M, N, rdt.__name__, ins, outs, epi, norm, scale, err)
    print(f'fuzz_real_any: {ncases} cases, {nfail} failures, worst err/tol {worst:.3f}')
    return nfail


def fuzz_hermitian(ncases, rng, lib):
    """pm_fft2 of a REAL power-of-two field on the Hermitian path, both forms (round 6: knob herm_t -1 / 0 / 1, its fold -1 / 0 / 1):
    rotations of 0 or half a length (sometimes arbitrary: the planner must then leave the transposed form), the DC normalisation, the four
    epilogues, scales, row pitches wider than the row, both precisions -- against numpy."""
    lens = [32, 64, 128, 256, 512, 1024, 2048, 4096, 8192]
    nfail, worst = 0, 0.0
    lib.pm_set_tuning(b'r2c', 2)
    for case in range(ncases):
        M, N = int(rng.choice(lens)), int(rng.choice(lens[:-1]))
        while M * N > (1 << 23):
            M, N = (M // 2, N) if rng.random() < 0.5 else (M, N // 2)
        rdt = np.float32 if rng.random() < 0.6 else np.float64
        lib.pm_set_tuning(b'herm_t', int(rng.choice([-1, 0, 1, 1])))
        lib.pm_set_tuning(b'herm_t_fold', int(rng.choice([-1, 0, 1])))
        sh = lambda n: int(rng.choice([0, n // 2, n // 2, int(rng.integers(0, n))]))
        ins, outs = (sh(M), int(rng.choice([0, N // 2]))), (sh(M), sh(N))
        epi = int(rng.choice([L.PM_EPI_NONE, L.PM_EPI_ABS, L.PM_EPI_ABS2, L.PM_EPI_ARG]))
        norm = bool(rng.random() < 0.5)
        scale = float(rng.choice([1.0, 0.5, 1.0 / np.sqrt(M * N)]))
        x = (rng.random((M, N)) + 0.1).astype(rdt)
        xt = torch.from_numpy(x).cuda()
        if rng.random() < 0.3:          # a view with a row pitch wider than the row (even offsets: the array is read as pairs)
            wide = torch.zeros((M, N + 8), dtype=xt.dtype, device='cuda')
            wide[:, 4:N + 4] = xt
            xt = wide[:, 4:N + 4]
        got = _ops.fft2(xt, direction=-1, scale=scale, in_shift=ins, out_shift=outs, epilogue=epi,
                        flags=L.PM_FLAG_REAL_INPUT | (L.PM_FLAG_NORM_DC if norm else 0)).cpu().numpy()
        F = np.fft.fft2(np.roll(x.astype(np.float64), (-ins[0], -ins[1]), axis=(0, 1)))
        if norm:
            F = F / F[0, 0]
        F = np.roll(F * scale, outs, axis=(0, 1))
        tol = 1e-10 if rdt == np.float64 else 3e-5
        if epi == L.PM_EPI_ARG:
            strong = np.abs(F) > 1e-3 * np.abs(F).max()
            err = np.max(np.abs(np.angle(np.exp(1j * (got - np.angle(F))))[strong])) / (1e-7 if rdt == np.float64 else 3e-3)
        else:
            ref = {L.PM_EPI_NONE: F, L.PM_EPI_ABS: np.abs(F), L.PM_EPI_ABS2: np.abs(F) ** 2}[epi]
            err = np.max(np.abs(got - ref)) / (np.max(np.abs(ref)) * tol * (2 if epi == L.PM_EPI_ABS2 else 1))
        worst = max(worst, err)
        if err > 1:
            nfail += 1
            print('FAIL hermitian', M, N, rdt.__name__, ins, outs, epi, norm, scale, err)
    lib.pm_set_tuning(b'r2c', 1)
    lib.pm_set_tuning(b'herm_t', -1)
    lib.pm_set_tuning(b'herm_t_fold', -1)
    print(f'fuzz_hermitian: {ncases} cases, {nfail} failures, worst err/tol {worst:.3f}')
    return nfail


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    lib = L.load()
    sizes = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 3, 5, 9, 12, 20, 36, 100, 96, 127, 160, 200, 224, 384, 1000, 1536, 45, 77, 143, 250, 360, 729, 1001, 1250,
             2592, 323, 1020, 1900,   # engine, direct, Bluestein, radix-R step and mixed-radix (composite; round 4: primes 17 / 19 too) lengths
             500, 900, 1500, 1600, 1800, 2000, 2500, 3000, 384, 768, 1152, 1280, 2304, 2560, 3072]   # round 5: lengths of the composite register engine (fft_ce.h)
    worst = 0.0
    nfail = 0
    for case in range(ncases):
        big = rng.random() < 0.25
        M = int(rng.choice([2048, 4096, 8192] if big else sizes))
        N = int(rng.choice([2048, 4096] if big else sizes))
        if M * N > (1 << 23):
            N = 2048 if M == 8192 else N
            M = min(M, 4096) if N == 4096 else M
        if rng.random() < 0.04:     # round 4: a composite above 8192 beside a short axis (one radix-R step around mixed-radix sub-transforms)
            M, N = (int(rng.choice([9000, 10000, 12000])), int(rng.choice([64, 96, 100]))) if rng.random() < 0.5 else (int(rng.choice([64, 100])), int(rng.choice([10000, 20000])))
        if rng.random() < 0.05:     # round 5: the long plans of the composite register engine beside a short axis
            big_ce = int(rng.choice([3600, 4000, 4500, 5000, 5120, 6000, 6144, 8000]))
            M, N = (big_ce, int(rng.choice([64, 100, 500]))) if rng.random() < 0.5 else (int(rng.choice([64, 100, 900])), big_ce)
        lib.pm_set_tuning(b'mix_pad', int(rng.random() < 0.7))
        lib.pm_set_tuning(b'mix_engine', int(rng.random() < 0.75))     # ... whose lengths also keep running on the general kernel
        cdt = np.complex64 if rng.random() < 0.5 else np.complex128
        rdt = np.float32 if cdt == np.complex64 else np.float64
        m = M if rng.random() < 0.6 else int(rng.integers(1, M + 1))
        n = N if rng.random() < 0.6 else int(rng.integers(1, N + 1))
        in_off = (int(rng.integers(0, M - m + 1)), int(rng.integers(0, N - n + 1)))
        shift = lambda s: int(rng.choice([0, s // 2, int(rng.integers(0, s))]))
        in_shift = (shift(M), shift(N))
        out_shift = (shift(M), shift(N))
        om = M if rng.random() < 0.6 else int(rng.integers(1, M + 1))
        on = N if rng.random() < 0.6 else int(rng.integers(1, N + 1))
        out_off = (int(rng.integers(0, M - om + 1)), int(rng.integers(0, N - on + 1)))
        direction = -1 if rng.random() < 0.6 else +1
        kind = rng.choice(['complex', 'real', 'synth']) if direction < 0 else 'complex'
        B = int(rng.choice([0, 0, 0, 2, 3])) if M * N <= (1 << 20) else 0
        epi = int(rng.choice([0, 0, 1]))
        fold = int(rng.choice([-1, 0, 1]))
        lib.pm_set_tuning(b'fold', fold)
        # routes of awkward lengths: with the native length lowered to 32 and the Bluestein path opened from 20 points, 64 / 128
        # take the radix-2 / radix-4 step of the 16384 / 32768-point path and 20 .. 64 the long both-axes Bluestein form
        route = str(rng.choice(['default', 'default', 'default', 'small_native', 'unfused', 'bluestein', 'radix_r']))
        lib.pm_set_tuning(b'big_native_log', 5 if route == 'small_native' else 13)
        lib.pm_set_tuning(b'blue_min', 20 if route == 'small_native' else 96)
        lib.pm_set_tuning(b'blue_fuse', 0 if route == 'unfused' else 1)
        lib.pm_set_tuning(b'mix', 0 if route in ('unfused', 'bluestein', 'small_native') else (2 if route == 'radix_r' else 1))   # composite lengths: their own kernel, or round 2's routes
        scale = float(rng.choice([1.0, 1.0 / np.sqrt(M * N)]))
        shp = (B, m, n) if B else (m, n)
        amp = None
        if kind == 'complex':
            x = (rng.standard_normal(shp) + 1j * rng.standard_normal(shp)).astype(cdt)
            xs = x
        elif kind == 'real':
            x = rng.standard_normal(shp).astype(rdt)
            xs = x.astype(cdt)
        else:
            if cdt != np.complex64 or B or not (((N & (N - 1)) == 0 and 2 <= N <= 8192) or (_ops._is_mix_length(N) and route in ('default', 'radix_r'))):
                kind = 'real'
                x = rng.standard_normal(shp).astype(rdt)
                xs = x.astype(cdt)
            else:
                x = (rng.standard_normal(shp) * 300).astype(np.float32)
                amp = rng.random(shp).astype(np.float32) if rng.random() < 0.7 else None
                k = 2 * np.pi / 0.55 / 1e3
                xs = ((1 if amp is None else amp.astype(np.float64)) * np.exp(1j * k * x.astype(np.float64)))
        xd = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        kw = dict(direction=direction, scale=scale, shape=(M, N), in_off=in_off, in_shift=in_shift, out_shape=(om, on),
                  out_off=out_off, out_shift=out_shift, epilogue=epi)
        if kind == 'synth':
            kw['synth'] = (None if amp is None else torch.from_numpy(amp).cuda(), k)
        try:
            got = _ops.fft2(xd, **kw).cpu().numpy()
        except Exception as exc:
            print('case', case, 'EXC', repr(exc)[:200], (M, N, m, n, cdt.__name__, kind, B, fold))
            nfail += 1
            continue
        fields = xs if B else xs[None]
        err = 0.0
        for b in range(fields.shape[0]):
            F = ref_fft2(fields[b].astype(np.complex128), M, N, in_off, in_shift, direction, scale)
            W = window(F, (om, on), out_off, out_shift)
            want = (W.real ** 2 + W.imag ** 2) if epi else W
            g = got[b] if B else got
            den = max(np.abs(want).max(), 1e-30)
            err = max(err, float(np.abs(g - want).max() / den))
        tol = (3e-5 if cdt == np.complex64 else 1e-10) * (4 if epi else 1)
        ok = err < tol
        worst = max(worst, err / tol)
        if not ok:
            nfail += 1
            print('case', case, 'FAIL err', err, (M, N, m, n, in_off, in_shift, (om, on), out_off, out_shift, direction, cdt.__name__, kind, B, epi, fold, route))
    for key, val in ((b'fold', -1), (b'big_native_log', 13), (b'blue_min', 96), (b'blue_fuse', 1), (b'mix', 1), (b'mix_pad', 1), (b'mix_engine', 1)):
        lib.pm_set_tuning(key, val)
    print(f'fuzz_fft2: {ncases} cases, {nfail} failures, worst err/tol {worst:.3f}')
    nfail += fuzz_fused(max(20, ncases // 2), rng, lib)
    nfail += fuzz_gemm(max(20, ncases // 2), rng, lib)
    nfail += fuzz_fft1(max(30, ncases // 2), rng, lib)
    nfail += fuzz_real_conv(max(20, ncases // 3), rng, lib)
    nfail += fuzz_spectral(max(12, ncases // 4), rng, lib)
    nfail += fuzz_real_any(max(30, ncases // 3), rng, lib)
    nfail += fuzz_hermitian(max(30, ncases // 3), rng, lib)
    return 1 if nfail else 0

if __name__ == '__main__':
    sys.exit(main())
