"""Thin Python wrappers over the C ABI: build descriptors, allocate outputs, enqueue.

Everything here operates on torch tensors resident in HBM and enqueues on
torch's current stream.  No arithmetic happens in Python.
"""
import ctypes
import math
import threading

import torch

from . import _lib as L
from .graph import sequenced


# The kernels write through raw pointers, which torch's version counters do not see: a caller-supplied `out=` tensor is marked
# modified here, so caches keyed on (tensor identity, version) -- polychromatic.packed_pupil -- notice maps rewritten in place.
_bump = torch.autograd.graph.increment_version


def _axis(n, length=None, off=0, shift=0):
    return L.pm_axis(int(n), int(n if length is None else length), int(off), int(shift))


def _fill_views(d, x, shape, in_off, in_shift, out_shape, out_off, out_shift):
    """Common part of the 2-D descriptors: views, leading dimensions and the batch (leading axis of a 3-D x)."""
    if x.dim() not in (2, 3):
        raise ValueError('2-D transforms take a 2-D array or a (batch, rows, cols) stack of them')
    if x.stride(-1) != 1:
        x = x.contiguous()
    m, n = x.shape[-2:]
    M, N = (m, n) if shape is None else shape
    om, on = (M, N) if out_shape is None else out_shape
    d.dtype = L.code(x)
    d.in_y = _axis(M, m, in_off[0], in_shift[0])
    d.in_x = _axis(N, n, in_off[1], in_shift[1])
    d.out_y = _axis(M, om, out_off[0], out_shift[0])
    d.out_x = _axis(N, on, out_off[1], out_shift[1])
    d.in_ld = x.stride(-2) if m > 1 else n
    if not x.is_complex():
        d.flags |= L.PM_FLAG_REAL_INPUT      # float32 / float64 field read as it is
    if x.dim() == 3:
        d.batch = x.shape[0]
        d.in_bstride = x.stride(0) if x.shape[0] > 1 else m * n
    return x, (M, N), (om, on)


def _fill_mul(d, x, mul, mul_x, mul_conj, keep):
    """Multiplier: full (M, N) array or the (hy, hx) vector pair; with a batch, optionally one per field."""
    batched = x.dim() == 3
    if mul is not None and mul_x is not None:
        d.mul_kind = L.PM_MUL_SEPARABLE
        if batched and mul.dim() == 2:
            mul, mul_x = mul.contiguous(), mul_x.contiguous()
            d.mul_bstride, d.mul_x_bstride = mul.stride(0), mul_x.stride(0)
        d.mul = mul.data_ptr()
        d.mul_x = mul_x.data_ptr()
        keep += [mul, mul_x]
    elif mul is not None:
        d.mul_kind = L.PM_MUL_FULL
        if batched and mul.dim() == 3:
            d.mul_bstride = mul.stride(0)
        d.mul = mul.data_ptr()
        d.mul_ld = mul.stride(-2) if mul.shape[-2] > 1 else mul.shape[-1]
        keep.append(mul)
    d.mul_conj = 1 if mul_conj else 0


def _is_mix_length(n):
    """lengths the mixed-radix kernel takes (csrc/fft_mixed.h): 32 .. 8192, not a power of two, primes <= 19"""
    if n < 32 or n > 8192 or (n & (n - 1)) == 0:
        return False
    for p in (2, 3, 5, 7, 11, 13, 17, 19):
        while n % p == 0:
            n //= p
    return n == 1


def synth_supported(opd, amp, N):
    """Whether pm_fft2 can synthesise amp * exp(i k opd) while loading (PM_FLAG_SYNTH_INPUT): a 2-D float32 / float64 OPD map
    (complex64 / complex128 transform), a row length that is a power of two or (round 4) a composite of primes <= 19 -- the two row
    kernels whose loaders synthesise --, real / bool amplitude."""
    if opd.dim() != 2 or opd.dtype not in (torch.float32, torch.float64) or not (_is_pow2_engine(N) or _is_mix_length(N)):
        return False
    if not _is_pow2_engine(N) and opd.stride(0) >= 1 << 24:
        return False
    return amp is None or (amp.dtype in _AMP_CODE and amp.dim() == 2 and amp.shape == opd.shape and amp.stride(-1) == 1)


# Filled descriptors and their workspace sizes, keyed by everything in them that is not a pointer or a per-call scalar: a
# repeated propagation (the inner loop of every model) then costs one struct copy, four pointer stores and ONE library
# call pair instead of ~40 ctypes field stores.
_fft2_plans = {}


def _check_out(out, x, oshape, odt):
    """A caller-supplied output / accumulator must be exactly what the kernel writes (it only receives a pointer and a
    leading dimension): anything else would be silent corruption or an out-of-bounds write."""
    if not isinstance(out, torch.Tensor) or out.device != x.device:
        raise ValueError('fft2: `out` must be a tensor on the device of the input')
    if out.dtype != odt:
        raise ValueError(f'fft2: `out` must be {odt} for this input and epilogue, got {out.dtype}')
    if tuple(out.shape) != tuple(oshape):
        raise ValueError(f'fft2: `out` must have shape {tuple(oshape)}, got {tuple(out.shape)}')
    if out.stride(-1) != 1:
        raise ValueError('fft2: the last axis of `out` must be contiguous')


def fft2(x, *, synth=None, spectral=None, **kw):
    """Fused 2-D transform (pm_fft2); see _fft2_call for the arguments.

    synth (pupil synthesis inside the row pass) is a property of the ROUTE the library's planner picks -- this module's
    synth_supported() restates the planner's test without its inputs (the mix / mix_min knobs, a tuning_local block, workspace
    limits), so it can say yes where the planner then says PM_ERR_UNSUPPORTED.  When that happens for a single 2-D field the pupil
    is materialised (pm_pupil_synth, what the call would have cost before the fusion existed) and the transform runs on it:
    a disagreement costs time, never an exception."""
    if synth is None or spectral is not None or x.dim() != 2:
        return _fft2_call(x, synth=synth, spectral=spectral, **kw)
    try:
        return _fft2_call(x, synth=synth, **kw)
    except NotImplementedError:
        cdt = L._COMPLEX_OF[x.dtype] if not x.is_complex() else x.dtype
        if isinstance(synth[0], str):       # ('packed', k): (amplitude, OPD) pairs
            pairs = torch.view_as_real(x)
            field = pupil_synth(pairs[..., 0].contiguous(), pairs[..., 1].contiguous(), synth[1], cdt)
        else:
            field = pupil_synth(synth[0], x, synth[1], cdt)
        return _fft2_call(field, **kw)


def _fft2_call(x, *, direction, scale, shape=None, in_off=(0, 0), in_shift=(0, 0),
               out_shape=None, out_off=(0, 0), out_shift=(0, 0), epilogue=L.PM_EPI_NONE,
               mul=None, mul_x=None, mul_conj=False, out=None, weight=1.0, flags=0, synth=None, spectral=None):
    """Fused 2-D transform (pm_fft2).

    x          : (m, n) complex tensor, or a (B, m, n) stack transformed in one launch pair; each field sits at
                 `in_off` inside the logical `shape` = (M, N) array (zero elsewhere), which is then rotated by
                 `in_shift`
    out_shape  : stored window of the output (crop), at `out_off` of the rotated result
    mul, mul_x : None | full (M, N) multiplier | (column vector hy (M,), row vector hx (N,)); with a stack,
                 (B, M, N) / ((B, M), (B, N)) give one multiplier per field
    synth      : (amp | None, k): x is the real float32 OPD map and the transformed field is amp * exp(i k x),
                 synthesised while the row pass loads it (see synth_supported); ('packed', k): x is a complex64 tensor
                 holding (amplitude, OPD) pairs (pack_amp_opd)
    spectral   : (k values, weights) with synth and PM_EPI_ABS2_ACCUM: out += sum_b weights[b] |transform at k[b]|^2, the whole
                 wavelength loop in one call (pm_fft2_spectral: groups of wavelengths share one launch pair); synth[1] is ignored
    out        : optional result / accumulator (PM_EPI_ABS2_ACCUM); must match the dtype and shape the call produces
    """
    lib = L.load()
    if x.dim() not in (2, 3):
        raise ValueError('2-D transforms take a 2-D array or a (batch, rows, cols) stack of them')
    if x.stride(-1) != 1:
        x = x.contiguous()
    m, n = x.shape[-2:]
    M, N = (m, n) if shape is None else shape
    om, on = (M, N) if out_shape is None else out_shape
    cdt = L.cdtype_of(x)
    odt = cdt if epilogue == L.PM_EPI_NONE else L._REAL_OF[cdt]
    oshape = (om, on) if x.dim() == 2 else (x.shape[0], om, on)
    if out is None:
        out = torch.empty(oshape, dtype=odt, device=x.device)
    else:
        _check_out(out, x, oshape, odt)
        _bump(out)
    packed = synth is not None and isinstance(synth[0], str)
    amp = synth[0] if (synth is not None and not packed) else None
    if mul is not None and mul_x is not None and x.dim() == 3 and mul.dim() == 2:
        mul, mul_x = mul.contiguous(), mul_x.contiguous()      # per-field (hy, hx) vectors: rows of two (B, .) arrays
    key = (x.dtype, tuple(x.shape), x.stride(), M, N, tuple(in_off), tuple(in_shift), om, on, tuple(out_off), tuple(out_shift),
           direction, epilogue, flags, float(scale), bool(mul_conj), out.stride(),
           None if synth is None else (True, packed, None if amp is None else (amp.dtype, amp.stride())),
           None if mul is None else (tuple(mul.shape), mul.stride()), None if mul_x is None else (tuple(mul_x.shape), mul_x.stride()))
    plan = _fft2_plans.get(key)
    if plan is None:
        d = L.pm_fft2_desc()
        d.flags = flags
        _fill_views(d, x, shape, in_off, in_shift, out_shape, out_off, out_shift)
        d.direction = direction
        d.epilogue = epilogue
        d.scale = float(scale)
        d.weight = 1.0
        if synth is not None:
            d.flags = (d.flags & ~L.PM_FLAG_REAL_INPUT) | L.PM_FLAG_SYNTH_INPUT | (L.PM_FLAG_SYNTH_PACKED if packed else 0)
            if amp is not None:
                d.synth_amp_dtype = _AMP_CODE[amp.dtype]
                d.synth_amp_ld = amp.stride(0) if amp.shape[0] > 1 else amp.shape[1]
        keep = []
        _fill_mul(d, x, mul, mul_x, mul_conj, keep)
        d.out_ld = out.stride(-2) if om > 1 else on
        if x.dim() == 3:
            d.out_bstride = out.stride(0) if out.shape[0] > 1 else om * d.out_ld
        if synth is not None:     # the workspace query validates the descriptor: give it the pointers of this call
            d.synth_k = float(synth[1])
            d.synth_amp = amp.data_ptr() if amp is not None else None
        nbytes = lib.pm_fft2_workspace(ctypes.byref(d))
        if nbytes == 0:
            L.check(lib.pm_fft2(ctypes.byref(d), L.ptr(x), L.ptr(out), None, 0, L.stream_ptr()))  # raises with the reason
        plan = (d,)
        if len(_fft2_plans) > 512:
            _fft2_plans.clear()
        _fft2_plans[key] = plan
    d = L.pm_fft2_desc.from_buffer_copy(plan[0])
    d.weight = float(weight)
    if synth is not None:
        d.synth_k = float(synth[1])
        d.synth_amp = amp.data_ptr() if amp is not None else None
    if mul is not None:
        d.mul = mul.data_ptr()
        if mul_x is not None:
            d.mul_x = mul_x.data_ptr()
    if spectral is not None:
        ks, wts = spectral
        cnt = len(ks)
        if synth is None or epilogue != L.PM_EPI_ABS2_ACCUM or len(wts) != cnt:
            raise ValueError('spectral=(k, weights) needs synth, the accumulate epilogue and as many weights as wavenumbers')
        ka = (ctypes.c_double * cnt)(*[float(v) for v in ks])
        wa = (ctypes.c_double * cnt)(*[float(v) for v in wts])
        nbytes = lib.pm_fft2_spectral_workspace(ctypes.byref(d), cnt)
        ws = L.workspace(nbytes)
        L.check(lib.pm_fft2_spectral(ctypes.byref(d), cnt, ka, wa, x.data_ptr(), out.data_ptr(), ws.data_ptr() if ws is not None else None,
                                     nbytes, L.stream_ptr()))
        return out
    nbytes = lib.pm_fft2_workspace(ctypes.byref(d))      # not cached: it follows the tuning knobs (pm_set_tuning)
    ws = L.workspace(nbytes)
    rc = lib.pm_fft2(ctypes.byref(d), x.data_ptr(), out.data_ptr(), ws.data_ptr() if ws is not None else None, nbytes, L.stream_ptr())
    if rc:
        L.check(rc)
    return out


def real_pairs_ok(x):
    """Whether a real 2-D array can be read as rows of complex pairs (fft2_real): float32 / float64, contiguous rows of EVEN length, even
    row pitch, base address aligned like a complex element."""
    return (isinstance(x, torch.Tensor) and not x.is_complex() and x.dim() == 2 and x.dtype in (torch.float32, torch.float64) and
            x.shape[1] >= 2 and x.shape[1] % 2 == 0 and x.shape[0] >= 1 and x.stride(1) == 1 and (x.stride(0) % 2 == 0 or x.shape[0] == 1) and
            x.data_ptr() % (2 * x.element_size()) == 0)


def fft2_real(x, *, scale=1.0, in_shift=(0, 0), out_shift=(0, 0), epilogue=L.PM_EPI_NONE, norm_dc=False):
    """fft2 of a REAL (M, N) array, N even, on any lengths: the array read as (M, N/2) complex pairs, one half-size pm_fft2 on whatever
    route those lengths take, and one untangling sweep (pm_r2c_untangle) that also applies the input rotation, the optional division by
    the DC bin, `scale`, the epilogue (complex, |.|, |.|^2, angle) and the output rotation.  What the library's Hermitian path does for
    powers of two, for every other size (1000^2, 3000^2, 1536 x 2000 ...): prysm/otf.py:28-33, 62-135."""
    lib = L.load()
    if not real_pairs_ok(x):
        raise ValueError('fft2_real: a real float32 / float64 2-D array with contiguous rows of even length is required')
    M, N = x.shape
    z = torch.view_as_complex(x.as_strided((M, N // 2, 2), (x.stride(0) if M > 1 else N, 2, 1)))
    zf = _fft2_call(z, direction=-1, scale=1.0)
    cdt = zf.dtype
    out = torch.empty((M, N), dtype=cdt if epilogue == L.PM_EPI_NONE else L._REAL_OF[cdt], device=x.device)
    L.check(lib.pm_r2c_untangle(L._COMPLEX_CODE[cdt], M, N, L.ptr(zf), zf.stride(0), int(in_shift[0]) % M, int(in_shift[1]) % N,
                                int(out_shift[0]) % M, int(out_shift[1]) % N, int(epilogue), 1 if norm_dc else 0, float(scale),
                                L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


_REGISTER_ROUTES = {}


def on_register_engine(M, N, cdtype):
    """Whether BOTH passes of a plain complex (M, N) transform run on the composite register engine (csrc/fft_ce.h) -- the composite
    grids whose (B, M, N) stacks go out as one launch pair.  Asked of the library's planner (pm_plan_explain: host logic, no GPU work),
    once per shape, precision and set of open tuning_local blocks OF THE CALLING THREAD (the knobs are per thread: an answer cached
    under another thread's block, or under a process-wide epoch, could be wrong for this one -- ADVICE r5).  A process-wide
    pm_set_tuning after the first question is not seen: call _REGISTER_ROUTES.clear()."""
    key = (int(M), int(N), cdtype, L.tuning_local.signature())
    hit = _REGISTER_ROUTES.get(key)
    if hit is None:
        lib = L.load()
        d = L.pm_fft2_desc()
        d.dtype, d.direction = L._COMPLEX_CODE[cdtype], -1
        d.in_y, d.in_x, d.in_ld = L.pm_axis(M, M, 0, 0), L.pm_axis(N, N, 0, 0), N
        d.out_y, d.out_x, d.out_ld = L.pm_axis(M, M, 0, 0), L.pm_axis(N, N, 0, 0), N
        buf = ctypes.create_string_buffer(256)
        hit = lib.pm_plan_explain(ctypes.byref(d), 0, buf, 256) == 0 and buf.value.count(b'mixed-radix-registers') == 2
        if len(_REGISTER_ROUTES) > 256:
            _REGISTER_ROUTES.clear()
        _REGISTER_ROUTES[key] = hit
    return hit


def pack_amp_opd(amp, opd):
    """(amplitude, OPD) pairs in the OPD map's precision as a complex tensor -- the input of fft2(..., synth=('packed', k)).  A loop
    over wavelengths packs its two maps once and then reads ONE element per sample in every transform."""
    rd = opd.dtype if opd.dtype in (torch.float32, torch.float64) else torch.float32
    a = torch.ones_like(opd, dtype=rd) if amp is None else amp.to(rd)
    return torch.view_as_complex(torch.stack((a, opd.to(rd)), dim=-1).contiguous())


def _is_pow2_engine(n):
    return 2 <= n <= 8192 and (n & (n - 1)) == 0


def fft2_mul_ifft2(x, *, scale, mul, mul_x=None, mul_conj=False, shape=None, in_off=(0, 0), in_shift=(0, 0),
                   out_shape=None, out_off=(0, 0), out_shift=(0, 0), real_out=False):
    """window(ifft2(fft2(pad(x)) * H)) * scale.

    real_out: for a REAL x return the real part of the result as a real tensor -- on half spectra end to end when the shapes allow
    (PM_FLAG_REAL_OUTPUT: unpadded power-of-two sizes, a full multiplier), else `.real` of the complex result.

    Power-of-two transform sizes run the fused three-pass kernel chain (pm_fft2_mul_ifft2: the multiply and
    both column transforms happen in registers); so do composite grids whose column length has primes <= 19 (round 4: the middle
    pass keeps the column in LDS through forward stages, multiplier and transposed stages); what is left composes two pm_fft2 calls.
    """
    lib = L.load()
    m, n = x.shape[-2:]
    M, N = (m, n) if shape is None else shape
    def composed():
        # two fused transforms, the multiplier in the first one's column store (measured, profiles/r01/fused_as.log: the 3-pass chain wins
        # at every engine size, e.g. 4096^2 complex128 448 vs 473 us, 2048^2 complex64 55 vs 75 us)
        F = fft2(x, direction=-1, scale=1.0, shape=shape, in_off=in_off, in_shift=in_shift, mul=mul, mul_x=mul_x,
                 mul_conj=mul_conj)
        r = fft2(F, direction=+1, scale=scale, out_shape=out_shape, out_off=out_off, out_shift=out_shift)
        return r.real if (real_out and not x.is_complex()) else r
    pow2 = _is_pow2_engine(M) and _is_pow2_engine(N)
    if not pow2 and x.dim() != 2:
        return composed()
    d = L.pm_fft2_desc()
    x, (M, N), (om, on) = _fill_views(d, x, shape, in_off, in_shift, out_shape, out_off, out_shift)
    d.direction = -1
    d.scale = float(scale)
    d.weight = 1.0
    keep = [x]
    _fill_mul(d, x, mul, mul_x, mul_conj, keep)
    oshape = (om, on) if x.dim() == 2 else (x.shape[0], om, on)
    if real_out and not x.is_complex() and x.dim() == 2 and mul_x is None and x.data_ptr() % (2 * x.element_size()) == 0:
        # the half-spectrum chain: the workspace query answers whether this descriptor is one it takes
        d.flags |= L.PM_FLAG_REAL_OUTPUT
        out = torch.empty(oshape, dtype=L._REAL_OF[L.cdtype_of(x)], device=x.device)
        d.out_ld = out.stride(-2) if om > 1 else on
        nbytes = lib.pm_fft2_mul_ifft2_workspace(ctypes.byref(d))
        if nbytes:
            ws = L.workspace(int(nbytes))
            L.check(lib.pm_fft2_mul_ifft2(ctypes.byref(d), L.ptr(x), L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr()))
            return out
        d.flags &= ~L.PM_FLAG_REAL_OUTPUT
    out = torch.empty(oshape, dtype=L.cdtype_of(x), device=x.device)
    d.out_ld = out.stride(-2) if om > 1 else on
    if x.dim() == 3:
        d.out_bstride = out.stride(0) if out.shape[0] > 1 else om * d.out_ld
    nbytes = lib.pm_fft2_mul_ifft2_workspace(ctypes.byref(d))
    if not nbytes and not pow2:
        # not a shape of the composite-grid chain (a column length with a prime above 13, a Bluestein row length ...)
        del out
        return composed()
    ws = L.workspace(max(int(nbytes), 16))
    L.check(lib.pm_fft2_mul_ifft2(ctypes.byref(d), L.ptr(x), L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr()))
    return out.real if (real_out and not x.is_complex()) else out


def fft1(x, n=None, axis=-1, direction=-1, scale=1.0, out_len=None, out_off=0, in_off=0):
    """Batched 1-D transform along `axis` of a 2-D tensor (pm_fft1).

    The input is zero padded (placed at `in_off`) or truncated to length n; the stored output is the
    window [out_off, out_off + out_len) of the n bins.  direction -1: exp(-2 pi i ..); +1: exp(+..).
    """
    lib = L.load()
    if x.dim() == 1:
        return fft1(x[None, :], n, 1, direction, scale, out_len, out_off, in_off)[0]
    axis = axis % 2
    rows, cols = x.shape
    length = rows if axis == 0 else cols
    n = length if n is None else int(n)
    batch = cols if axis == 0 else rows
    olen = n if out_len is None else int(out_len)
    t_in = _axis(n, min(length, n - in_off), in_off, 0)
    t_out = _axis(n, olen, out_off, 0)
    oshape = (olen, cols) if axis == 0 else (rows, olen)
    out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    nbytes = lib.pm_fft1_workspace(L.code(x), axis, batch, n)   # non-zero: a non power-of-two length on the Bluestein path
    ws = L.workspace(nbytes) if nbytes else None
    L.check(lib.pm_fft1_ws(L.code(x), direction, axis, batch, ctypes.byref(t_in), ctypes.byref(t_out), float(scale),
                           L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), L.ptr(ws), nbytes, L.stream_ptr()))
    return out


def czt_vectors(N, M, K, shift, half, cdtype):
    """The chirps (b, a, h) of one chirp-Z axis from its scalars in one launch (pm_czt_vectors)."""
    lib = L.load()
    dev = L.device()
    b = torch.empty(N, dtype=cdtype, device=dev)
    a = torch.empty(M, dtype=cdtype, device=dev)
    h = torch.empty(K, dtype=cdtype, device=dev)
    L.check(lib.pm_czt_vectors(L.code(b), N, M, K, float(shift), float(half), L.ptr(b), L.ptr(a), L.ptr(h), L.stream_ptr()))
    return b, a, h


def czt_axis(x, K, axis, H, *, pre=None, post=None, out_len, in_off=0, out_off=0, conj=False, scale=1.0):
    """One axis of a chirp-Z transform in one kernel (pm_czt_axis): scale * post * IFFT_K(FFT_K(pad_K(pre * x)) * H)[out_off : out_off
    + out_len] along `axis` of the 2-D tensor x; `conj` conjugates pre, H and post (the adjoint).  K must be an engine length."""
    lib = L.load()
    axis = axis % 2
    rows, cols = x.shape
    nseq, in_len = (rows, cols) if axis == 1 else (cols, rows)
    oshape = (rows, out_len) if axis == 1 else (out_len, cols)
    out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    c = 1 if conj else 0
    L.check(lib.pm_czt_axis(L.code(x), axis, nseq, int(K), in_len, int(in_off), int(out_len), int(out_off), L.ptr(pre), c, L.ptr(H), c,
                            L.ptr(post), c, float(scale), L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def fft1_ramp(x, K, axis, direction, *, pre=None, post=None, out_len, conj=False, scale=1.0):
    """One axis of an FFT-accelerated DFT in one kernel (pm_fft1_ramp): scale * post * T_K(pad_K(pre * x))[:out_len] along `axis` of
    the 2-D tensor x, T_K unnormalised with the sign of `direction`; `conj` conjugates pre and post (the adjoint).  K must be an
    engine length."""
    lib = L.load()
    axis = axis % 2
    rows, cols = x.shape
    nseq, in_len = (rows, cols) if axis == 1 else (cols, rows)
    oshape = (rows, out_len) if axis == 1 else (out_len, cols)
    out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    c = 1 if conj else 0
    L.check(lib.pm_fft1_ramp(L.code(x), int(direction), axis, nseq, int(K), in_len, 0, int(out_len), 0, L.ptr(pre), c, L.ptr(post), c,
                             float(scale), L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def cmul(a, b, conj_b=False):
    lib = L.load()
    out = torch.empty_like(a)
    rows, cols = a.shape
    L.check(lib.pm_cmul(L.code(a), 1 if conj_b else 0, rows, cols, L.ptr(a), a.stride(0), L.ptr(b), b.stride(0),
                        L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def rmul(r, a, scale=1.0):
    """scale * r * a for a REAL image r and a complex field a of the matching precision (pm_rmul; one sweep)."""
    lib = L.load()
    out = torch.empty_like(a)
    rows, cols = a.shape
    L.check(lib.pm_rmul(L.code(a), rows, cols, L.ptr(r), r.stride(0), L.ptr(a), a.stride(0), float(scale), L.ptr(out), out.stride(0),
                        L.stream_ptr()))
    return out


def scale_sep(x, row_vec=None, col_vec=None, row_conj=False, col_conj=False, scale=1.0):
    """out[i, j] = x[i, j] * row_vec[i] * col_vec[j] * scale (row_vec indexes rows, col_vec columns)."""
    lib = L.load()
    out = torch.empty_like(x)
    rows, cols = x.shape
    L.check(lib.pm_scale_sep(L.code(x), rows, cols, L.ptr(x), x.stride(0), L.ptr(row_vec), 1 if row_conj else 0,
                             L.ptr(col_vec), 1 if col_conj else 0, float(scale), L.ptr(out), out.stride(0),
                             L.stream_ptr()))
    return out


def abs2(x, out=None, weight=None):
    """|x|^2, or out += weight * |x|^2 when `out` and `weight` are given."""
    lib = L.load()
    rows, cols = x.shape
    acc = 0
    if out is None:
        out = torch.empty((rows, cols), dtype=L._REAL_OF[x.dtype], device=x.device)
    else:
        _bump(out)
        if weight is not None:
            acc = 1
    L.check(lib.pm_abs2(L.code(x), rows, cols, L.ptr(x), x.stride(0), L.ptr(out), out.stride(0), acc,
                        float(1.0 if weight is None else weight), L.stream_ptr()))
    return out


def abs_arg(x):
    """(|x|, angle(x)) of a complex 2-D array in one sweep (pm_abs_arg)."""
    lib = L.load()
    rows, cols = x.shape
    rd = L._REAL_OF[x.dtype]
    oabs = torch.empty((rows, cols), dtype=rd, device=x.device)
    oarg = torch.empty((rows, cols), dtype=rd, device=x.device)
    L.check(lib.pm_abs_arg(L.code(x), rows, cols, L.ptr(x), x.stride(0), L.ptr(oabs), oabs.stride(0), L.ptr(oarg), oarg.stride(0),
                           L.stream_ptr()))
    return oabs, oarg


def sum_modes(modes, weights, out=None, accumulate=False):
    """sum_b weights[b] * modes[b] over a (B, rows, cols) stack of real images (pm_sum_modes);
    with `out` and accumulate=True adds into `out`."""
    lib = L.load()
    if modes.dim() != 3 or modes.is_complex():
        raise ValueError('sum_modes takes a (B, rows, cols) stack of real images')
    if modes.stride(-1) != 1:
        modes = modes.contiguous()
    B, rows, cols = modes.shape
    w = [float(v) for v in weights]
    if len(w) != B:
        raise ValueError('one weight per mode is required')
    if out is None:
        out = torch.empty((rows, cols), dtype=modes.dtype, device=modes.device)
        accumulate = False
    else:
        _bump(out)
    code = L.PM_C64 if modes.dtype == torch.float32 else L.PM_C128
    if modes.dtype not in (torch.float32, torch.float64):
        raise TypeError('sum_modes: float32 or float64 images')
    arr = (ctypes.c_double * max(B, 1))(*w)
    L.check(lib.pm_sum_modes(code, B, rows, cols, L.ptr(modes), modes.stride(0) if B > 1 else rows * cols, modes.stride(1),
                             arr, 1 if accumulate else 0, L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def encircled_energy(mtf, df, radii_mm):
    """EE(r) for every r of radii_mm (host floats, mm) from the real centre-normalised MTF (pm_encircled_energy):
    a float64 device vector."""
    lib = L.load()
    if mtf.dim() != 2 or mtf.dtype not in (torch.float32, torch.float64):
        raise TypeError('encircled_energy: the MTF is a real 2-D float32 / float64 array')
    if mtf.stride(-1) != 1:
        mtf = mtf.contiguous()
    rows, cols = mtf.shape
    r = [float(v) for v in radii_mm]
    arr = (ctypes.c_double * max(len(r), 1))(*r)
    out = torch.empty(len(r), dtype=torch.float64, device=mtf.device)
    nbytes = lib.pm_encircled_energy_workspace()
    ws = L.workspace(nbytes)
    code = L.PM_C64 if mtf.dtype == torch.float32 else L.PM_C128
    L.check(lib.pm_encircled_energy(code, rows, cols, L.ptr(mtf), mtf.stride(0), float(df), len(r), arr, L.ptr(out), L.ptr(ws),
                                    ws.numel(), L.stream_ptr()))
    return out


def encircled_energy_adjoint(shape, df, radii_mm, ee_bar, rdtype):
    """sum_r ee_bar_r r J1(2 pi r nu) / nu df^2 on the frequency grid of `shape` (pm_encircled_energy_adjoint)."""
    lib = L.load()
    rows, cols = shape
    r = [float(v) for v in radii_mm]
    b = [float(v) for v in ee_bar]
    if len(r) != len(b):
        raise ValueError('one ee_bar value per radius is required')
    ra = (ctypes.c_double * max(len(r), 1))(*r)
    ba = (ctypes.c_double * max(len(b), 1))(*b)
    out = torch.empty((rows, cols), dtype=rdtype, device=L.device())
    code = L.PM_C64 if rdtype == torch.float32 else L.PM_C128
    L.check(lib.pm_encircled_energy_adjoint(code, rows, cols, float(df), len(r), ra, ba, L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def spline_prefilter(measurement, order):
    """B-spline coefficients of a measured complex map for orders 2 .. 5 (pm_spline_prefilter): complex128,
    (rows + 24, cols + 24) -- scipy's edge padding by 12 included."""
    lib = L.load()
    m = L.as_field(measurement)
    if not m.is_complex():
        m = m.to(L.cdtype_of(m))
    m = m.contiguous()
    coeff = torch.empty((m.shape[0] + 24, m.shape[1] + 24), dtype=torch.complex128, device=m.device)
    L.check(lib.pm_spline_prefilter(L.code(m), int(order), m.shape[0], m.shape[1], L.ptr(m), m.stride(0), L.ptr(coeff), coeff.stride(0),
                                    L.stream_ptr()))
    return coeff


def sample_map(measurement, dx, center, xf, yf, fill=1.0, order=1, coeff=None):
    """Complex map resampled at focal coordinates (pm_sample_map / pm_sample_spline): map_coordinates(order 0 .. 5, mode='nearest')
    inside the measured extent, `fill` (scalar or array of the shape of xf) outside.  xf / yf broadcast against each other."""
    lib = L.load()
    m = L.as_field(measurement)
    if not m.is_complex():
        m = m.to(L.cdtype_of(m))
    m = m.contiguous()
    rdt = torch.float32 if m.dtype == torch.complex64 else torch.float64
    xf, yf = L.as_device(xf).to(rdt), L.as_device(yf).to(rdt)
    shape = torch.broadcast_shapes(xf.shape, yf.shape)
    if len(shape) != 2:
        raise ValueError('sample_map: focal coordinates must broadcast to a 2-D grid')
    xb, yb = xf.expand(shape), yf.expand(shape)
    rows, cols = shape
    out = torch.empty(shape, dtype=m.dtype, device=m.device)
    fill_t, fre, fim = None, 0.0, 0.0
    if isinstance(fill, (int, float, complex)):
        fre, fim = float(complex(fill).real), float(complex(fill).imag)
    else:
        fill_t = torch.broadcast_to(L.as_device(fill).to(m.dtype), shape).contiguous()
    cxo, cyo = center
    if order >= 2:   # coeff: the prefiltered map (spline_prefilter), made once per measured map by the caller
        if coeff is None:
            coeff = spline_prefilter(m, order)
        L.check(lib.pm_sample_spline(L.code(m), int(order), m.shape[0], m.shape[1], L.ptr(coeff), coeff.stride(0), float(dx), float(cxo),
                                     float(cyo), rows, cols, L.ptr(xb), xb.stride(0), xb.stride(1), L.ptr(yb), yb.stride(0),
                                     yb.stride(1), L.ptr(fill_t) if fill_t is not None else None,
                                     fill_t.stride(0) if fill_t is not None else 0, fre, fim, L.ptr(out), out.stride(0), L.stream_ptr()))
        return out
    L.check(lib.pm_sample_map(L.code(m), int(order), m.shape[0], m.shape[1], L.ptr(m), m.stride(0), float(dx), float(cxo),
                              float(cyo), rows, cols, L.ptr(xb), xb.stride(0), xb.stride(1), L.ptr(yb), yb.stride(0),
                              yb.stride(1), L.ptr(fill_t) if fill_t is not None else None,
                              fill_t.stride(0) if fill_t is not None else 0, fre, fim, L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


_AMP_CODE = {torch.float32: L.PM_F32, torch.float64: L.PM_F64, torch.bool: L.PM_BOOL, torch.uint8: L.PM_BOOL}


def pupil_synth(amp, opd, k, cdtype, out=None):
    """amp * exp(i k opd); opd real tensor of the real dtype of `cdtype`; `out`: e.g. one field of a stack."""
    lib = L.load()
    rows, cols = opd.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=cdtype, device=opd.device)
    else:
        _bump(out)
    a_code, a_ld = L.PM_F32, cols
    if amp is not None:
        if amp.dtype not in _AMP_CODE:
            amp = amp.to(L._REAL_OF[cdtype])
        a_code, a_ld = _AMP_CODE[amp.dtype], amp.stride(0)
    L.check(lib.pm_pupil_synth(L._COMPLEX_CODE[cdtype], rows, cols, L.ptr(amp), a_code, a_ld, L.ptr(opd),
                               opd.stride(0), float(k), L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def quadratic_phase(x, y, c, cdtype):
    lib = L.load()
    rows, cols = x.shape
    out = torch.empty((rows, cols), dtype=cdtype, device=x.device)
    L.check(lib.pm_quadratic_phase(L._COMPLEX_CODE[cdtype], rows, cols, L.ptr(x), x.stride(0), L.ptr(y), y.stride(0),
                                   float(c), L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


# (shape, wavelength, dx, z, dtype, device, stream) -> (hy, hx): a model that steps the same distance again and again (a time series,
# the planes of a relay, an optimiser's forward passes) re-uses the two vectors instead of re-synthesising them (one launch of 5 us
# against a 320 us step at 4096^2 complex128 -- and nothing at all to wait for at small sizes, where the launch is the cost).  The key
# is scalars only, so nothing can go stale; the values are READ-ONLY by contract (angular_spectrum never hands them out).  Keyed on
# the stream because the vectors are produced on it: another stream would have to wait for that work.
_AS_TF_CACHE = {}
_AS_TF_CACHE_MAX = 16
_AS_TF_LOCK = threading.Lock()      # one pipeline per host thread is the advertised use: lookups, inserts and evictions are serialised


def clear_tf_cache(stream=None):
    """Forget the cached transfer-function vectors -- all of them, or those made on one stream (its raw handle): a destroyed stream's
    handle can be handed out again, and a new stream with an old handle must not find vectors it never waited for."""
    with _AS_TF_LOCK:
        if stream is None:
            _AS_TF_CACHE.clear()
        else:
            for k in [k for k in _AS_TF_CACHE if k[-1] == stream]:
                _AS_TF_CACHE.pop(k, None)


def as_tf_vectors(shape, wvl, dx, z, cdtype, cache=True):
    lib = L.load()
    rows, cols = shape
    key = (int(rows), int(cols), float(wvl), float(dx), float(z), cdtype, L._cur_dev(), L._cur_stream())
    if cache and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        cache = False       # tensors made during a graph capture live in the graph's pool: never handed to eager code
    if cache:
        with _AS_TF_LOCK:
            hit = _AS_TF_CACHE.get(key)
        if hit is not None:
            return hit
    hy = torch.empty(rows, dtype=cdtype, device=L.device())
    hx = torch.empty(cols, dtype=cdtype, device=L.device())
    L.check(lib.pm_as_tf_vectors(L._COMPLEX_CODE[cdtype], rows, cols, float(wvl), float(dx), float(z), L.ptr(hy),
                                 L.ptr(hx), L.stream_ptr()))
    if cache:
        with _AS_TF_LOCK:
            while len(_AS_TF_CACHE) >= _AS_TF_CACHE_MAX:
                _AS_TF_CACHE.pop(next(iter(_AS_TF_CACHE)), None)
            _AS_TF_CACHE[key] = (hy, hx)
    return hy, hx


def outer(hy, hx):
    lib = L.load()
    rows, cols = hy.numel(), hx.numel()
    out = torch.empty((rows, cols), dtype=hy.dtype, device=hy.device)
    L.check(lib.pm_outer(L.code(hy), rows, cols, L.ptr(hy), L.ptr(hx), L.ptr(out), out.stride(0), L.stream_ptr()))
    return out


def embed(x, out_shape, off, fill=0):
    """out = fill; out[off_y:off_y+m, off_x:off_x+n] = x (negative offsets crop)."""
    lib = L.load()
    m, n = x.shape
    om, on = out_shape
    out = torch.empty((om, on), dtype=x.dtype, device=x.device)
    es = x.element_size()
    if es == 2:   # float16: no kernel for 2-byte elements; widen is not acceptable silently
        raise NotImplementedError('pad2d of 2-byte element arrays is not supported')
    fill_t = torch.tensor([fill], dtype=x.dtype)   # host scalar, passed by pointer
    L.check(lib.pm_embed(es, m, n, L.ptr(x), x.stride(0) if m > 1 else n, om, on, int(off[0]), int(off[1]),
                         ctypes.c_void_p(fill_t.data_ptr()), L.ptr(out), out.stride(0) if om > 1 else on,
                         L.stream_ptr()))
    return out


_PAD_MODES = {'edge': 1, 'reflect': 2, 'symmetric': 3, 'wrap': 4}


def pad_index(x, out_shape, off, mode):
    """np.pad(x, ..., mode=mode) for the index-mapping modes ('edge', 'reflect', 'symmetric', 'wrap'): x lands at `off` of `out_shape`."""
    lib = L.load()
    m, n = x.shape
    om, on = out_shape
    if x.stride(-1) != 1:
        x = x.contiguous()
    es = x.element_size()
    if es == 2:
        raise NotImplementedError('pad2d of 2-byte element arrays is not supported')
    out = torch.empty((om, on), dtype=x.dtype, device=x.device)
    L.check(lib.pm_pad_index(es, _PAD_MODES[mode], m, n, L.ptr(x), x.stride(0) if m > 1 else n, om, on, int(off[0]), int(off[1]),
                             L.ptr(out), out.stride(0) if om > 1 else on, L.stream_ptr()))
    return out


def mdft_basis(f, x, sign, cdtype):
    """E[m, n] = exp(sign 2 pi i f[m] x[n])."""
    lib = L.load()
    M, N = f.numel(), x.numel()
    E = torch.empty((M, N), dtype=cdtype, device=f.device)
    L.check(lib.pm_mdft_basis(L._COMPLEX_CODE[cdtype], M, N, L.ptr(f), L.ptr(x), int(sign), L.ptr(E), E.stride(0),
                              L.stream_ptr()))
    return E


def mdft_basis_grid(M, N, f_step, f_shift, f_scale, x_step, sign, cdtype):
    """E[m, n] = exp(sign 2 pi i f[m] x[n]) for x = fftrange(N) * x_step, f = (fftrange(M) * f_step + f_shift) * f_scale, each
    rounded in the real type of cdtype as the elementwise array expressions would (pm_mdft_basis_grid)."""
    lib = L.load()
    E = torch.empty((M, N), dtype=cdtype, device=L.device())
    L.check(lib.pm_mdft_basis_grid(L._COMPLEX_CODE[cdtype], M, N, float(f_step), float(f_shift), float(f_scale), float(x_step),
                                   int(sign), L.ptr(E), E.stride(0), L.stream_ptr()))
    return E


def cgemm(A, B, opA=0, opB=0, alpha=1.0):
    """alpha * op(A) @ op(B) on the MFMA cores.  op: 0 none, 1 conj, 2 transpose, 3 conjugate transpose."""
    lib = L.load()
    M, K = (A.shape[1], A.shape[0]) if opA & 2 else A.shape
    K2, N = (B.shape[1], B.shape[0]) if opB & 2 else B.shape
    if K != K2:
        raise ValueError(f'matmul: inner dimensions differ ({K} vs {K2})')
    C = torch.empty((M, N), dtype=A.dtype, device=A.device)
    nbytes = lib.pm_cgemm_workspace(L.code(A), M, N, K)
    ws = L.workspace(nbytes)
    L.check(lib.pm_cgemm(L.code(A), opA, opB, M, N, K, float(alpha), L.ptr(A), A.stride(0), L.ptr(B), B.stride(0),
                         L.ptr(C), C.stride(0), L.ptr(ws), 0 if ws is None else ws.numel(), L.stream_ptr()))
    return C


def cgemm_abs2(A, B, opA=0, opB=0, alpha=1.0, out=None, weight=1.0):
    """weight * |alpha op(A) @ op(B)|^2 as a REAL image, added to `out` when given (pm_cgemm_abs2: the |.|^2 and the weighted
    accumulate in the product's epilogue, no complex result in memory).  Returns None when the shapes are not the LDS-DMA
    kernel's (the caller composes cgemm + abs2)."""
    lib = L.load()
    M, K = (A.shape[1], A.shape[0]) if opA & 2 else A.shape
    K2, N = (B.shape[1], B.shape[0]) if opB & 2 else B.shape
    if K != K2:
        raise ValueError(f'matmul: inner dimensions differ ({K} vs {K2})')
    if A.dtype != torch.complex64 or B.dtype != torch.complex64 or M % 64 or N % 64 or K % 16 or M < 64 or N < 64:
        return None
    acc = 1
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
        acc = 0
    else:
        if out.dtype != torch.float32 or tuple(out.shape) != (M, N) or out.stride(1) != 1 or out.device != A.device:
            raise ValueError('cgemm_abs2: `out` must be a float32 (M, N) image on the device of the operands')
    nbytes = lib.pm_cgemm_workspace(L.code(A), M, N, K)
    ws = L.workspace(nbytes)
    rc = lib.pm_cgemm_abs2(L.code(A), opA, opB, M, N, K, float(alpha), L.ptr(A), A.stride(0), L.ptr(B), B.stride(0), L.ptr(out),
                           out.stride(0), float(weight), acc, L.ptr(ws), 0 if ws is None else ws.numel(), L.stream_ptr())
    if rc == L.PM_ERR_UNSUPPORTED:
        return None
    L.check(rc)
    if acc:
        _bump(out)      # written: caches keyed on (identity, version) must notice (not before the early return above: nothing was written)
    return out


def ceil_half(d):
    return math.ceil(d / 2)


# Inside a ``prysm_amd.graph.sequence()`` block every array-level entry point asks the block for its stream (independent calls alternate
# between the streams of a ring, dependent ones follow their producer); outside of one the wrapper is one thread-local read.
for _name in ('fft2', 'fft2_real', 'pack_amp_opd', 'fft2_mul_ifft2', 'fft1', 'czt_vectors', 'czt_axis', 'fft1_ramp', 'cmul', 'rmul', 'scale_sep', 'abs2',
              'abs_arg', 'sum_modes', 'encircled_energy', 'encircled_energy_adjoint', 'spline_prefilter', 'sample_map', 'pupil_synth',
              'quadratic_phase', 'as_tf_vectors', 'outer', 'embed', 'pad_index', 'mdft_basis', 'mdft_basis_grid', 'cgemm', 'cgemm_abs2'):
    globals()[_name] = sequenced(globals()[_name])
del _name
