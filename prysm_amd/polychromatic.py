"""Polychromatic / multi-field driver: shard wavelengths across GPUs, one RCCL reduce.

The reference has no library function for this -- it is the user-level loop of
docs/source/how-tos/Polychromatic Propagation.ipynb (cell 3): for each wavelength
``from_amp_and_phase -> prepare_executor -> focus_dft -> intensity``, then
``polynomials.sum_of_2d_modes(components, weights)`` (prysm/polynomials/fitting.py:7-37), with the
advice to map wavelengths over devices (GPU and Exascale Computing.ipynb).  Each wavelength is
independent, so the path shards over wavelengths with NO data-path collective until the end: every
rank accumulates ``sum_k w_k |E_k|^2`` for its contiguous block of wavelengths in HBM, then ONE
sum-reduce of the real image (RCCL over xGMI; `backend="nccl"` is RCCL on ROCm) produces the
incoherent sum.  One process per GPU; world size 1 needs no process group.
"""
import math

import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world_size):
    """Contiguous block [lo, hi) of `n_items` owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def _reduce_image(acc, world, group, reduce_to_all, method='reduce'):
    """The one data-path collective: sum-reduce of the real image over the ranks (RCCL; gloo in the CPU tests).

    reduce_to_all           : all_reduce, every rank gets the image.
    root only, 'reduce'     : torch.distributed.reduce to the first rank of the group; the buffers of the other ranks are
                              scratch afterwards (gloo and RCCL both leave partial sums in them).
    root only, 'a2a'        : the fully connected xGMI form of SURVEY 8(e): every rank sends slice j of its image to rank j
                              (one all-to-all, 7 distinct links per GPU), sums the `world` slices it received in rank order
                              (pm_sum_modes: fixed order, bitwise reproducible) and the first rank gathers the reduced slices.
                              Needs numel % world == 0; falls back to 'reduce' otherwise.
    """
    if world <= 1:
        return acc
    if reduce_to_all:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        return acc
    root = dist.get_global_rank(group, 0) if group is not None else 0
    if method == 'a2a' and acc.numel() % world == 0 and acc.is_contiguous():
        from . import _ops
        per = acc.numel() // world
        send = acc.view(world, per)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=group)
        if acc.is_cuda:
            part = _ops.sum_modes(recv.view(world, 1, per), [1.0] * world).view(per)
        else:
            part = recv.sum(0)
        me = dist.get_rank()
        parts = [torch.empty_like(part) for _ in range(world)] if me == root else None
        dist.gather(part, parts, dst=root, group=group)
        if me == root:
            torch.cat(parts, out=acc.view(-1))
        return acc
    dist.reduce(acc, dst=root, op=dist.ReduceOp.SUM, group=group)
    return acc


def incoherent_sum(propagate, wavelengths, weights, *, group=None, reduce_to_all=True, out=None, reduce_method='reduce'):
    """Weighted incoherent sum over wavelengths (or fields), sharded over the ranks of `group`.

    propagate(wavelength, weight, acc) -> acc
        computes ``acc += weight * |E(wavelength)|^2`` (acc is None for the first item of a rank and
        must then be created); the fused-epilogue form ``focus_intensity(x, Q, out=acc, weight=w)``
        does this without materialising the field.
    Returns the summed image on every rank (reduce_to_all) or on the first rank of the group only (what the
    other ranks get back is scratch).
    """
    if len(wavelengths) != len(weights):
        raise ValueError('wavelengths and weights must have the same length')
    use_dist = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if use_dist else 0
    world = dist.get_world_size(group) if use_dist else 1
    lo, hi = shard_bounds(len(wavelengths), rank, world)
    acc = out
    for k in range(lo, hi):
        acc = propagate(float(wavelengths[k]), float(weights[k]), acc)
    if acc is None:
        raise ValueError('a rank received no wavelengths and no `out` buffer to define the image shape')
    return _reduce_image(acc, world, group, reduce_to_all, reduce_method)


def _fields_per_launch(shape, cdtype_bytes, Q, limit_bytes=1 << 30):
    """How many wavelengths go into one batched launch: bounded by ~1 GiB of stack (pupils + intensities)."""
    m, n = shape
    M, N = math.ceil(m * Q), math.ceil(n * Q)
    per_field = m * n * cdtype_bytes + M * N * cdtype_bytes // 2
    return max(1, min(64, limit_bytes // per_field))


def polychromatic_psf(amplitude, opd, wavelengths, weights, dx, efl, *, Q=None, focal_dx=None, samples=None,
                      kind='mdft', group=None, reduce_to_all=True, batched=None, reduce_method='reduce', spectral=True):
    """Polychromatic PSF of a pupil (amplitude, OPD in nm) -- the how-to's recipe on N GPUs.

    Q given            : FFT focus per wavelength with the |.|^2 fused into the transform (multi-field throughput
                         variant; focal sampling is chromatic, as the reference docs note).  With `batched` the
                         wavelengths of a rank are propagated as stacks -- one launch pair per stack, then one
                         weighted sum (sum_of_2d_modes) -- which is what makes <= 2048^2 transforms bandwidth-bound
                         instead of launch-bound; batched=False is the field-by-field loop with the accumulate
                         epilogue, the faster form from 4096^2 transforms (single fields take the folded kernels).
                         Default (None): stacks below 4096^2 transforms.
    spectral           : with Q and float maps of power-of-two width, a rank's wavelength loop runs as ONE call whose launch pairs
                         each cover a group of wavelengths (pm_fft2_spectral; False: one transform pair per wavelength).
    reduce_method      : 'reduce' (one torch.distributed.reduce) or 'a2a' (all-to-all of slices + ordered local sum + gather:
                         bitwise reproducible, one message per xGMI link) when only the first rank needs the image.
    focal_dx + samples : per-wavelength fixed-sampling focus (prepare_executor + focus_dft, `kind`),
                         all wavelengths on one focal grid -- the variant of the how-to.
    """
    from . import _lib as L
    from . import _ops
    from .propagation import Wavefront, focus_intensity
    from .propagation.wavefront import _synth_args

    amp = L.as_device(amplitude)
    phs = L.as_device(opd)

    packed = None
    if Q is not None and not batched:
        # the pupil is synthesised inside every wavelength's transform (float maps, power-of-two width): pack (amplitude, OPD)
        # once so each of those row passes reads one 8-byte element per sample instead of two 4-byte ones from two arrays
        probe = Wavefront.from_amp_and_phase(amp, phs, float(wavelengths[0]), dx)._fusable(Q) if len(wavelengths) else None
        if probe is not None and len(wavelengths) > 1:
            packed = _ops.pack_amp_opd(probe[0], probe[1])
    if Q is not None and batched is None:
        batched = (packed is None or not spectral) and math.ceil(amp.shape[-2] * Q) * math.ceil(amp.shape[-1] * Q) < 4096 * 4096
    if Q is not None and batched:
        if len(wavelengths) != len(weights):
            raise ValueError('wavelengths and weights must have the same length')
        use_dist = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if use_dist else 0
        world = dist.get_world_size(group) if use_dist else 1
        lo, hi = shard_bounds(len(wavelengths), rank, world)
        a, o, cd = _synth_args(amp, phs)
        m, n = o.shape
        M, N = math.ceil(m * Q), math.ceil(n * Q)
        acc = torch.zeros((M, N), dtype=L._REAL_OF[cd], device=o.device)
        step = _fields_per_launch((m, n), torch.empty((), dtype=cd).element_size(), Q)
        for b0 in range(lo, hi, step):
            b1 = min(hi, b0 + step)
            stack = torch.empty((b1 - b0, m, n), dtype=cd, device=o.device)
            for i, k in enumerate(range(b0, b1)):
                kk = 2 * math.pi / float(wavelengths[k]) / 1e3
                if a is not None and a.is_complex():
                    stack[i] = Wavefront.from_amp_and_phase(a, o, float(wavelengths[k]), dx).data
                else:
                    _ops.pupil_synth(a, o, kk, cd, out=stack[i])     # synthesised straight into the stack
            _ops.sum_modes(focus_intensity(stack, Q), [float(w) for w in weights[b0:b1]], out=acc, accumulate=True)
        return _reduce_image(acc, world, group, reduce_to_all, reduce_method)

    if packed is not None and spectral:
        # the rank's whole wavelength loop in one call: groups of wavelengths share a launch pair (the packed map is read once per
        # group, the image touched once per group -- csrc/fft_spectral.h)
        if len(wavelengths) != len(weights):
            raise ValueError('wavelengths and weights must have the same length')
        use_dist = dist.is_available() and dist.is_initialized()
        rank = dist.get_rank(group) if use_dist else 0
        world = dist.get_world_size(group) if use_dist else 1
        lo, hi = shard_bounds(len(wavelengths), rank, world)
        m, n = packed.shape
        acc = torch.zeros((math.ceil(m * Q), math.ceil(n * Q)), dtype=L._REAL_OF[packed.dtype], device=packed.device)
        if hi > lo:
            ks = [2 * math.pi / float(wavelengths[k]) / 1e3 for k in range(lo, hi)]
            focus_intensity(packed, Q, out=acc, synth=('packed', ks[0]), spectral=(ks, [float(w) for w in weights[lo:hi]]))
        return _reduce_image(acc, world, group, reduce_to_all, reduce_method)

    def propagate(wvl, w, acc):
        if packed is not None:
            syn = ('packed', 2 * math.pi / wvl / 1e3)
            if acc is None:
                first = focus_intensity(packed, Q, synth=syn)
                return first * w if w != 1.0 else first
            return focus_intensity(packed, Q, out=acc, weight=w, synth=syn)
        wf = Wavefront.from_amp_and_phase(amp, phs, wvl, dx)
        if Q is not None:
            fus = wf._fusable(Q)      # float maps, power-of-two width: the pupil is synthesised inside the transform
            src, syn = (fus[1], (fus[0], fus[2])) if fus is not None else (wf.data, None)
            if acc is None:
                first = focus_intensity(src, Q, synth=syn)
                return first * w if w != 1.0 else first
            return focus_intensity(src, Q, out=acc, weight=w, synth=syn)
        ex = wf.prepare_executor(efl, focal_dx, samples, kind=kind)
        E = wf.focus_dft(ex).data
        if acc is None:
            acc = torch.zeros(E.shape, dtype=L._REAL_OF[E.dtype], device=E.device)
        return _ops.abs2(E, out=acc, weight=w)

    return incoherent_sum(propagate, wavelengths, weights, group=group, reduce_to_all=reduce_to_all, reduce_method=reduce_method)
