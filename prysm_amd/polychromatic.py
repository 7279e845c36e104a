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
import threading
import weakref

import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world_size):
    """Contiguous block [lo, hi) of `n_items` owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


# (amplitude, OPD) -> packed map, keyed on tensor IDENTITY and torch's version counters: a model that calls polychromatic_psf
# repeatedly with the same two maps (a wavelength sweep per frame, an optimiser's forward pass with unchanged amplitude and OPD
# tensors) packs them once instead of paying two extra sweeps of the pupil inside every call (6 % of an 8-wavelength share at 4096^2).
# Identity, not address: a freed map's address is reused by the next one.  In-place torch ops bump the version; the library's own
# `out=` writes do too (_ops._bump).
# OPT-IN (`cache_pupil=True`), because the key cannot see every write: `t.data.copy_()` / `t.data.add_()`, DLPack or cupy aliases,
# custom kernels and any raw-pointer write leave identity and version unchanged, and the cache would then hand back a STALE map.  The
# contract of cache_pupil=True: between calls the two maps change only through torch in-place ops or this library's `out=` arguments --
# or the caller calls clear_packed_pupil_cache().  Tensors without a version counter (torch.inference_mode) are never cached.
_PACK_CACHE = []
_PACK_CACHE_MAX = 4
_PACK_LOCK = threading.Lock()


def clear_packed_pupil_cache():
    """Forget every cached packed (amplitude, OPD) map (after a write the version counters cannot see)."""
    with _PACK_LOCK:
        del _PACK_CACHE[:]


def _version(t):
    """torch's version counter of `t`; None for no tensor, -1 when it has none (inference tensors raise): never equal to a cached key."""
    if t is None:
        return None
    try:
        return t._version
    except RuntimeError:
        return -1


def packed_pupil(amp, opd, a_syn, o_syn, cache=False):
    """pack_amp_opd(a_syn, o_syn); with `cache`, kept per identity + version of the caller's tensors `amp` / `opd` (see above)."""
    from . import _ops
    va, vo = _version(amp), _version(opd)
    if not cache or va == -1 or vo == -1:
        return _ops.pack_amp_opd(a_syn, o_syn)
    # ... and per STREAM: the map is packed by work queued on the current stream, and a reader on another stream (StreamRing, one
    # pipeline per thread) would not be ordered behind it
    from . import _lib as L
    st = L._cur_stream() if opd.is_cuda else 0
    with _PACK_LOCK:
        for i, (ra, ca, ro, co, ps, packed) in enumerate(_PACK_CACHE):
            if (ra() if ra is not None else None) is amp and ro() is opd and ca == va and co == vo and ps == st:
                if i:
                    _PACK_CACHE.insert(0, _PACK_CACHE.pop(i))
                return packed
    packed = _ops.pack_amp_opd(a_syn, o_syn)
    with _PACK_LOCK:
        _PACK_CACHE.insert(0, (None if amp is None else weakref.ref(amp), _version(amp), weakref.ref(opd), _version(opd), st, packed))
        del _PACK_CACHE[_PACK_CACHE_MAX:]
        _PACK_CACHE[:] = [e for e in _PACK_CACHE if (e[0] is None or e[0]() is not None) and e[2]() is not None]
    return packed


def _group_info(group):
    """(process group in use?, rank in group, world).  A group of ONE rank still runs its collective: the call is then the same
    code path at every N (and the world-1 RCCL test on a one-GPU box executes exactly what the 8-GPU node executes)."""
    use = dist.is_available() and dist.is_initialized()
    return use, (dist.get_rank(group) if use else 0), (dist.get_world_size(group) if use else 1)


_SCRATCH = {}
_SCRATCH_LOCK = threading.Lock()


def _scratch(tag, shape, dtype, device):
    """A receive buffer kept between calls, per (use, shape, dtype, device, stream): the reduce sits on the critical path of every image, and
    a fresh 67 MB allocation per call (plus a torch.cat of as much at the root) was an extra sweep of the image inside it.
    Stream-ordered like _lib.workspace: one image's collective and the next one's are queued on the same stream."""
    st = 0
    if device.type == 'cuda':
        from . import _lib as L
        st = L._cur_stream()
    key = (tag, tuple(shape), dtype, device, st)
    with _SCRATCH_LOCK:
        t = _SCRATCH.pop(key, None)
        if t is None:
            while len(_SCRATCH) >= 8:           # least recently used first (dicts keep insertion order; a hit is re-inserted below)
                _SCRATCH.pop(next(iter(_SCRATCH)))
            t = torch.empty(shape, dtype=dtype, device=device)
        _SCRATCH[key] = t
    return t


def clear_scratch():
    """Release the persistent receive buffers of the reduce forms (up to eight image-sized tensors, keyed by shape / dtype / device /
    stream).  Call after destroying streams the driver ran on: a raw stream handle can be handed out again."""
    with _SCRATCH_LOCK:
        _SCRATCH.clear()


REDUCE_METHODS = ('reduce', 'a2a', 'rs')


def _reduce_image(acc, world, group, reduce_to_all, method='reduce', use_dist=None):
    """The one data-path collective: sum-reduce of the real image over the ranks (RCCL; gloo in the CPU tests).

    reduce_to_all           : all_reduce, every rank gets the image.
    root only, 'reduce'     : torch.distributed.reduce to the first rank of the group; the buffers of the other ranks are
                              scratch afterwards (gloo and RCCL both leave partial sums in them).
    root only, 'a2a'        : the fully connected xGMI form of SURVEY 8(e): every rank sends slice j of its image to rank j
                              (one all-to-all, 7 distinct links per GPU), sums the `world` slices it received in rank order
                              (pm_sum_modes: fixed order, bitwise reproducible whatever the transport does) and the first rank
                              gathers the reduced slices STRAIGHT INTO the slices of its image (no list of parts, no concatenation).
    root only, 'rs'         : the same exchange as ONE library collective, reduce_scatter_tensor (RCCL picks ring or direct by
                              size), then the same gather.  The order of the additions is the library's: reproducible on a fixed
                              topology, not bit-equal to 'a2a'.
    'a2a' and 'rs' need numel % world == 0 and a contiguous image; they fall back to 'reduce' otherwise.
    """
    if method not in REDUCE_METHODS:
        raise ValueError(f'reduce_method must be one of {REDUCE_METHODS}, got {method!r}')
    if use_dist is None:
        use_dist = world > 1
    if not use_dist:
        return acc
    if reduce_to_all:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        return acc
    root = dist.get_global_rank(group, 0) if group is not None else 0
    if method in ('a2a', 'rs') and acc.numel() % world == 0 and acc.is_contiguous():
        per = acc.numel() // world
        send = acc.view(world, per)
        if method == 'rs':
            part = _scratch('rs', (per,), acc.dtype, acc.device)
            dist.reduce_scatter_tensor(part, acc.view(-1), op=dist.ReduceOp.SUM, group=group)
        else:
            recv = _scratch('a2a', (world, per), acc.dtype, acc.device)
            dist.all_to_all_single(recv, send, group=group)
            if acc.is_cuda:
                from . import _ops
                part = _ops.sum_modes(recv.view(world, 1, per), [1.0] * world, out=_scratch('a2a_part', (1, per), acc.dtype, acc.device)).view(per)
            else:
                part = _scratch('a2a_part', (per,), acc.dtype, acc.device)
                torch.sum(recv, 0, out=part)       # rank order, one pass
        me = dist.get_rank()
        # the root receives slice j of the reduced image from rank j directly in acc[j * per : (j + 1) * per] (its own send data there is
        # spent: the exchange above has completed in stream order)
        dist.gather(part, [send[j] for j in range(world)] if me == root else None, dst=root, group=group)
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
    use_dist, rank, world = _group_info(group)
    lo, hi = shard_bounds(len(wavelengths), rank, world)
    acc = _loop_sum(propagate, wavelengths, weights, lo, hi, out)
    return _reduce_image(acc, world, group, reduce_to_all, reduce_method, use_dist)


def _loop_sum(propagate, wavelengths, weights, lo, hi, out=None):
    acc = out
    for k in range(lo, hi):
        acc = propagate(float(wavelengths[k]), float(weights[k]), acc)
    if acc is None:
        raise ValueError('a rank received no wavelengths and no `out` buffer to define the image shape')
    return acc


def _fields_per_launch(shape, cdtype_bytes, Q, limit_bytes=1 << 30):
    """How many wavelengths go into one batched launch: bounded by ~1 GiB of stack (pupils + intensities)."""
    m, n = shape
    M, N = math.ceil(m * Q), math.ceil(n * Q)
    per_field = m * n * cdtype_bytes + M * N * cdtype_bytes // 2
    return max(1, min(64, limit_bytes // per_field))


def _local_sum(amplitude, opd, wavelengths, weights, lo, hi, dx, efl, Q, focal_dx, samples, kind, batched, spectral, cache_pupil=False):
    """sum_k w_k |E_k|^2 over this rank's wavelengths [lo, hi): the compute half of polychromatic_psf (no collective)."""
    from . import _lib as L
    from . import _ops
    from .propagation import Wavefront, focus_intensity
    from .propagation.wavefront import _synth_args

    if len(wavelengths) != len(weights):
        raise ValueError('wavelengths and weights must have the same length')
    amp = L.as_device(amplitude)
    phs = L.as_device(opd)

    packed = None
    if Q is not None and batched is None and len(wavelengths) > 1:
        # small composite grids whose stacks run as ONE launch pair on the composite register engine (round 5): 16 wavelengths of a 500^2
        # grid 160 us as stacks against 202 us for per-wavelength launch pairs, 1000^2 220 / 281, 1536^2 a tie
        # (profiles/r05/exp_poly_composite.log)
        MN = (math.ceil(amp.shape[-2] * Q), math.ceil(amp.shape[-1] * Q))
        if MN[0] * MN[1] <= 1400 * 1400 and phs.dtype in (torch.float32, torch.float64) and \
                _ops.on_register_engine(MN[0], MN[1], torch.complex128 if phs.dtype == torch.float64 else torch.complex64):
            batched = True
    if Q is not None and not batched:
        # the pupil is synthesised inside every wavelength's transform (float maps, power-of-two width): pack (amplitude, OPD)
        # once -- per pair of maps, not per call (packed_pupil) -- so each of those row passes reads one 8-byte element per sample
        # instead of two 4-byte ones from two arrays
        probe = Wavefront.from_amp_and_phase(amp, phs, float(wavelengths[0]), dx)._fusable(Q) if len(wavelengths) else None
        if probe is not None and len(wavelengths) > 1:
            packed = packed_pupil(amp, phs, probe[0], probe[1], cache_pupil)
    if Q is not None and batched is None:
        batched = (packed is None or not spectral) and math.ceil(amp.shape[-2] * Q) * math.ceil(amp.shape[-1] * Q) < 4096 * 4096
    if Q is not None and batched:
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
        return acc

    if packed is not None and spectral:
        # the rank's whole wavelength loop in one call: groups of wavelengths share a launch pair (the packed map is read once per
        # group, the image touched once per group -- csrc/fft_spectral.h)
        m, n = packed.shape
        acc = torch.zeros((math.ceil(m * Q), math.ceil(n * Q)), dtype=L._REAL_OF[packed.dtype], device=packed.device)
        if hi > lo:
            ks = [2 * math.pi / float(wavelengths[k]) / 1e3 for k in range(lo, hi)]
            focus_intensity(packed, Q, out=acc, synth=('packed', ks[0]), spectral=(ks, [float(w) for w in weights[lo:hi]]))
        return acc

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
        # matrix DFT: |.|^2 and the weighted accumulate in the second product's epilogue; chirp-Z / FFT-DFT: composed
        return wf.focus_dft_intensity(ex, out=acc, weight=w).data

    return _loop_sum(propagate, wavelengths, weights, lo, hi)


def polychromatic_psf(amplitude, opd, wavelengths, weights, dx, efl, *, Q=None, focal_dx=None, samples=None,
                      kind='mdft', group=None, reduce_to_all=True, batched=None, reduce_method='reduce', spectral=True, cache_pupil=False):
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

    cache_pupil        : keep the packed (amplitude, OPD) map of this pair of tensors for the next call (keyed on identity + torch's
                         version counters; two sweeps of the pupil saved per call).  Off by default: a write the counters cannot see
                         (`.data` ops, DLPack / cupy aliases, raw-pointer kernels) would be answered with a stale map -- see
                         packed_pupil / clear_packed_pupil_cache.

    A process group of one rank still runs its collective (same code path at every N); without a process group there is none.
    """
    use_dist, rank, world = _group_info(group)
    lo, hi = shard_bounds(len(wavelengths), rank, world)
    acc = _local_sum(amplitude, opd, wavelengths, weights, lo, hi, dx, efl, Q, focal_dx, samples, kind, batched, spectral, cache_pupil)
    return _reduce_image(acc, world, group, reduce_to_all, reduce_method, use_dist)


class PendingImage:
    """A polychromatic PSF whose image reduce may still be running on the pipeline's side stream."""

    def __init__(self, image, event, keep):
        self._image, self._event, self._keep = image, event, keep

    def result(self):
        """The image (first rank of the group, or every rank with reduce_to_all); torch's CURRENT stream waits for the reduce."""
        if self._event is not None:
            torch.cuda.current_stream().wait_event(self._event)
            self._event = None
        self._keep = None
        return self._image

    def done(self):
        return self._event is None or self._event.query()


class PsfPipeline:
    """A SEQUENCE of polychromatic PSFs (frames of a time series, the forward passes of an optimiser, field points ...) with the
    one collective of frame k running on a side stream while frame k + 1 computes.

    A single polychromatic_psf call ends with its image reduce exposed: at 8 GPUs, 8 wavelengths x ~105 us of transforms per rank
    stand against a 67 MB fp32 reduce that takes a comparable time over xGMI, so one call alone cannot scale near-linearly.  The
    frames of a sequence are independent, and the reduce needs no compute units to speak of, so it hides behind the next frame's
    transforms: ``submit`` enqueues a frame's wavelength loop on the current stream, then hands its image to the side stream
    (which waits for the loop, runs `_reduce_image` there -- RCCL orders itself behind the stream it is called on -- and records
    an event); ``PendingImage.result()`` makes the consumer's stream wait for that event.  Every rank submits the same frames in
    the same order, so the collectives match.  At most `depth` frames are in flight (their accumulators are live).
    CPU tensors (the gloo tests) run the same calls synchronously.
    """

    def __init__(self, wavelengths, weights, dx, efl, *, Q=None, focal_dx=None, samples=None, kind='mdft', group=None,
                 reduce_to_all=False, batched=None, reduce_method='reduce', spectral=True, depth=2, propagate=None, cache_pupil=False):
        """propagate(amplitude, opd, wavelength, weight, acc) -> acc: optional replacement of the per-wavelength step (as in
        incoherent_sum; the CPU tests inject the oracle here).  cache_pupil: as in polychromatic_psf (frames that submit the SAME
        two tensors pack them once)."""
        if len(wavelengths) != len(weights):
            raise ValueError('wavelengths and weights must have the same length')
        self.wavelengths, self.weights, self.dx, self.efl = list(wavelengths), list(weights), dx, efl
        self.kw = dict(Q=Q, focal_dx=focal_dx, samples=samples, kind=kind, batched=batched, spectral=spectral, cache_pupil=cache_pupil)
        self.group, self.reduce_to_all, self.reduce_method = group, reduce_to_all, reduce_method
        self.depth = max(1, int(depth))
        self._propagate = propagate
        self._side = None
        self._inflight = []

    def submit(self, amplitude, opd):
        use_dist, rank, world = _group_info(self.group)
        lo, hi = shard_bounds(len(self.wavelengths), rank, world)
        while len(self._inflight) >= self.depth:      # bound the live accumulators: wait (host side) for the oldest reduce
            ev = self._inflight.pop(0)
            ev.synchronize()
        if self._propagate is not None:
            acc = _loop_sum(lambda wvl, w, a: self._propagate(amplitude, opd, wvl, w, a), self.wavelengths, self.weights, lo, hi)
        else:
            acc = _local_sum(amplitude, opd, self.wavelengths, self.weights, lo, hi, self.dx, self.efl, **self.kw)
        if not acc.is_cuda or not use_dist:
            return PendingImage(_reduce_image(acc, world, self.group, self.reduce_to_all, self.reduce_method, use_dist), None, None)
        if self._side is None:
            self._side = torch.cuda.Stream(device=acc.device)
        main = torch.cuda.current_stream(acc.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            img = _reduce_image(acc, world, self.group, self.reduce_to_all, self.reduce_method, use_dist)
            ev = torch.cuda.Event()
            ev.record(self._side)
        acc.record_stream(self._side)      # the accumulator came from the main stream's allocator pool
        self._inflight.append(ev)
        return PendingImage(img, ev, acc)

    def drain(self):
        """Block the host until every submitted frame's reduce has finished."""
        for ev in self._inflight:
            ev.synchronize()
        self._inflight = []
