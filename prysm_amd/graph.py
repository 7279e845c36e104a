"""hipGraph capture of a propagation chain.

The library only enqueues kernels on the caller's stream: no synchronisation, no hidden allocation (twiddle tables
are built at first use of a size, workspaces are cached per stream).  A whole model -- pupil synthesis, several
propagations, intensity -- can therefore be captured once into a hipGraph (through ``torch.cuda.CUDAGraph``, which is
hipGraph on ROCm) and replayed with one launch: the 21 us of Python + ctypes issue cost per call, which dominates
fields of 1024^2 and below, disappears.  This is the MI355X answer to the reference's advice of fusing work to
amortise launch overhead (docs: GPU and Exascale Computing.ipynb).
"""
import functools
import threading
import weakref

import torch

from . import _lib as L


class CapturedModel:
    """fn(*tensors) -> tensor | tuple | object, captured into a hipGraph.

    model = CapturedModel(fn, example_a, example_b);  out = model(a, b)
    Inputs are copied into the graph's static buffers (same shapes / dtypes as the examples); the returned object is
    the graph's static output, overwritten by the next call -- clone what must be kept.  `fn` must not read device
    values on the host (no .item(), no data-dependent Python branches).
    """

    def __init__(self, fn, *example_inputs, warmup=2):
        L.load()
        self._static_in = [L.as_device(x).clone() for x in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):     # warm-up off the default stream: plan cache, workspaces, allocator pools
            for _ in range(max(1, warmup)):
                fn(*self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._static_out = fn(*self._static_in)

    def __call__(self, *inputs):
        if len(inputs) != len(self._static_in):
            raise ValueError(f'expected {len(self._static_in)} inputs, got {len(inputs)}')
        for dst, src in zip(self._static_in, inputs):
            src = L.as_device(src)
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError(f'input of shape {tuple(src.shape)} / {src.dtype} does not match the captured '
                                 f'{tuple(dst.shape)} / {dst.dtype}')
            dst.copy_(src)
        self.graph.replay()
        return self._static_out


def capture(fn, *example_inputs, warmup=2):
    """Capture fn(*example_inputs) into a hipGraph; returns the replayable model."""
    return CapturedModel(fn, *example_inputs, warmup=warmup)


def _tensors_of(obj, out):
    """device tensors reachable from a result: a tensor, a sequence / dict of results, an object that holds its array in `data`
    (Wavefront -- not materialising a lazy one --, RichData)"""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _tensors_of(o, out)
    elif isinstance(obj, dict):
        for o in obj.values():
            _tensors_of(o, out)
    elif obj is not None and not isinstance(obj, (int, float, complex, str, bytes)):
        d = getattr(obj, '__dict__', None)
        if d:
            t = d.get('_data', d.get('data'))
            if isinstance(t, torch.Tensor):
                if t.is_cuda:
                    out.append(t)
            elif t is None and d.get('_synth') is not None:
                # a lazy Wavefront (from_amp_and_phase: `_data` None, the maps held in `_synth`): whoever reads it reads the maps
                # (an OPD summed inside the block on one ring stream must order the synthesis that reads it)
                for o in d['_synth']:
                    if isinstance(o, torch.Tensor) and o.is_cuda:
                        out.append(o)
    return out


def _holders_of(obj, out):
    """argument objects that may materialise an array DURING a call (lazy Wavefronts: `_data` None, `_synth` set)"""
    if isinstance(obj, (list, tuple)):
        for o in obj:
            _holders_of(o, out)
    elif obj is not None and not isinstance(obj, (torch.Tensor, int, float, complex, str, bytes, dict)):
        d = getattr(obj, '__dict__', None)
        if d and d.get('_data', 0) is None and d.get('_synth') is not None:
            out.append(obj)
    return out


class StreamRing:
    """Round-robin HIP streams for a SEQUENCE of independent propagations (the fields of a stack that does not fit one launch, the
    wavelengths or field points of a model, one pipeline per thread as the reference advises -- GPU and Exascale Computing.ipynb).

    One propagation is two dependent launches; at 2048^2 and below each launch is a single round of workgroups, so its tail (the last
    workgroups storing) and the next launch's head (the first loads) leave most of the chip idle: 30.6 us per 2048^2 complex64
    `focus` back to back on one stream, 26.3 us when consecutive calls alternate between two streams (experiments/scripts/exp_two_streams.py,
    profiles/r04/exp_two_streams.log).  Only while BOTH fields' arrays fit the 256 MiB Infinity Cache together: at 4096^2 two
    streams evict each other's intermediates (95 -> 133 us) -- `worth_it(shape, dtype)` says which side of that a shape is on.

        ring = StreamRing(2)
        outs = [ring.run(P.focus, x, 1) for x in fields]     # each call on the next stream (the first run of a batch forks)
        ring.join()                                           # the caller's stream now waits for every stream of the ring

    Synchronisation is one event each way per stream and per BATCH, not per call (a wait per call would cost more host time than a
    2048^2 propagation takes on the device):
      * fork() orders every stream of the ring behind what the caller's stream has queued; the first run() after a join() (or after
        construction) forks by itself, so "loop {run ...; join; consume}" is ordered in both directions on every trip.  Inputs made
        on the caller's stream AFTER that fork need another fork() (or must come from the same stream of the ring).
      * join() orders the caller's stream behind every stream of the ring and notes every result of the batch that is STILL ALIVE as
        "also used on the caller's stream" (Tensor.record_stream): it was allocated on a ring stream, and without that note torch's
        caching allocator would hand its block to a later run() on that stream as soon as the caller dropped the tensor -- while
        the caller's stream could still have reads of it queued.  A result dropped BEFORE the join was never read by
        the caller; one dropped after it is covered by the note; and the next batch forks before it reuses anything.  (Per result
        this costs a weak reference in run() and one record_stream at the join -- not a call on the critical path of every run().)

    Workspaces are per stream (_lib.workspace), outputs come from torch's stream-aware allocator; results are the same bits.
    """

    def __init__(self, n=2, device=None):
        L.load()
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, int(n)))]
        self._next = 0
        self._forked = False      # a batch is open: the ring's streams are ordered behind the caller's stream
        self._made = []           # weak references to the results of the open batch

    @staticmethod
    def worth_it(shape, dtype=torch.complex64, streams=2):
        """True when `streams` concurrent propagations of this shape keep input + intermediate + output inside the Infinity Cache."""
        m, n = shape[-2:]
        return 3 * m * n * torch.empty((), dtype=dtype).element_size() * streams <= 200 << 20

    def fork(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(cur)
        self._forked = True

    def next_stream(self):
        s = self.streams[self._next]
        self._next = (self._next + 1) % len(self.streams)
        return s

    def run_on(self, s, fn, *args, **kwargs):
        """fn(*args, **kwargs) on stream `s` of the ring; its result tensors are recorded on the caller's stream"""
        if not self._forked:
            self.fork()
        with torch.cuda.stream(s):
            out = fn(*args, **kwargs)
        self.note(out)
        return out

    def note(self, out):
        made = self._made
        for t in _tensors_of(out, []):
            made.append(weakref.ref(t))

    def run(self, fn, *args, **kwargs):
        return self.run_on(self.next_stream(), fn, *args, **kwargs)

    def join(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)
        for r in self._made:
            t = r()
            if t is not None:
                t.record_stream(cur)
        self._made = []
        self._forked = False


_seq_state = threading.local()


def active_sequence():
    """the Sequence the calling thread is inside of, or None"""
    return getattr(_seq_state, 'seq', None)


def _key(t):
    """identity of the memory a tensor lives in: the data pointer of its base (views share their base's producer)"""
    b = t._base
    return (t if b is None else b).data_ptr()


class Sequence:
    """``with prysm_amd.graph.sequence(): ...`` around an UNMODIFIED loop of Wavefront / propagation calls: consecutive calls that do
    not depend on each other run on alternating HIP streams (a StreamRing), calls that do depend stay on their producer's stream.

        with graph.sequence() as seq:
            psfs = [P.Wavefront.from_amp_and_phase(amp, opd, wvl, dx).focus(efl, Q=2).intensity for opd in opds]
        # here the caller's stream is ordered behind all of it

    What prysm users write for small fields (prysm/propagation/wavefront.py:478-504 in a loop over wavelengths or field points;
    prysm/x/polarization.py:478-553 is the reference's own batch precedent) gets the two-stream overlap of StreamRing without
    being restructured into stacks or ring.run(...) calls (DESIGN.md 3.4 has the measurements; below ~1500^2 the host's ~18 us per
    call is the limit either way and stacks / graph.capture remain the tools).

    How: every array-level entry point of the library (prysm_amd._ops, and Wavefront arithmetic) asks `dispatch` for its stream.  A
    call whose tensor arguments were all made before the block (or on the host) takes the next stream of the ring; a call that
    reads a tensor produced inside the block runs on the stream that produced it (same-stream order IS the dependence), waiting for
    any other producers it reads from.  Leaving the block joins.  The results are bit-identical to the one-stream run: same
    kernels, per-stream workspaces.

    Inputs made by plain torch operations on the caller's stream inside the block (amp.to(dtype), a mask built in the loop) are
    waited for when the caller's stream is not idle the first time the block sees them.  The one rule: RESULTS must not be read by
    plain torch operations or copied to the host before the block ends (or seq.join()) -- Wavefront arithmetic, .intensity /
    .phase and prysm_amd's own host conversions (array_to_true_numpy, Wavefront.__array__) are part of the library and do the
    right thing.

    The dispatch is on the host's critical path (a 2048^2 propagation is 26 us of device time): streams are switched through the raw
    setter, tensors are identified by data pointer, and an input is checked against the caller's stream once, not per call.
    """

    def __init__(self, streams=2, device=None):
        self.ring = StreamRing(streams, device)
        self.cache_budget = 200 << 20      # bytes of the 256 MiB Infinity Cache the concurrent propagations may claim
        self._producer = {}       # data pointer of a tensor made inside the block -> its stream
        # outside inputs already ordered behind the caller's stream: data pointer -> (the tensor, its version counter).  The tensor is
        # HELD until the next fork / join: a per-iteration temporary (amp.to(dtype), a mask built in the loop) that was dropped could
        # otherwise hand its address to the next iteration's temporary, which would then count as ordered without being so;
        # the version counter catches a buffer the caller rewrites in place between two calls.
        self._checked = {}
        self._depth = 0
        self._caller = None
        self._capturing = False
        self._set = getattr(torch._C, '_cuda_setStream', None)
        self._ids = {}            # id(stream) -> the three integers of the raw setter

    def _switch(self, s):
        if self._set is None:
            torch.cuda.set_stream(s)
            return
        ids = self._ids.get(id(s))
        if ids is None:
            ids = self._ids[id(s)] = (s.stream_id, s.device_index, s.device_type)
        self._set(stream_id=ids[0], device_index=ids[1], device_type=ids[2])

    def __enter__(self):
        if active_sequence() is not None:
            raise RuntimeError('prysm_amd.graph.sequence() blocks do not nest')
        self._caller = torch.cuda.current_stream()
        # inside a hipGraph capture (graph.capture of a function that opens a sequence block: the captured graph then has one branch
        # per ring stream, and independent chains -- the wavelengths of a model -- overlap on the device at no host cost) a stream
        # must not be queried: every outside input counts as "the caller's stream is busy"
        self._capturing = torch.cuda.is_current_stream_capturing()
        _seq_state.seq = self
        self.ring.fork()
        return self

    def __exit__(self, *exc):
        _seq_state.seq = None
        self.join()
        return False

    def fork(self):
        self.ring.fork()
        self._checked.clear()

    def join(self):
        self.ring.join()
        self._producer.clear()
        self._checked.clear()

    def dispatch(self, fn, args, kwargs):
        """run fn(*args, **kwargs) on the stream the data dependences pick; nested library calls run inside the outer call's stream"""
        if self._depth:
            return fn(*args, **kwargs)
        ins = _tensors_of(args, [])
        if kwargs:
            _tensors_of(kwargs, ins)
        prod, checked = self._producer, self._checked
        s = None
        others = None
        fresh = None
        for t in ins:
            k = _key(t)
            ps = prod.get(k)
            if ps is None:
                seen = checked.get(k)
                if seen is None or seen[1] != t._version:
                    fresh = (fresh or []) + [(k, t)]
                    if seen is None:
                        for r in self.ring.streams:      # allocated elsewhere, read on the ring: its block must not be recycled under those reads
                            t.record_stream(r)
            elif ps is not s:
                if s is None:
                    s = ps
                else:
                    others = (others or []) + [ps]
        if s is None:
            # an independent call: the next stream of the ring -- unless its arrays are too large for two propagations to share the
            # Infinity Cache (StreamRing.worth_it: input + intermediate + output per stream), where two streams evict each other's
            # intermediates (2048^2 complex128: 47 -> 54 us per call; 4096^2 complex64: 95 -> 133) and everything stays on stream 0
            big = 0
            for t in ins:
                nb = t.numel() * t.element_size()
                big = nb if nb > big else big
            s = self.ring.next_stream() if 3 * big * len(self.ring.streams) <= self.cache_budget else self.ring.streams[0]
        if not self.ring._forked:
            self.ring.fork()
        if others:
            for o in others:
                s.wait_stream(o)
            for t in ins:                   # made on another ring stream, read on this one: its block must outlive these reads
                if prod.get(_key(t)) not in (None, s):
                    t.record_stream(s)
        if fresh:
            # inputs that were not made inside the block: from before it (ordered by the fork) or from a plain torch operation on the
            # caller's stream since (amp.to(dtype), a mask built in the loop).  If the caller's stream is not idle, EVERY ring stream waits
            # for it once; after that THIS tensor at THIS version counts as ordered.
            if self._capturing or not self._caller.query():
                for r in self.ring.streams:
                    r.wait_stream(self._caller)
            if len(checked) > 256:              # a long block of temporaries: let the old ones go (they carry record_stream notes)
                checked.clear()
            for k, t in fresh:
                checked[k] = (t if t._base is None else t._base, t._version)
        lazy = _holders_of(args, [])
        before = {_key(t): t._version for t in ins} if ins else {}
        self._depth = 1
        self._switch(s)
        try:
            out = fn(*args, **kwargs)
        finally:
            self._switch(self._caller)
            self._depth = 0
        made = self.ring._made
        if len(made) > 4096:                # a long block: forget the results that are gone already
            made[:] = [r for r in made if r() is not None]
        news = _tensors_of(out, [])
        for h in lazy:                      # a lazy Wavefront among the arguments that materialised its array inside this call: made on s
            t = h.__dict__.get('_data')
            if isinstance(t, torch.Tensor) and t.is_cuda:
                news.append(t)
        for t in news:
            k = _key(t)
            if k not in prod and before.get(k) == t._version:
                continue                    # an outside input handed through UNCHANGED (the amplitude map of a lazy product): still an outside
                                            # input, not something this stream made -- or every later chain that reads it would follow this
                                            # stream.  (An `out=` accumulator is written: its version counter moved, it is registered.)
            made.append(weakref.ref(t))     # recorded on the caller's stream at the join if still alive (StreamRing.join)
            prod[k] = s
        return out


def sequence(streams=2, device=None):
    """Context manager: independent propagations inside the block alternate between `streams` HIP streams (see Sequence)."""
    return Sequence(streams, device)


def sequenced(fn):
    """decorator of the library's array-level entry points: inside a sequence() block the call goes through Sequence.dispatch"""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        seq = getattr(_seq_state, 'seq', None)
        if seq is None:
            return fn(*args, **kwargs)
        return seq.dispatch(fn, args, kwargs)
    return wrapper
