"""hipGraph capture of a propagation chain.

The library only enqueues kernels on the caller's stream: no synchronisation, no hidden allocation (twiddle tables
are built at first use of a size, workspaces are cached per stream).  A whole model -- pupil synthesis, several
propagations, intensity -- can therefore be captured once into a hipGraph (through ``torch.cuda.CUDAGraph``, which is
hipGraph on ROCm) and replayed with one launch: the 21 us of Python + ctypes issue cost per call, which dominates
fields of 1024^2 and below, disappears.  This is the MI355X answer to the reference's advice of fusing work to
amortise launch overhead (docs: GPU and Exascale Computing.ipynb).
"""
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


class StreamRing:
    """Round-robin HIP streams for a SEQUENCE of independent propagations (the fields of a stack that does not fit one launch, the
    wavelengths or field points of a model, one pipeline per thread as the reference advises -- GPU and Exascale Computing.ipynb).

    One propagation is two dependent launches; at 2048^2 and below each launch is a single round of workgroups, so its tail (the last
    workgroups storing) and the next launch's head (the first loads) leave most of the chip idle: 30.6 us per 2048^2 complex64
    `focus` back to back on one stream, 26.3 us when consecutive calls alternate between two streams (tools/exp_two_streams.py,
    profiles/r04/exp_two_streams.log).  Only while BOTH fields' arrays fit the 256 MiB Infinity Cache together: at 4096^2 two
    streams evict each other's intermediates (95 -> 133 us) -- `worth_it(shape, dtype)` says which side of that a shape is on.

        ring = StreamRing(2)
        ring.fork()                                           # every stream of the ring waits for what the caller's stream has queued
        outs = [ring.run(P.focus, x, 1) for x in fields]     # each call on the next stream
        ring.join()                                           # the caller's stream now waits for every stream of the ring

    fork / join are the only synchronisation (one event each way per stream): a wait per call would cost more host time than a 2048^2
    propagation takes on the device.  Inputs must come from before the fork (or from the same stream of the ring).

    Workspaces are per stream (_lib.workspace), outputs come from torch's stream-aware allocator; results are the same bits.
    """

    def __init__(self, n=2, device=None):
        L.load()
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, int(n)))]
        self._next = 0

    @staticmethod
    def worth_it(shape, dtype=torch.complex64, streams=2):
        """True when `streams` concurrent propagations of this shape keep input + intermediate + output inside the Infinity Cache."""
        m, n = shape[-2:]
        return 3 * m * n * torch.empty((), dtype=dtype).element_size() * streams <= 200 << 20

    def fork(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(cur)

    def run(self, fn, *args, **kwargs):
        s = self.streams[self._next]
        self._next = (self._next + 1) % len(self.streams)
        with torch.cuda.stream(s):
            return fn(*args, **kwargs)

    def join(self):
        cur = torch.cuda.current_stream()
        for s in self.streams:
            cur.wait_stream(s)
