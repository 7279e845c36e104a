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
