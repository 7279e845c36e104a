"""ctypes binding of libprysm_amd.so (the C ABI in include/prysm_amd.h).

There is no CPU fallback: if the shared library is missing, or no MI355X is
visible, every compute call raises.  PyTorch is used only as the owner of
device memory and streams; all arithmetic on the hot path happens in the HIP
kernels behind this binding.
"""
import ctypes
import threading
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PRYSM_AMD_LIB: another BUILD of the same library (tools/exp_ab_libs.py: two builds -- e.g. two commits -- against each other on one box)
LIB_PATH = os.environ.get('PRYSM_AMD_LIB') or os.path.join(_HERE, 'libprysm_amd.so')

PM_C64, PM_C128, PM_F32, PM_F64, PM_BOOL = 0, 1, 2, 3, 4
PM_EPI_NONE, PM_EPI_ABS2, PM_EPI_ABS2_ACCUM, PM_EPI_ABS, PM_EPI_ARG = 0, 1, 2, 3, 4
PM_MUL_NONE, PM_MUL_FULL, PM_MUL_SEPARABLE = 0, 1, 2
PM_FLAG_PASS1_ONLY, PM_FLAG_PASS2_ONLY, PM_FLAG_REAL_INPUT, PM_FLAG_SYNTH_INPUT, PM_FLAG_NORM_DC, PM_FLAG_SYNTH_PACKED = 1, 2, 4, 8, 16, 32
PM_FLAG_REAL_OUTPUT = 64
PM_ERR_ARG, PM_ERR_UNSUPPORTED, PM_ERR_WORKSPACE = -1, -2, -3

c_i32, c_i64, c_f64, c_vp, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t


class pm_axis(ctypes.Structure):
    _fields_ = [('n', c_i64), ('len', c_i64), ('off', c_i64), ('shift', c_i64)]


class pm_fft2_desc(ctypes.Structure):
    _fields_ = [
        ('dtype', c_i32), ('direction', c_i32), ('epilogue', c_i32), ('flags', c_i32),
        ('scale', c_f64), ('weight', c_f64),
        ('in_y', pm_axis), ('in_x', pm_axis), ('out_y', pm_axis), ('out_x', pm_axis),
        ('in_ld', c_i64), ('out_ld', c_i64),
        ('mul_kind', c_i32), ('mul_conj', c_i32), ('mul', c_vp), ('mul_x', c_vp), ('mul_ld', c_i64),
        ('batch', c_i64), ('in_bstride', c_i64), ('out_bstride', c_i64), ('mul_bstride', c_i64), ('mul_x_bstride', c_i64),
        ('synth_amp', c_vp), ('synth_amp_dtype', c_i32), ('synth_reserved', c_i32), ('synth_amp_ld', c_i64), ('synth_k', c_f64),
    ]


# name -> (restype, argtypes); tests/test_capi_symbols.py checks this table against include/prysm_amd.h
SIGNATURES = {
    'pm_version': (c_i32, []),
    'pm_last_error': (ctypes.c_char_p, []),
    'pm_plan_prepare': (c_i32, [c_i32, c_i64]),
    'pm_shutdown': (None, []),
    'pm_set_tuning': (c_i32, [ctypes.c_char_p, c_i32]),
    'pm_set_tuning_local': (c_i32, [ctypes.c_char_p, c_i32]),
    'pm_reset_tuning_local': (None, []),
    'pm_plan_explain': (c_i32, [ctypes.POINTER(pm_fft2_desc), c_i32, ctypes.c_char_p, c_sz]),
    'pm_r2c_untangle': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_f64, c_vp, c_i64, c_vp]),
    'pm_fft2_workspace': (c_sz, [ctypes.POINTER(pm_fft2_desc)]),
    'pm_fft2': (c_i32, [ctypes.POINTER(pm_fft2_desc), c_vp, c_vp, c_vp, c_sz, c_vp]),
    'pm_fft2_spectral_workspace': (c_sz, [ctypes.POINTER(pm_fft2_desc), c_i32]),
    'pm_fft2_spectral': (c_i32, [ctypes.POINTER(pm_fft2_desc), c_i32, ctypes.POINTER(c_f64), ctypes.POINTER(c_f64), c_vp, c_vp, c_vp, c_sz,
                         c_vp]),
    'pm_fft2_mul_ifft2_workspace': (c_sz, [ctypes.POINTER(pm_fft2_desc)]),
    'pm_fft2_mul_ifft2': (c_i32, [ctypes.POINTER(pm_fft2_desc), c_vp, c_vp, c_vp, c_sz, c_vp]),
    'pm_fft2_time_passes': (c_i32, [ctypes.POINTER(pm_fft2_desc), c_vp, c_vp, c_vp, c_sz, c_i32,
                                    ctypes.POINTER(c_f64), c_vp]),
    'pm_fft1': (c_i32, [c_i32, c_i32, c_i32, c_i64, ctypes.POINTER(pm_axis), ctypes.POINTER(pm_axis), c_f64,
                        c_vp, c_i64, c_vp, c_i64, c_vp]),
    'pm_fft1_workspace': (c_sz, [c_i32, c_i32, c_i64, c_i64]),
    'pm_fft1_ws': (c_i32, [c_i32, c_i32, c_i32, c_i64, ctypes.POINTER(pm_axis), ctypes.POINTER(pm_axis), c_f64,
                           c_vp, c_i64, c_vp, c_i64, c_vp, c_sz, c_vp]),
    'pm_czt_vectors': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_f64, c_f64, c_vp, c_vp, c_vp, c_vp]),
    'pm_czt_axis': (c_i32, [c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_i32, c_vp, c_i32, c_vp, c_i32, c_f64,
                            c_vp, c_i64, c_vp, c_i64, c_vp]),
    'pm_fft1_ramp': (c_i32, [c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_i32, c_vp, c_i32, c_f64,
                             c_vp, c_i64, c_vp, c_i64, c_vp]),
    'pm_cmul': (c_i32, [c_i32, c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'pm_rmul': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_f64, c_vp, c_i64, c_vp]),
    'pm_scale_sep': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_i32, c_f64, c_vp, c_i64, c_vp]),
    'pm_abs2': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_f64, c_vp]),
    'pm_abs_arg': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'pm_sum_modes': (c_i32, [c_i32, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, ctypes.POINTER(c_f64), c_i32, c_vp, c_i64, c_vp]),
    'pm_encircled_energy_workspace': (c_sz, []),
    'pm_encircled_energy': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i64, c_f64, c_i64, ctypes.POINTER(c_f64), c_vp, c_vp, c_sz, c_vp]),
    'pm_encircled_energy_adjoint': (c_i32, [c_i32, c_i64, c_i64, c_f64, c_i64, ctypes.POINTER(c_f64), ctypes.POINTER(c_f64), c_vp,
                                            c_i64, c_vp]),
    'pm_sample_map': (c_i32, [c_i32, c_i32, c_i64, c_i64, c_vp, c_i64, c_f64, c_f64, c_f64, c_i64, c_i64, c_vp, c_i64, c_i64,
                              c_vp, c_i64, c_i64, c_vp, c_i64, c_f64, c_f64, c_vp, c_i64, c_vp]),
    'pm_spline_prefilter': (c_i32, [c_i32, c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    'pm_sample_spline': (c_i32, [c_i32, c_i32, c_i64, c_i64, c_vp, c_i64, c_f64, c_f64, c_f64, c_i64, c_i64, c_vp, c_i64, c_i64,
                                 c_vp, c_i64, c_i64, c_vp, c_i64, c_f64, c_f64, c_vp, c_i64, c_vp]),
    'pm_pupil_synth': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i32, c_i64, c_vp, c_i64, c_f64, c_vp, c_i64, c_vp]),
    'pm_quadratic_phase': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_f64, c_vp, c_i64, c_vp]),
    'pm_as_tf_vectors': (c_i32, [c_i32, c_i64, c_i64, c_f64, c_f64, c_f64, c_vp, c_vp, c_vp]),
    'pm_outer': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    'pm_embed': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp]),
    'pm_pad_index': (c_i32, [c_i32, c_i32, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp]),
    'pm_mdft_basis': (c_i32, [c_i32, c_i64, c_i64, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp]),
    'pm_mdft_basis_grid': (c_i32, [c_i32, c_i64, c_i64, c_f64, c_f64, c_f64, c_f64, c_i32, c_vp, c_i64, c_vp]),
    'pm_cgemm': (c_i32, [c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_f64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                         c_vp, c_sz, c_vp]),
    'pm_cgemm_workspace': (c_sz, [c_i32, c_i64, c_i64, c_i64]),
    'pm_cgemm_abs2': (c_i32, [c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_f64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_f64, c_i32,
                              c_vp, c_sz, c_vp]),
}

_lib = None


def load():
    """Load libprysm_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `make -C prysm_amd/csrc -j8` (or '
            '`python -c "import __graft_entry__ as g; g.build()"`).  prysm_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    other_build = bool(os.environ.get('PRYSM_AMD_LIB'))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)   # AttributeError here = the library does not export a declared symbol
        except AttributeError:
            if other_build:           # an OLDER build under test (tools/exp_ab_libs.py): it may predate an entry point; calling it raises
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class PrysmAmdError(RuntimeError):
    """A HIP runtime error reported by libprysm_amd.so."""


def check(rc):
    """Translate a C return code into the exception class the reference would raise."""
    if rc == 0:
        return
    lib = load()
    if rc < 0:
        msg = lib.pm_last_error().decode(errors='replace')
        if rc == PM_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise ValueError(msg)
    raise PrysmAmdError(f'libprysm_amd: HIP error {rc} (hipError_t); is an MI355X visible?')


# --------------------------------------------------------------------------- tensors

_COMPLEX_CODE = {torch.complex64: PM_C64, torch.complex128: PM_C128}
_REAL_OF = {torch.complex64: torch.float32, torch.complex128: torch.float64}
_COMPLEX_OF = {torch.float32: torch.complex64, torch.float64: torch.complex128, torch.float16: torch.complex64}
_NP2TORCH = {
    np.dtype('float32'): torch.float32, np.dtype('float64'): torch.float64, np.dtype('float16'): torch.float16,
    np.dtype('complex64'): torch.complex64, np.dtype('complex128'): torch.complex128, np.dtype('bool'): torch.bool,
    np.dtype('int32'): torch.int32, np.dtype('int64'): torch.int64, np.dtype('uint8'): torch.uint8,
}


_have_gpu = None
_devices = {}


def device():
    """The MI355X this process computes on (torch's current device)."""
    global _have_gpu
    if _have_gpu is None:
        _have_gpu = torch.cuda.is_available()       # asked once: the answer does not change within a process, the question costs 2 us
        if _have_gpu:
            torch.cuda.init()      # the raw accessors below assume torch's lazy runtime initialisation has happened
    if not _have_gpu:
        raise RuntimeError('prysm_amd needs an AMD MI355X (gfx950) visible to PyTorch-ROCm; there is no CPU path')
    i = _cur_dev()
    d = _devices.get(i)
    if d is None:
        d = _devices[i] = torch.device('cuda', i)
    return d


# torch.cuda.current_stream() / current_device() build Python objects and re-check the runtime on every call (~6 us of the ~21 us a
# small propagation costs on the host, experiments/scripts/exp_host_profile.py); the raw accessors return the same integers
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_raw_device = getattr(torch._C, '_cuda_getDevice', None)


def _cur_dev():
    return _raw_device() if _raw_device is not None else torch.cuda.current_device()


def _cur_stream():
    """hipStream_t of torch's current stream on the current device, as an integer."""
    if _raw_stream is not None:
        return _raw_stream(_cur_dev())
    return torch.cuda.current_stream().cuda_stream


def torch_dtype(dt):
    """numpy dtype-like or torch dtype -> torch dtype."""
    if isinstance(dt, torch.dtype):
        return dt
    return _NP2TORCH[np.dtype(dt)]


def as_device(x, dtype=None):
    """numpy array / torch tensor / scalar sequence -> contiguous tensor in HBM."""
    dev = device()
    if isinstance(x, torch.Tensor):
        t = x.to(dev) if x.device != dev else x
    else:
        a = np.ascontiguousarray(x)
        t = torch.from_numpy(a).to(dev)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.is_conj() or t.is_neg():       # torch's lazy conj / neg views: the library reads the stored bytes
        t = t.resolve_conj().resolve_neg()
    if not t.is_contiguous():
        t = t.contiguous()
    return t


def as_complex(x, at_least=None):
    """Promote like numpy does for fft input: f32 -> c64, everything else real -> c128; keep complex.

    `at_least` (a torch complex dtype) raises c64 to c128 when config.precision demands it.
    """
    t = as_device(x)
    if not t.is_complex():
        t = t.to(_COMPLEX_OF.get(t.dtype, torch.complex128))
    if at_least is not None and at_least == torch.complex128 and t.dtype == torch.complex64:
        t = t.to(torch.complex128)
    return t


def as_field(x):
    """Like as_complex, but float32 / float64 arrays stay REAL: the 2-D transforms read them directly
    (PM_FLAG_REAL_INPUT), which halves the bytes of the first pass and avoids a complex copy."""
    t = as_device(x)
    if t.is_complex() or t.dtype in (torch.float32, torch.float64):
        return t
    return as_complex(t)


def cdtype_of(t):
    """Complex dtype a transform of `t` produces."""
    return t.dtype if t.is_complex() else _COMPLEX_OF[t.dtype]


def code(t):
    return _COMPLEX_CODE[cdtype_of(t)]


def stream_ptr():
    return ctypes.c_void_p(_cur_stream())


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class tuning_local:
    """``with tuning_local(fold=0, mix=0): ...`` -- performance knobs for the CALLING THREAD only (pm_set_tuning_local), restored on
    exit.  The process-wide pm_set_tuning is for start-up configuration; two threads driving the library at once (one pipeline per
    thread, prysm's own advice) each take this.

    Blocks nest: the library keeps ONE private copy of the knobs per thread and can only reset it as a whole, so this class keeps the
    thread's stack of open blocks and re-applies the outer ones when an inner block ends (or fails to start -- a knob the build
    refuses leaves nothing half-applied).  Knobs set by direct pm_set_tuning_local calls are outside that stack and are lost at the
    first exit."""

    _tls = threading.local()
    epoch = 0       # bumped whenever a block starts or ends: answers derived from the planner (e.g. _ops.on_register_engine) are cached per epoch

    def __init__(self, **knobs):
        self.knobs = knobs

    @classmethod
    def signature(cls):
        """the calling thread's open blocks as a hashable value: what a cached planner answer is valid for"""
        return tuple(tuple(sorted(b.items())) for b in getattr(cls._tls, 'stack', ()))

    @classmethod
    def _replay(cls, lib):
        lib.pm_reset_tuning_local()
        for outer in getattr(cls._tls, 'stack', []):
            for k, v in outer.items():
                lib.pm_set_tuning_local(k.encode(), int(v))

    def __enter__(self):
        lib = load()
        try:
            for k, v in self.knobs.items():
                check(lib.pm_set_tuning_local(k.encode(), int(v)))
        except Exception:
            self._replay(lib)
            raise
        if not hasattr(self._tls, 'stack'):
            self._tls.stack = []
        self._tls.stack.append(dict(self.knobs))
        tuning_local.epoch += 1
        return self

    def __exit__(self, *exc):
        stack = getattr(self._tls, 'stack', [])
        if stack:
            stack.pop()
        self._replay(load())
        tuning_local.epoch += 1
        return False


_workspaces = {}


def workspace(nbytes):
    """Scratch buffer in HBM, reused per (device, stream); stream-ordered so back-to-back calls are safe."""
    if nbytes <= 0:
        return None
    key = (_cur_dev(), _cur_stream())
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(int(nbytes), dtype=torch.uint8, device=device())
        _workspaces[key] = w
    return w
