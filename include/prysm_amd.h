/* prysm_amd -- C ABI of the MI355X (gfx950) physical-optics propagation engine.
 *
 * This is the drop-in boundary for the ONE hot path of brandondube/prysm v0.22
 * named in BASELINE.json: pupil<->focus FFT propagation, angular-spectrum free
 * space propagation, the matrix-DFT fixed-sampling focus and the |.|^2
 * intensity / incoherent polychromatic sum.  The reference is pure Python and has
 * no FFI; its plug surface is the `prysm.mathops` backend shim
 * (prysm/mathops.py:11-45) through which every hot-path module reaches
 * numpy / scipy.fft / BLAS.  Each entry point below names the reference call
 * site(s) whose arithmetic it replaces.  The host-side mirror of the reference's
 * Python interface (same names, arguments, error behaviour) is the `prysm_amd`
 * package, which binds these symbols with ctypes; INTEGRATION.md shows the
 * binding a prysm maintainer would add.
 *
 * Conventions
 *  - All pointers are DEVICE pointers (HBM) unless stated otherwise; buffers are
 *    owned by the caller (PyTorch allocates them).  The library allocates only
 *    immutable per-(N, dtype, device) twiddle tables in a plan cache.
 *  - Arrays are row-major `a[y][x]` (prysm convention), leading dimension in
 *    ELEMENTS.  Complex values are interleaved (re, im).
 *  - Every function only ENQUEUES work on `stream` (a hipStream_t; NULL = the
 *    default stream) and never synchronises.
 *  - Return value: 0 = ok, < 0 = argument error (see pm_last_error()),
 *    > 0 = a hipError_t.
 *  - gfx950 only.  There is no CPU path: without a GPU every compute entry
 *    point fails with a hipError_t.
 */
#ifndef PRYSM_AMD_H
#define PRYSM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PM_VERSION 107 /* 0.1.0 */

/* dtype codes */
enum { PM_C64 = 0, PM_C128 = 1, PM_F32 = 2, PM_F64 = 3, PM_BOOL = 4 };

/* error codes (negative) */
enum {
    PM_OK = 0,
    PM_ERR_ARG = -1,         /* bad argument (null pointer, negative size, bad enum) */
    PM_ERR_UNSUPPORTED = -2, /* size / dtype combination not implemented */
    PM_ERR_WORKSPACE = -3    /* workspace too small; query with pm_fft2_workspace() */
};

/* epilogues of the last FFT pass */
enum {
    PM_EPI_NONE = 0,       /* complex output */
    PM_EPI_ABS2 = 1,       /* real output  |scale * X|^2           (Wavefront.intensity fused) */
    PM_EPI_ABS2_ACCUM = 2, /* real output  out += weight*|scale*X|^2 (incoherent polychromatic sum) */
    PM_EPI_ABS = 3,        /* real output  |scale * X|    (otf.mtf_from_psf, prysm/otf.py:77-103)  -- real-input transforms only */
    PM_EPI_ARG = 4         /* real output  angle(scale*X) (otf.ptf_from_psf, prysm/otf.py:106-135) -- real-input transforms only */
};

/* multiplier applied to the complex result before it is stored */
enum { PM_MUL_NONE = 0, PM_MUL_FULL = 1, PM_MUL_SEPARABLE = 2 };

/* flags */
enum {
    PM_FLAG_PASS1_ONLY = 1, /* profiling: run only the row pass    */
    PM_FLAG_PASS2_ONLY = 2, /* profiling: run only the column pass */
    PM_FLAG_SYNTH_INPUT = 8, /* `in` is the real OPD map (float); the transformed field is synth_amp * exp(i synth_k opd),
                             * synthesised while the row pass loads it -- Wavefront.from_amp_and_phase
                             * (prysm/propagation/wavefront.py:58-79) fused into focus: the complex pupil never exists in
                             * memory.  `in` is float for PM_C64, double for PM_C128 (fp64 sincospi per sample: 4096^2 329 -> 255 us
                             * against synthesis + transform).  Row lengths: powers of two (the engine's row loader) and composites of primes
                             * <= 19 up to 8192 (the mixed-radix row kernel's first stage, round 4); PM_ERR_UNSUPPORTED otherwise. */
    PM_FLAG_SYNTH_PACKED = 32, /* with PM_FLAG_SYNTH_INPUT: `in` holds (amplitude, OPD) float / double PAIRS (in_ld in pairs), synth_amp is ignored.
                             * One 8-byte load per element instead of two 4-byte loads from two arrays: a loop over wavelengths packs
                             * its two maps once (the polychromatic recipe: 133 -> 101 us per wavelength at 4096^2) */
    PM_FLAG_NORM_DC = 16,   /* divide the result by its DC bin X[0][0] before the epilogue -- the centre normalisation
                             * `data / data[cy, cx]` of the OTF routines (prysm/otf.py:62-74).  Real-input transforms on the
                             * Hermitian path only (see PM_FLAG_REAL_INPUT), where that bin is real; PM_ERR_UNSUPPORTED otherwise */
    PM_FLAG_REAL_OUTPUT = 64, /* pm_fft2_mul_ifft2 with PM_FLAG_REAL_INPUT: `out` is a REAL array (out_ld in real elements) that receives
                             * the REAL PART of the result -- what convolution.conv / apply_transfer_functions keep for a real object
                             * (prysm/convolution.py:29-31, 110-113).  The chain then runs on half spectra end to end (real rows as
                             * N/2 packed complex points, the Hermitian part of the multiplier, N/2-point inverse row transforms):
                             * 32 instead of 56 bytes per sample.  A full (PM_MUL_FULL) multiplier, unpadded power-of-two sizes from
                             * 2048^2 samples (rows of 64 .. 8192; smaller fields: tuning key "r2c" = 2), rotations by 0 or N/2 along x,
                             * unwindowed output, one field;
                             * PM_ERR_UNSUPPORTED otherwise (the caller takes the real part of the complex chain instead). */
    PM_FLAG_REAL_INPUT = 4  /* `in` is a REAL array of the precision that goes with dtype (float / double); in_ld and
                             * in_bstride count real elements.  fft2 of a real PSF / object / actuator map
                             * (prysm/otf.py:31, prysm/convolution.py:27-28,82-85) without a complex copy: pass 1 reads
                             * half the bytes.  A FORWARD transform of an unpadded real field with power-of-two lengths (>= 32
                             * per row) and an unwindowed output takes the Hermitian path: half the spectrum is computed and each
                             * result stored twice (at (u, k) and, conjugated, at (-u, -k)) -- along x (half-length row transforms,
                             * N/2 + 1 columns through the column pass) or, where every rotation is 0 or half a length and it
                             * measured faster, along y (real-input column transforms, then M/2 row transforms that store each
                             * row and its mirror image as whole lines; tuning key "herm_t");
                             * PM_EPI_ABS / PM_EPI_ARG / PM_FLAG_NORM_DC exist on that path */
};

/* One axis of a windowed, rotated view.  A logical (transform-sized) axis of
 * length n is related to memory by
 *     position p = (i + shift) mod n,   memory index q = p - off,   0 <= q < len.
 * On the INPUT side logical element i reads mem[q] (zero outside the window):
 *     fft.ifftshift            -> shift = n/2            (prysm/propagation/fft.py:24)
 *     fttools.pad2d            -> off = ceil((n-len)/2)  (prysm/fttools.py:88-94), never materialised
 * On the OUTPUT side transform bin k is written to mem[q] (dropped outside):
 *     fft.fftshift             -> shift = n/2
 *     fttools.crop_center      -> off = ceil((n-len)/2)  (prysm/fttools.py:122-124)
 */
typedef struct pm_axis {
    int64_t n;
    int64_t len;
    int64_t off;
    int64_t shift;
} pm_axis;

/* 2-D complex transform with fused pad / shift / crop / scale / multiply / |.|^2.
 * Replaces, in one call:
 *   fft.fftshift(fft.fft2(fft.ifftshift(pad2d(x, Q)), norm=...))      prysm/propagation/fft.py:23-25 (focus)
 *   ... ifft2 ...                                                      fft.py:44,65,84 (unfocus, adjoints) + crop_center
 *   fft.fft2(field) * tf  and  fft.ifft2(.)                            prysm/propagation/angular_spectrum.py:35,41-42,76
 *   re*re + im*im of the result                                        prysm/propagation/wavefront.py:146-151
 *   fft.fftshift(fft.fft2(fft.ifftshift(psf)))                         prysm/otf.py:31
 */
typedef struct pm_fft2_desc {
    int32_t dtype;      /* PM_C64 or PM_C128 */
    int32_t direction;  /* -1: exp(-2 pi i ..) (fft2), +1: exp(+2 pi i ..) (ifft2, unnormalised) */
    int32_t epilogue;   /* PM_EPI_* */
    int32_t flags;      /* PM_FLAG_* */
    double scale;       /* multiplies the complex result: 1/sqrt(MN) for norm='ortho', 1/(MN) for ifft2 */
    double weight;      /* PM_EPI_ABS2_ACCUM weight */
    pm_axis in_y, in_x;   /* input view  (rows, columns) */
    pm_axis out_y, out_x; /* output view (rows, columns) */
    int64_t in_ld, out_ld;
    int32_t mul_kind;   /* PM_MUL_*: result *= mul[k_y][k_x] (FULL) or mul_y[k_y]*mul_x[k_x] (SEPARABLE), */
    int32_t mul_conj;   /*           indexed by the unshifted transform bin; conj -> multiply by conj(mul) */
    const void* mul;    /* FULL: (M x N) complex array; SEPARABLE: length-M complex vector (rows) */
    const void* mul_x;  /* SEPARABLE: length-N complex vector (columns) */
    int64_t mul_ld;
    /* Batch of independent fields in ONE launch pair (wavelengths / field points of a polychromatic or
     * multi-field model -- the per-wavelength loop of docs/source/how-tos/Polychromatic Propagation.ipynb).
     * Field b reads in + b*in_bstride and writes out + b*out_bstride (elements of the respective array; for the
     * |.|^2 epilogues out elements are real).  batch = 0 means 1.  mul_bstride / mul_x_bstride: elements between
     * per-field multipliers (FULL: arrays; SEPARABLE: the row-factor and column-factor vectors), 0 = shared.
     * The workspace grows by the batch factor.  PM_EPI_ABS2_ACCUM needs distinct outputs per field. */
    int64_t batch;
    int64_t in_bstride, out_bstride;
    int64_t mul_bstride, mul_x_bstride;
    /* PM_FLAG_SYNTH_INPUT: amplitude array of the in window's shape (NULL: unit amplitude), its type (PM_F32, PM_F64,
     * PM_BOOL), leading dimension, and k = 2 pi / (wavelength_um * 1e3) for an OPD in nm */
    const void* synth_amp;
    int32_t synth_amp_dtype;
    int32_t synth_reserved;
    int64_t synth_amp_ld;
    double synth_k;
} pm_fft2_desc;

/* Transform lengths (per axis): powers of two from 2 to 8192 run on the Stockham engine; composite lengths from 32 to 8192 whose prime
 * factors are all <= 19 (1000, 1020, 1536, 2592, 3000, 6000 ...: what scipy.fft factors natively) run on their own factors in one LDS-resident
 * mixed-radix kernel per axis with no scratch (3000^2 complex64: 87 us; arrays of 4 GiB and more keep the routes below); 16384 and
 * 32768 take one radix-2 / radix-4 step around engine transforms (16384^2 complex64: 4.3 ms), and so do 3 / 5 / 7 x 2^k above 8192
 * (10240, 12288 ...: radix 3 / 5 / 7) and -- round 4 -- composites above 8192 whose cofactor of 2, 3, 4, 5 or 7 is a length the mixed-radix
 * kernel takes (10000 = 2 x 5000, 9000, 12000, 20000 ...: 10000^2 complex64 1.9 ms), when the other axis is a power of two, such a length
 * or a composite the mixed-radix kernel takes; other lengths from 96 to 4096 (a prime
 * factor above 19: 997, 1009 ...) run on the engine through Bluestein's identity (chirp multiply, power-of-two convolution of length >= 2n - 1, chirp multiply;
 * when both axes are such lengths the 2-D convolution is ONE fused fft2 x B ifft2 chain, and that form reaches 16384 per axis by
 * convolving at 16384 / 32768 points: 8000^2 complex64 8.6 ms); shorter lengths, and other lengths
 * up to 32768, run on a direct O(n^2) kernel with fp64 accumulation.  Anything else is PM_ERR_UNSUPPORTED.  The reference
 * takes any length through scipy.fft (prysm/propagation/fft.py:24).
 *
 * bytes of workspace pm_fft2 needs for this descriptor (the tiled intermediate, plus the Bluestein scratch) */
size_t pm_fft2_workspace(const pm_fft2_desc* d);

int pm_fft2(const pm_fft2_desc* d, const void* in, void* out, void* workspace, size_t workspace_bytes,
            void* stream);

/* The wavelength loop of a polychromatic PSF (docs/source/how-tos/Polychromatic Propagation.ipynb cell 3:
 *     for wvl, w in zip(wavelengths, weights): psf += w * abs(focus(amp * exp(1j * 2 pi / wvl * opd)))**2  ),
 * result-equivalent to `count` pm_fft2 calls with d->synth_k = k[b], d->weight = weight[b] on the same input and accumulator
 * (the descriptor must carry PM_FLAG_SYNTH_INPUT and PM_EPI_ABS2_ACCUM; its own synth_k / weight are ignored; k and weight
 * are HOST arrays).  With PM_FLAG_SYNTH_PACKED and engine lengths below 4096^2 bins (complex128: rows of up to 2048 samples) the loop
 * runs as one launch pair per group of wavelengths: the row pass reads the packed (amplitude, OPD) map once per group, the column
 * pass sums w_b |.|^2 over the group in registers and touches the accumulator once -- 16 + 16 / B bytes per sample and wavelength
 * instead of 32 in complex64 (B = 8: tuning keys "spectral", "spectral_area_log"; measured 1.8 - 2.5x the loop from 1024^2 to 2048^2,
 * no gain at 4096^2, which keeps the loop).  The sum runs in wavelength order; only its association differs from the loop's
 * (acc + (w_0 i_0 + w_1 i_1 + ..) per group).  Other descriptors run the plain loop.
 * Workspace: pm_fft2_spectral_workspace(d, count) bytes. */
size_t pm_fft2_spectral_workspace(const pm_fft2_desc* d, int32_t count);
int pm_fft2_spectral(const pm_fft2_desc* d, int32_t count, const double* k, const double* weight, const void* in, void* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* Fused  out = window( ifft2( fft2( pad(in) ) * H ) )  in THREE passes (row FFT, column FFT x H x column IFFT
 * in registers, row IFFT): 6 N^2 s bytes of HBM traffic instead of the 8 N^2 s of two pm_fft2 calls.
 * Replaces fft.ifft2(fft.fft2(field) * tf) of angular_spectrum / angular_spectrum_adjoint
 * (prysm/propagation/angular_spectrum.py:35,41-42,76) and the fft2 * fft2 -> ifft2 core of convolution.conv
 * (prysm/convolution.py:27-30).  Transform lengths: powers of two <= 8192 on both axes, or (round 4: one field, complex output) a COLUMN
 * length from 32 to 8192 whose primes are <= 19 beside a row length of either kind -- the middle pass then keeps each column in LDS through
 * the forward stages of its factorisation, the multiplier and the same stages transposed (csrc/fft_mixed.h; angular spectrum 3000^2
 * complex128: 281 us against 369 for two pm_fft2 calls).  PM_ERR_UNSUPPORTED (workspace query: 0) otherwise: the caller composes two
 * pm_fft2 calls.  Uses the fields of pm_fft2_desc: dtype, scale (applied
 * once, at the end), in_* / out_* views, mul_* (required); direction / epilogue / weight are ignored. */
size_t pm_fft2_mul_ifft2_workspace(const pm_fft2_desc* d);
int pm_fft2_mul_ifft2(const pm_fft2_desc* d, const void* in, void* out, void* workspace, size_t workspace_bytes,
                      void* stream);

/* Batched 1-D complex transform along one axis of a 2-D array, zero padded or
 * truncated to n on input (numpy `fft.fft(x, n, axis=)` semantics).
 * Replaces fft.fft / fft.ifft at prysm/fttools.py:301-321,335-355,387,519-533 (CZT, FFTDFT).
 *   axis = 1: transform each row;  axis = 0: transform each column.
 *   `t` describes the transform axis on the input (zero pad: len < n) and `t_out` on the output
 *   (crop / shift); `batch` is the extent of the other axis.  */
int pm_fft1(int32_t dtype, int32_t direction, int32_t axis, int64_t batch, const pm_axis* t_in,
            const pm_axis* t_out, double scale, const void* in, int64_t in_ld, void* out, int64_t out_ld,
            void* stream);
/* The same with a workspace: pm_fft1_workspace() bytes (256 B aligned; 0 when none is needed) put 16384 / 32768 and 3 / 5 / 7 x 2^k
 * points on the FFT engine by one radix-R step (an unrotated input view), other lengths that are not powers of two (96 .. 4096)
 * through Bluestein's identity; without it (pm_fft1, or a smaller / NULL workspace) such lengths run on the direct O(n^2) kernel.  FFTDFT with K = 1 / (dx dfx) not a power of two
 * (prysm/fttools.py:484-533) is the caller that needs it. */
size_t pm_fft1_workspace(int32_t dtype, int32_t axis, int64_t batch, int64_t n);
int pm_fft1_ws(int32_t dtype, int32_t direction, int32_t axis, int64_t batch, const pm_axis* t_in,
               const pm_axis* t_out, double scale, const void* in, int64_t in_ld, void* out, int64_t out_ld,
               void* workspace, size_t workspace_bytes, void* stream);

/* One axis of a chirp-Z transform in ONE kernel (fttools.CZT.__call__ / .adjoint, prysm/fttools.py:297-361: per axis
 * `fft(x * b, K) -> * H -> ifft -> slice -> * a * phase`):
 *     out[.., m] = scale * post[m] * IFFT_K( FFT_K( pad_K(pre . in) ) . H )[out_off + m],     m < out_len,
 * the 1 / K of the inverse included.  axis = 1: the nseq sequences are rows of `in` (in_len samples each, placed at
 * [in_off, in_off + in_len) of the K-point sequence -- 0 for the forward transform, Mx - 1 ... for the adjoint's zero embedding);
 * axis = 0: columns.  pre (in_len), H (K) and post (out_len) are complex device vectors of `dtype`, each optionally conjugated
 * (the adjoint); pre and post may be NULL.  K: a power of two from 16 to 8192 (PM_ERR_UNSUPPORTED otherwise: the caller composes
 * pm_fft1 and pm_scale_sep).  The K-point sequence stays in the registers of its workgroup between the two transforms. */
int pm_czt_axis(int32_t dtype, int32_t axis, int64_t nseq, int64_t K, int64_t in_len, int64_t in_off, int64_t out_len, int64_t out_off,
                const void* pre, int32_t pre_conj, const void* H, int32_t h_conj, const void* post, int32_t post_conj, double scale,
                const void* in, int64_t in_ld, void* out, int64_t out_ld, void* stream);

/* One axis of fttools.FFTDFT (prysm/fttools.py:392-535: phase ramp, zero-padded FFT of length K, crop, phase ramp) in ONE kernel:
 *     out[.., m] = scale * post[m] * T_K( pad_K(pre . in) )[out_off + m],     m < out_len,
 * T_K the unnormalised K-point transform with exp(-2 pi i ..) (direction = -1) or exp(+2 pi i ..) (+1); windows, vectors and axis as
 * in pm_czt_axis (same kernel with the multiplier and the second transform switched off).  K: a power of two from 16 to 8192
 * (PM_ERR_UNSUPPORTED otherwise: compose pm_fft1_ws and pm_scale_sep). */
int pm_fft1_ramp(int32_t dtype, int32_t direction, int32_t axis, int64_t nseq, int64_t K, int64_t in_len, int64_t in_off, int64_t out_len,
                 int64_t out_off, const void* pre, int32_t pre_conj, const void* post, int32_t post_conj, double scale, const void* in,
                 int64_t in_ld, void* out, int64_t out_ld, void* stream);

/* --- pointwise / synthesis kernels -------------------------------------------------------- */

/* out = a * b (op 0), a * conj(b) (op 1); complex, same shape (rows x cols).
 * Wavefront.__mul__ (prysm/propagation/wavefront.py:360-411), _adjoint_multiply (_kernels.py:29-37). */
int pm_cmul(int32_t dtype, int32_t op, int64_t rows, int64_t cols, const void* a, int64_t a_ld, const void* b,
            int64_t b_ld, void* out, int64_t out_ld, void* stream);

/* out = scale * r * a: r REAL (float for PM_C64, double for PM_C128), a complex, same shape.
 * Wavefront.intensity_adjoint, Gbar = 2 * Ibar * E (prysm/propagation/wavefront.py:282-298), as one sweep. */
int pm_rmul(int32_t dtype, int64_t rows, int64_t cols, const void* r, int64_t r_ld, const void* a, int64_t a_ld, double scale,
            void* out, int64_t out_ld, void* stream);

/* out[i][j] = in[i][j] * ry[i] * cx[j] * scale, ry / cx optional complex vectors, each optionally conjugated.
 * The chirp / phase-ramp multiplies of CZT and FFTDFT (prysm/fttools.py:297-323,508-535). */
int pm_scale_sep(int32_t dtype, int64_t rows, int64_t cols, const void* in, int64_t in_ld, const void* ry,
                 int32_t ry_conj, const void* cx, int32_t cx_conj, double scale, void* out, int64_t out_ld,
                 void* stream);

/* out = re^2 + im^2 (accumulate = 0) or out += weight * (re^2 + im^2).  wavefront.py:146-151;
 * polynomials.sum_of_2d_modes (prysm/polynomials/fitting.py:7-37) as a running weighted sum. */
int pm_abs2(int32_t dtype, int64_t rows, int64_t cols, const void* in, int64_t in_ld, void* out, int64_t out_ld,
            int32_t accumulate, double weight, void* stream);

/* out_abs = |in|, out_arg = atan2(im, re) of a complex array in ONE sweep (either output may be NULL): the MTF and PTF of
 * otf.mtf_ptf_otf_from_psf (prysm/otf.py:167-203) from the centre-normalised OTF the transform already produced. */
int pm_abs_arg(int32_t dtype, int64_t rows, int64_t cols, const void* in, int64_t in_ld, void* out_abs, int64_t abs_ld, void* out_arg,
               int64_t arg_ld, void* stream);

/* out = sum_b weights[b] * modes[b] (accumulate = 0) or out += ...; REAL images of the precision that goes with
 * dtype (PM_C64: float, PM_C128: double), modes[b] at modes + b*mode_stride elements; weights is a HOST array.
 * polynomials.sum_of_2d_modes = tensordot(weights, modes, axes=(0, 0)) (prysm/polynomials/fitting.py:7-37), the
 * incoherent sum of the polychromatic recipe over a batch of intensities. */
int pm_sum_modes(int32_t dtype, int64_t nmodes, int64_t rows, int64_t cols, const void* modes, int64_t mode_stride,
                 int64_t modes_ld, const double* weights, int32_t accumulate, void* out, int64_t out_ld, void* stream);

/* Encircled energy of a PSF from its centre-normalised MTF (Baliga & Cohn 1988):
 *   out[r] = radius_r * df^2 * sum_ij mtf[i][j] * J1(2 pi radius_r nu_ij) / nu_ij,
 * nu = hypot of the FFT-centred frequency grid of spacing df (cy/mm; the zero bin is nudged to 1e-16 like the reference),
 * radii in mm in a HOST array (the reference divides its micron radii by 1e3), `out` a DEVICE array of nradii doubles.
 * Replaces otf._encircled_energy_geometry / _encircled_energy_core (prysm/otf.py:319-343,390-414) for every radius of
 * otf.encircled_energy (otf.py:346-387) in ceil(nradii / 8) passes over the MTF.  mtf is REAL (PM_C64: float, PM_C128: double);
 * J1 and the sums are fp64, reduced in a fixed order (reproducible).  workspace: pm_encircled_energy_workspace() bytes. */
size_t pm_encircled_energy_workspace(void);
int pm_encircled_energy(int32_t dtype, int64_t rows, int64_t cols, const void* mtf, int64_t mtf_ld, double df, int64_t nradii,
                        const double* radii_mm, double* out, void* workspace, size_t workspace_bytes, void* stream);
/* MTF-plane gradient of the encircled energies: mtf_bar[i][j] = sum_r ee_bar[r] * radius_r * J1(2 pi radius_r nu_ij) / nu_ij * df^2
 * (otf.encircled_energy_adjoint, prysm/otf.py:417-472; the caller routes it through mtf_from_psf_adjoint).  radii_mm and
 * ee_bar are HOST arrays; mtf_bar is a REAL rows x cols device array of the precision that goes with dtype. */
int pm_encircled_energy_adjoint(int32_t dtype, int64_t rows, int64_t cols, double df, int64_t nradii, const double* radii_mm,
                                const double* ee_bar, void* mtf_bar, int64_t mtf_bar_ld, void* stream);

/* Resample a measured complex focal-plane-mask map at focal coordinates: scipy.ndimage.map_coordinates(order 0 | 1,
 * mode='nearest') of the real and imaginary parts at row = (yf - center_y)/dx + map_rows/2, col = (xf - center_x)/dx +
 * map_cols/2; points outside [0, n-1] on either axis take fill (a rows x cols complex array) or, with fill NULL, the
 * constant fill_re + i fill_im.  xf / yf are REAL arrays of the precision that goes with dtype, addressed as
 * xf[r*xf_sy + c*xf_sx] (a stride of 0 broadcasts a coordinate vector).  prepare_measured_fpm
 * (prysm/propagation/coronagraph.py:128-200).  Other spline orders: PM_ERR_UNSUPPORTED here, see pm_sample_spline. */
int pm_sample_map(int32_t dtype, int32_t order, int64_t map_rows, int64_t map_cols, const void* map, int64_t map_ld, double dx,
                  double center_x, double center_y, int64_t rows, int64_t cols, const void* xf, int64_t xf_sy, int64_t xf_sx,
                  const void* yf, int64_t yf_sy, int64_t yf_sx, const void* fill, int64_t fill_ld, double fill_re, double fill_im,
                  void* out, int64_t out_ld, void* stream);

/* Spline orders 2 .. 5 of the same resampling (scipy.ndimage.map_coordinates(order, mode='nearest') as prepare_measured_fpm
 * calls it, prysm/propagation/coronagraph.py:193-194): pm_spline_prefilter pads the map by 12 edge samples and runs scipy's
 * recursive B-spline prefilter along both axes in fp64 (once per measured map) into `coeff`, a complex128
 * (map_rows + 24) x (map_cols + 24) array; pm_sample_spline evaluates the tensor-product B-spline at the focal coordinates,
 * arguments as pm_sample_map with `coeff` in place of the map. */
int pm_spline_prefilter(int32_t dtype, int32_t order, int64_t map_rows, int64_t map_cols, const void* map, int64_t map_ld,
                        void* coeff, int64_t coeff_ld, void* stream);
int pm_sample_spline(int32_t dtype, int32_t order, int64_t map_rows, int64_t map_cols, const void* coeff, int64_t coeff_ld,
                     double dx, double center_x, double center_y, int64_t rows, int64_t cols, const void* xf, int64_t xf_sy,
                     int64_t xf_sx, const void* yf, int64_t yf_sy, int64_t yf_sx, const void* fill, int64_t fill_ld,
                     double fill_re, double fill_im, void* out, int64_t out_ld, void* stream);

/* P = amp * exp(i * k * opd), k = 2 pi / (wavelength_um * 1e3) for opd in nm.
 * amp may be NULL (unit amplitude: phase_screen).  amp_dtype in {PM_F32, PM_F64, PM_BOOL}.
 * Wavefront.from_amp_and_phase / phase_screen (wavefront.py:58-96), phase_prefix (_kernels.py:40-43). */
int pm_pupil_synth(int32_t dtype, int64_t rows, int64_t cols, const void* amp, int32_t amp_dtype, int64_t amp_ld,
                   const void* opd, int64_t opd_ld, double k, void* out, int64_t out_ld, void* stream);

/* out[i][j] = exp(i * c * (x[i][j]^2 + y[i][j]^2)); Wavefront.thin_lens (wavefront.py:98-144), c = -pi/(w f). */
int pm_quadratic_phase(int32_t dtype, int64_t rows, int64_t cols, const void* x, int64_t x_ld, const void* y,
                       int64_t y_ld, double c, void* out, int64_t out_ld, void* stream);

/* Separable Fresnel transfer-function factors: hy[i] = exp(-i pi wvl_mm z ky[i]^2), ky = fftfreq(rows, dx)
 * rounded to the real dtype first (angular_spectrum.py:105-113).  hx likewise.  Vectors of length rows / cols. */
int pm_as_tf_vectors(int32_t dtype, int64_t rows, int64_t cols, double wvl_um, double dx, double z, void* hy,
                     void* hx, void* stream);

/* out = outer(hy, hx) -- materialises the transfer function for API parity (angular_spectrum.py:114). */
int pm_outer(int32_t dtype, int64_t rows, int64_t cols, const void* hy, const void* hx, void* out, int64_t out_ld,
             void* stream);

/* out window copy: out (orows x ocols) = fill everywhere, then in placed at (off_y, off_x); negative offsets
 * crop.  fttools.pad2d constant mode / crop_center (prysm/fttools.py:43-125).  elem_bytes in {1,4,8,16}. */
int pm_embed(int32_t elem_bytes, int64_t irows, int64_t icols, const void* in, int64_t in_ld, int64_t orows,
             int64_t ocols, int64_t off_y, int64_t off_x, const void* fill_elem_host, void* out, int64_t out_ld,
             void* stream);

/* The index-mapping modes of np.pad that fttools.pad2d(mode=...) forwards to (prysm/fttools.py:96-98): out (orows x ocols) holds
 * in at (off_y, off_x) and, around it, in[map(r)][map(c)] with mode 1 = 'edge', 2 = 'reflect', 3 = 'symmetric', 4 = 'wrap'
 * (any pad width, also wider than the array).  elem_bytes in {1, 4, 8, 16}.  The statistical modes ('mean', 'maximum', 'minimum',
 * 'median') and 'linear_ramp' are not entry points: the mirror package composes them from device tensor reductions (off the hot path). */
int pm_pad_index(int32_t elem_bytes, int32_t mode, int64_t irows, int64_t icols, const void* in, int64_t in_ld, int64_t orows,
                 int64_t ocols, int64_t off_y, int64_t off_x, void* out, int64_t out_ld, void* stream);

/* The chirps of one chirp-Z axis from its scalars, one launch (prysm/fttools.py:364-389 _prepare_czt_basis: arange / exp / zero-pad
 * as a dozen array operations per axis -- the polychromatic recipe builds one executor per wavelength): with e(t) = exp(2 pi i t),
 *   b[j] = e(half n^2), n = j - N/2 (j < N);   a[i] = e(half q^2), q = i - M/2 + shift (i < M);
 *   h[t] = e(-half (d + shift)^2), d = t - M/2 - (N - 1 - N/2) for t < N + M - 1, zero up to K  (its transform is the H of pm_czt_axis);
 * half = sign dx dfx / 2, shift = f[M/2] / dfx. */
int pm_czt_vectors(int32_t dtype, int64_t N, int64_t M, int64_t K, double shift, double half, void* b, void* a, void* h,
                   void* stream);

/* --- matrix DFT --------------------------------------------------------------------------- */

/* E[m][n] = exp(sign * 2 pi i * f[m] * x[n]) (M x N), phases reduced in fp64, rounded once.
 * f and x are real device vectors of the complex dtype's real type.  fttools.MDFT.__init__
 * (prysm/fttools.py:187-191). */
int pm_mdft_basis(int32_t dtype, int64_t M, int64_t N, const void* f, const void* x, int32_t sign, void* E,
                  int64_t E_ld, void* stream);

/* The same basis for the FFT-centred grids of dft.coordinates_for_focus (prysm/propagation/dft.py:58-65), generated inside the
 * kernel instead of read from vectors -- prepare_executor then costs two launches, not a dozen small array operations:
 *     x[n] = (n - N/2) * x_step                      (fftrange(N) * pupil_dx)
 *     f[m] = ((m - M/2) * f_step + f_shift) * f_scale   ((fftrange(M) * focal_dx + focal_shift) / (wavelength * efl))
 * every operation rounded once in the real type of dtype, i.e. bit for bit the vectors numpy builds at config.precision. */
int pm_mdft_basis_grid(int32_t dtype, int64_t M, int64_t N, double f_step, double f_shift, double f_scale, double x_step,
                       int32_t sign, void* E, int64_t E_ld, void* stream);

/* C (M x N) = alpha * opA(A) (M x K) @ opB(B) (K x N), complex, on the MFMA matrix cores.
 *   opA: 0 = A, 1 = conj(A), 2 = A^T, 3 = A^H     (A stored M x K for 0/1, K x M for 2/3)
 *   opB: likewise                                   (B stored K x N for 0/1, N x K for 2/3)
 * The two GEMMs of fttools.MDFT.__call__ / .adjoint (prysm/fttools.py:201-228).
 * Leading dimensions must be below 2^22 elements (PM_ERR_UNSUPPORTED otherwise). */
int pm_cgemm(int32_t dtype, int32_t opA, int32_t opB, int64_t M, int64_t N, int64_t K, double alpha,
             const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, void* workspace,
             size_t workspace_bytes, void* stream);
/* bytes of split-K workspace pm_cgemm wants for this shape (0 = none; without it the GEMM runs unsplit) */
size_t pm_cgemm_workspace(int32_t dtype, int64_t M, int64_t N, int64_t K);

/* R = (accumulate ? R : 0) + weight * |alpha * opA(A) @ opB(B)|^2, R REAL (float for PM_C64): the second product of a matrix-DFT
 * focus with Wavefront.intensity (prysm/propagation/wavefront.py:146-151) and the weighted sum of the polychromatic recipe
 * (polynomials.sum_of_2d_modes, prysm/polynomials/fitting.py:7-37) in its epilogue -- the complex focal field is never written.
 * Same workspace as pm_cgemm.  Only the shapes the LDS-DMA kernel takes (PM_C64, M and N multiples of 64, K of 16, 16-byte aligned
 * operands, even leading dimensions); PM_ERR_UNSUPPORTED otherwise: compose pm_cgemm and pm_abs2. */
int pm_cgemm_abs2(int32_t dtype, int32_t opA, int32_t opB, int64_t M, int64_t N, int64_t K, double alpha, const void* A,
                  int64_t lda, const void* B, int64_t ldb, void* R, int64_t ldr, double weight, int32_t accumulate, void* workspace,
                  size_t workspace_bytes, void* stream);

/* --- housekeeping ------------------------------------------------------------------------- */
/* Real-input 2-D spectrum on ANY even width (round 5; prysm/otf.py:28-33 transform_psf, :62-135 the centre-normalised MTF / PTF / OTF
 * take any size through scipy): the real M x N array IS an M x N/2 complex array z[r][j] = x[r][2j] + i x[r][2j+1]; transform that
 * with pm_fft2 (forward, no rotations -- half the work, on whatever route its lengths take) and hand the result `zf` to this sweep,
 * which untangles F = fft2(x) from it, applies the rotation of the INPUT by (in_shift_y, in_shift_x) samples as a phase, divides by
 * F[0][0] (norm_dc; real: the sum of the samples), multiplies by `scale`, takes the epilogue (PM_EPI_NONE complex out, PM_EPI_ABS,
 * PM_EPI_ABS2, PM_EPI_ARG real out) and writes all M x N bins rotated by (out_shift_y, out_shift_x).  Lengths the library's own
 * Hermitian path takes (powers of two, PM_FLAG_REAL_INPUT in pm_fft2) do not need it. */
int pm_r2c_untangle(int32_t dtype, int64_t M, int64_t N, const void* zf, int64_t zf_ld, int64_t in_shift_y, int64_t in_shift_x,
                    int64_t out_shift_y, int64_t out_shift_x, int32_t epilogue, int32_t norm_dc, double scale, void* out, int64_t out_ld,
                    void* stream);

int pm_version(void);
const char* pm_last_error(void);   /* thread-local message for the last negative return */
int pm_plan_prepare(int32_t dtype, int64_t n);   /* build + cache the tables of a transform length now (twiddles; for a length on the
                                                  * Bluestein path its chirp tables and the twiddles of the convolution length): the first
                                                  * transform of a length otherwise does it, with a blocking upload that a hipGraph capture
                                                  * cannot record */
void pm_shutdown(void);            /* free cached tables */
/* performance knobs (they choose among equivalent routes and tilings; results agree to rounding): "col_var", "row_var" pick kernel
 * tilings, "nt_in" / "nt_out" in {0,1} make the input loads / output stores non-temporal, "fold", "log_k", "batch_ws_mib",
 * "gemm_3m", "gemm_min_wgs" tune the engine and the GEMM; routing of awkward lengths: "blue_min" (shortest length on the Bluestein
 * path, 0 = off), "blue_2d" / "blue_fuse" (both-axes form; chirp multiplies inside the chain), "big_native_log" (log2 of the longest
 * length given to the engine as it is; the GPU tests lower it to run the 16384-point path on small arrays); real inputs: "r2c"
 * (Hermitian path: 0 never, 1 where it pays, 2 wherever legal), "herm_t" (its transposed form, real-input column transforms first:
 * -1 where it measured faster, 0 never, 1 wherever legal) with "herm_t_fold" (its column pass as planes of half-height tiles).  The full list with
 * defaults and measurements: struct Tuning in prysm_amd/csrc/pm_internal.h.  Also read once from the environment:
 * PM_TUNE="nt_in=1,fold=0". */
int pm_set_tuning(const char* key, int32_t value);
/* the same knob for the CALLING host thread only: its first call gives the thread a private copy of the process-wide values, which
 * every later library call on that thread reads; pm_reset_tuning_local() returns the thread to the shared values.  This is the form
 * to use when several host threads drive the library at once -- prysm's own advice for several pipelines / devices is one thread
 * each (docs/source/how-tos/GPU and Exascale Computing.ipynb, file line 66) -- pm_set_tuning changes what every thread without a
 * private copy sees. */
int pm_set_tuning_local(const char* key, int32_t value);
void pm_reset_tuning_local(void);
/* Which route does this descriptor take?  Writes ONE line into buf (n >= 64 bytes; longer lines are cut) that names the planner's
 * decisions for op = 0 (pm_fft2) or op = 1 (pm_fft2_mul_ifft2) under the calling thread's knobs: the route ("engine", "engine-fold",
 * "hermitian[-fold]", "hermitian-transposed", "natural-mixed", "natural", "radix-step", "bluestein-2d[-big]"; "fused", "fused-composite", "hermitian-chain",
 * "composed"), the kernel class of each axis ("stockham", "mixed-radix", "bluestein", "direct"), tile width, layout and workspace
 * bytes.  Host logic only -- no device is touched, so a table of shapes can be pinned to its routes on a machine without a GPU
 * (the reference reaches every size through one scipy call, prysm/fttools.py:23-31; here a shape that slips to a slow route
 * should fail a test, not a benchmark).  Returns 0, or the error pm_fft2 would return for an invalid descriptor. */
int pm_plan_explain(const pm_fft2_desc* d, int32_t op, char* buf, size_t n);
/* time `reps` launches of each pass of the transform with hipEvents on `stream`; ms[0] = row pass,
 * ms[1] = column pass (average per launch).  Used by bench.py for the roofline object. */
int pm_fft2_time_passes(const pm_fft2_desc* d, const void* in, void* out, void* workspace,
                        size_t workspace_bytes, int reps, double* ms, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PRYSM_AMD_H */
