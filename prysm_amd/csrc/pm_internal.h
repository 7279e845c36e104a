// Internal host-side declarations shared by the translation units of libprysm_amd.so
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/prysm_amd.h"
#include "bluestein.h"
#include "fft_io.h"

namespace pm {

bool mix_length(int64_t n);   // fft_mixed.hip: n (<= 8192) has a mixed-radix plan: at least two factors of at most 20 each

// error plumbing (capi.hip)
int fail(int code, const char* fmt, ...);
inline int hip_rc(hipError_t e) { return int(e); }

// plan cache (capi.hip): W_n^k = exp(-2 pi i k / n), k in [0, n), rounded once from long double
template <typename T> const cx<T>* twiddles(int64_t n, int* err);
const cx<double>* twiddles_f64(int64_t n, int* err);

// direct (any length) DFT, dft_direct.hip.  One transform axis with an input AxisMap, `nseq`
// sequences; element (seq, q) at src[seq*s_seq + q*s_i].
template <typename T>
struct DirectIn {
    const cx<T>* src;
    int64_t s_seq, s_i;
    AxisMap ax;
    int nseq;
    int conj;
    int real;   // src is a REAL array (T): strides count real elements
    // PUPIL SYNTHESIS on the fly (mixed-radix rows only, round 4): the element is amp exp(2 pi i k2 opd), as in the engine's row loader
    // (fft_io.h RowLoadNat).  synth 3: src holds packed (amplitude, OPD) pairs; 2: src is the real OPD map (strides in real elements)
    // and the amplitude comes from `amp` (amp_kind 0 unit, 1 float, 2 double, 3 bool / uint8; amp_ld elements between its rows)
    int synth;
    const void* amp;
    int amp_kind;
    int64_t amp_ld;
    double k2;
    // a stack of fields in ONE launch (round 5, mixed-radix entry points): nb fields, `bstride` elements between their inputs; the
    // composite register engine runs them as grid.y, the general kernels field by field
    int nb = 1;
    int64_t bstride = 0;
};
// rows: out[seq*ld + k] (natural complex intermediate, k in [0,n))
template <typename T>
int direct_rows(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, const cx<double>* tw, hipStream_t st);
// rows with full output treatment (1-D API): AxisMap / scale / conj on the output
template <typename T>
int direct_rows_out(const DirectIn<T>& in, const RowStoreNat<T>& out, const cx<double>* tw, hipStream_t st);
// columns: sequences are columns (seq = column index c), output through the ColStoreNat epilogue
template <typename T>
int direct_cols(const DirectIn<T>& in, const ColStoreNat<T>& out, const cx<double>* tw, hipStream_t st);

// Bluestein path (bluestein.hip): lengths that are not powers of two, up to 4096, as engine transforms of length
// MB = 2^ceil(log2(2n - 1)).  Same contracts as direct_rows / direct_cols; `scratch` (256 B aligned) holds
// blue_rows_scratch / blue_cols_scratch bytes.  blue_tables (capi.hip plan cache): [w (n) | B (MB)], see bluestein.h.
template <typename T> const cx<T>* blue_tables(int64_t n, int* err);
size_t blue_rows_scratch(size_t es, int64_t nseq, int64_t n);
size_t blue_cols_scratch(size_t es, int64_t ncols, int64_t n);
// `o` (1-D API): the result goes through that output view (window / rotation, scale, conj) instead of to out[seq*ld + k]
template <typename T>
int blue_rows(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, void* scratch, hipStream_t st, const RowStoreNat<T>* o = nullptr);
template <typename T>
int blue_cols(const DirectIn<T>& in, const ColStoreNat<T>& out, void* scratch, hipStream_t st);

// Mixed-radix path (fft_mixed.hip): lengths up to 8192 whose prime factors are all <= 19 (<= 13 until round 4), one kernel per axis with the data in LDS, no
// scratch.  Same contracts as direct_rows / direct_rows_out (`o`) / direct_cols.
// `fold` (round 4): the rows are taken in pairs (g, g + H) of an array of 2 H rows and the intermediate holds one radix-2 step of the
// column transform (two planes of H rows, fft_mixed.h MixRowOut): tw = the device table of W_M^g (M = 2 H), swap = rows rotated by H
template <typename T>
struct MixFold {
    int H, swap;
    const cx<T>* tw;
};
template <typename T>
int mix_rows(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, hipStream_t st, const RowStoreNat<T>* o = nullptr, const MixFold<T>* fold = nullptr,
             int64_t out_bstride = 0);      // out_bstride: elements between the natural outputs of the fields of a stack (in.nb > 1)
template <typename T>
int mix_cols(const DirectIn<T>& in, const ColStoreNat<T>& out, hipStream_t st);
// the composite register engine (fft_ce.h): lengths with a compile-time plan, plain views.  false: not taken (the general kernel runs);
// true: launched, *rc holds the status
template <typename T> bool ce_rows(const DirectIn<T>& in, cx<T>* out, int64_t out_ld, const RowStoreNat<T>* o, hipStream_t st, int* rc, int64_t out_bstride = 0);
template <typename T> bool ce_cols_mul(const DirectIn<T>& in, const MidMul<T>& mm, cx<T>* dst, int64_t dst_pitch, hipStream_t st, int* rc);
template <typename T> bool ce_cols(const DirectIn<T>& in, const ColStoreNat<T>& out, hipStream_t st, int* rc);
template <typename T> bool ce_has_plan(int n);      // lengths with a built plan (tools/ce_gen.py)
// its kernels address with one unsigned 32-bit byte offset per lane: the largest offset of a view must fit
static inline bool ce_fits32(int64_t elems, size_t es) { return elems >= 0 && uint64_t(elems) * es < (uint64_t(1) << 32); }
constexpr int kCeMaxSeqs = 16;      // sequences per workgroup of any built shape
// middle pass of fft2 -> x H -> ifft2 on a composite column length: the columns of the natural intermediate `in` (sequence = column) come
// back in dst[row * dst_pitch + column] as the unnormalised inverse column transform of (column spectrum x H); in place allowed
template <typename T>
int mix_cols_mul(const DirectIn<T>& in, const MidMul<T>& mm, cx<T>* dst, int64_t dst_pitch, hipStream_t st);

// both axes at once (capi.hip blue2d_run): the chirp multiplies around ONE fused fft2 -> x (B1 (x) B2) -> ifft2 chain of size
// MB1 x MB2 (the 2-D cyclic convolution with the separable chirp)
template <typename T> int blue_pre2d(const Blue2dIn<T>& in, cx<T>* a, const cx<T>* w1, const cx<T>* w2, hipStream_t st);
template <typename T>
int blue_post2d(const cx<T>* t, int n1, int n2, const cx<T>* w1, const cx<T>* w2, const ColStoreNat<T>& out, hipStream_t st);

// the same chain with the chirp multiplies riding on its first load and its last store (fft_row_*.hip instantiations)
template <typename T> int launch_row_chirp_tiled(int logn, int var, const RowLoadChirp<T>&, const RowStoreTiled<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t);
template <typename T> int launch_row_tiled_chirp(int logn, int var, const RowLoadTiled<T>&, const RowStoreChirp<T>&, const cx<T>* tw, int nseq, hipStream_t);

// power-of-two lengths above the engine's longest transform (bigfft.hip; orchestration: capi.hip big2d_run)
template <typename T> int big_pre_rows(const Blue2dIn<T>& in, int M, int np, int R, cx<T>* Y, const cx<T>* twN, hipStream_t st);
template <typename T>
int big_finish(const cx<T>* F, int mp, int np, int Rm, int Rn, const cx<T>* twM, const ColStoreNat<T>& out, hipStream_t st);

// engine launchers (fft_row_*.hip / fft_col_*.hip); `var` = tuning variant (see fft_kernels.h)
template <typename T> int launch_row_tiled(int logn, int var, const RowLoadNat<T>&, const RowStoreTiled<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_nat(int logn, int var, const RowLoadNat<T>&, const RowStoreNat<T>&, const cx<T>* tw, int nseq, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_tiled(int logm, int var, const ColLoadTiled<T>&, const ColStoreNat<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_nat(int logm, int var, const ColLoadNat<T>&, const ColStoreNat<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_col_mul(int logm, const ColLoadTiled<T>&, const MidMul<T>&, const ColStoreTiled<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t, int nbatch = 1, int mode = 0);
int pm_num_cus();   // compute units of the current device (cached per device)
template <typename T> int launch_col_mul_crop(int logm, const ColLoadTiled<T>&, const MidMul<T>&, const ColStoreTiledCrop<T>&, const cx<T>* tw, int ntiles, int log_g, hipStream_t);
template <typename T> int launch_row_from_tiled(int logn, int var, const RowLoadTiled<T>&, const RowStoreNat<T>&, const cx<T>* tw, int nseq, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_fold(int logn, const RowLoadNat<T>&, const RowStoreFold<T>&, const cx<T>* tw, int npairs, int log_g, hipStream_t, int nbatch = 1);
template <typename T> int launch_row_unfold(int logn, const RowLoadFold<T>&, const RowStoreNat<T>&, const cx<T>* tw, int npairs, hipStream_t, int nbatch = 1);
// tile width of the column pass; MUST match ColCfgSel in fft_kernels.h (CI * E)
inline int col_tile_width_for(int dtype, int logm, int var) {
    const int ci = (var == 2 && logm == 11) ? 8 : (logm <= 12 ? 4 : 2);
    return ci * (dtype == PM_C64 ? 2 : 1);
}

// runtime tuning knobs (capi.hip): PM_TUNE="row_var=0,nt_in=1,nt_out=1" or pm_set_tuning().  They choose among equivalent tilings
// and routes that each ship for some size / precision; one-off experiments live in tools/ (exp_*.cpp, exp_*.py), not here.
struct Tuning {
    int col_var = -1;   // column-pass tiling (ColCfgSel): 0 64 B tiles, 2 128 B tiles for 2048-point columns, -1 auto (2 for the planes of a
                        // folded 4096-row complex128 transform)
    int row_var = -1;   // (ignored since round 5) row-pass tiling (RowCfgSel): 0 plain, 1 half-LDS (re / im exchanged separately), 4 two rows per
                        // thread, 5 one row per workgroup: the measured best per length and precision is the one that is built (row_variant() below)
    // non-temporal input loads / output stores: 0 off, 1 on, -1 auto (by array size vs the 256 MiB
    // Infinity Cache: measured on MI355X, streaming hints pay once the arrays no longer fit beside the
    // intermediate -- input from ~128 MiB, output from ~256 MiB; they cost a few % below that)
    int nt_in = -1, nt_out = -1;
    int log_k = -1;     // layout tiles of the intermediate are 2^log_k column-pass tiles wide; -1 auto
    int gemm_3m = 1;          // complex products as three real MFMA chains (3M) instead of four
    int gemm_dma = 1;         // complex64 shapes that are multiples of 64 x 64 x 16 take the LDS-DMA kernel (cgemm_dma_kernel)
    int gemm_dma_wgs = 512;   // ... 128 x 128 tiles when there are at least this many, else 64 x 64 (two or three workgroups per CU hide each
                              // other's barriers); K is split until the launch has this many workgroups.  Measured (profiles/r02/exp_gemm_shapes.log):
                              // config 4 pair 148.5 us at 512 with 64 x 64 tiles against 155.5 (128 x 128, 256) and 164.1 (64 x 64, 256)
    int gemm_wk = 1;          // ... K split over wave groups INSIDE the workgroup when that fills the chip without slabs (cgemm.hip gemm_dma_plan):
                              // bit 0 64 x 64 tiles with two K-groups (8 waves); removed in round 5 (experiments/README.md; the value is refused): bit 1 64 x 32 with two, bit 2 32 x 32 with four;
                              // 0: round 2's forms only.  Config 4 (profiles/r03/exp_gemm_forms.log): 160.5 us at 0, 153.0 at 1, 155.3 at 2, 155.9 at 4
    int gemm_tile = 0;        // force its tile edge (64 / 128); 0 = auto
    int gemm_bm = 64;         // rows of the GEMM workgroup tile (64 or 128)
    int gemm_bk = 0;          // 0 default K-tile depth, 32 doubles it
    int gemm_min_wgs = 1024;   // split K until the GEMM launch has at least this many workgroups
    int row_log_g = 1;  // sibling group of row-pass workgroups (rows q .. q+2^g-1 on one XCD)
    int r2c = 1;             // real inputs take the Hermitian path where it is legal AND pays (capi.hip r2c_legal): real epilogues / centre
                             // normalisation; 0: never, 2: also for a plain complex spectrum
    int fold = -1;           // radix-2 step of the column transform folded into the row pass: -1 auto, 0 never, 1 wherever legal
    int blue_min = 96;        // shortest non-power-of-two length that takes the Bluestein path (shorter ones, and lengths
                             // above 4096, run on the direct O(n^2) kernel); 0 disables the path
    int mix = 1;              // composite lengths (primes <= 19, up to 8192) on the mixed-radix kernel (fft_mixed.h) instead of Bluestein and of
                              // the radix-R step (profiles/r03/exp_mix.log: 1536^2 complex64 42 us against 61, 2560^2 104 against 119); 0: as in
                              // round 2; 2: the 3 / 5 / 7 x 2^k lengths stay on the radix-R step (bigfft.hip)
    int mix_maxr = 20;        // ... plan within factors of at most this when the length allows (10 / 16 / 20: the kernel classes)
    int mix_log_g = -1;       // ... its column pass: 2^this adjacent tiles per XCD (-1 auto: as many as share a 128 B line)
    int mix_seqs = 0, mix_tc = 0, mix_nt = 0, mix_ntc = 0;   // ... force its rows per workgroup / columns per workgroup / threads per workgroup of the row pass / of the column pass (0 = auto)
    int mix_pad = 1;          // ... LDS slots padded after the blocks of the first two levels, pads chosen by a bank model (fft_mixed.hip mix_pick_pads); 0: none
    int mix_fused = 1;        // ... fft2 -> x H -> ifft2 on a composite column length as three passes with the mixed-radix middle pass (0: two pm_fft2)
    int mix_min = 32;         // ... from this length (shorter ones stay on the direct fp64-accumulating kernel)
    int blue_fuse = 1;        // both-axes form on engine lengths: chirp multiplies inside the chain's first load / last store (1)
                             // or as separate kernels around it (0)
    int blue_2d = 1;          // both axes on the Bluestein path: one fused fft2 x B ifft2 chain (1) or axis by axis (0)
    int mixed_radix = 1;      // lengths 3 / 5 / 7 x 2^k take one radix-R step around engine transforms (bigfft.hip); 0: Bluestein, as in round 1
    int big_native_log = 13;  // log2 of the longest length handed to the engine as it is; longer powers of two (up to 4x) take
                             // one radix-2 / radix-4 step around engine transforms (bigfft.hip).  Tests lower it to run that
                             // path on small arrays.
    int herm_wide = 1;        // Hermitian column pass: tiles of 16 complex64 / 8 complex128 columns for 2048-point columns -- its real epilogues then
                              // write 64 B pieces, direct and mirrored (MTF 4096^2 fp32 88.4 -> 81.4 us, fp64 176 -> 168; profiles/r02/exp_mtf_wide.log)
    int herm_t = -1;          // the TRANSPOSED Hermitian form (fft_hermt.h: real-input column transforms, then row transforms that store each row and its
                              // mirror image as whole lines): -1 auto (capi.hip hermt_legal: where it measured faster), 0 never, 1 wherever legal
    int herm_t_rowvar = -1;   // ... tiling of its row pass at 4096 complex64 points: 4 two rows per thread, 0 one row per workgroup; -1: the complex path's
    int herm_t_fold = -1;     // ... its column pass folded (planes of M/2-point tiles): -1 auto (from 4096 rows), 0 never, 1 from 2048 rows
    int spectral = 8;         // pm_fft2_spectral: wavelengths per launch pair (fft_spectral.h; <= 8); 1: the plain loop of pm_fft2 calls
    int spectral_area_log = 24;   // ... for transforms of fewer than 2^this bins (capi.hip spectral_fast has the measurements)
    int spectral2 = 0;        // removed in round 5 (experiments/README.md; the value is refused): groups of this many (2 .. 4) on the kernels that keep four waves per SIMD (fft_spectral2.h) where
                              // the shape qualifies (complex64, rows of 1024 .. 4096 samples, every output bin kept); 0: round 3's forms.
                              // Measured (profiles/r04/exp_spectral2.log, us per wavelength: loop / groups of 8 / these in groups of 2, 4):
                              // 4096^2 108.3 / 106.1 / 111.8, 109.0; 2048^2 39.6 / 21.5 / 28.0, 23.7; 1024^2 24.1 / 10.4 / 16.9, 13.7
    int spectral2_keep = 0;   // ... groups of 3 / 4: the raw (amplitude, OPD) values stay in registers between the pairs (three waves per SIMD)
                              // instead of being read again from L2 / Infinity Cache
    int spectral2_min_log = 0;    // ... for transforms of at least 2^this bins
    int spectral_mode = 3;    // ... bit 0: its row pass keeps the packed map in registers, bit 1: its column pass accumulates in registers
    int colmul_mode = 3;      // middle pass of the fused chain at 2048-point column tiles (fft_kernels.h launch_col_mul_one): 3 lean addressing,
                              // two 512-thread workgroups per CU (126 VGPRs) where the pass qualifies (whole unrotated tiles, separable
                              // multiplier), else 0 = one tile per workgroup (184 VGPRs, one per CU); removed in round 5 (experiments/README.md; the value is refused): 1 the generic kernel under
                              // a 128-VGPR cap (spills), 2 persistent prefetching workgroups (twiddles / hy in LDS).  Measured
                              // (profiles/r03/exp_colmul.log, us, modes 0 / 1 / 2 / 3): middle pass of config 3 159.1 / 198.1 / 135.7 / 115.4;
                              // chain 4096^2 complex128 383 / 430 / 365 / 355, complex64 186 / 176 / 175 / 165, 2048^2 complex128 90 / 94 / 86 / 81
    int engine_p8 = 0;        // removed in round 5 (experiments/README.md; the value is refused): the radix-8 engine (8 points per thread) for the folded 4096^2 complex64 transform: bit 0 its row pass,
                              // bit 1 its column pass, bit 2 a 64-register cap (eight waves per SIMD) instead of 128
    int fft_stagger = -1, fft_stagger_col = -1, fft_stagger_mid = 0, fft_stagger_r2c = 0, fft_stagger_herm = -1;    // start-up stagger of the engine's plain row / column kernels (fft_kernels.h engine_log_g), units of 512 cycles x 0 .. 7; 0 = off, -1 = auto
    int mix_fold = 0;         // removed in round 5 (experiments/README.md; the value is refused): a radix-2 step of a composite column transform folded into the mixed-radix row pass where the whole column's tile would
                              // take a CU's LDS (capi.hip plan_fft2 mix_fold)
    int mix_pers = 0;         // removed in round 5 (experiments/README.md; the value is refused): its column pass as persistent workgroups with the next tile prefetched where a CU holds one tile (mix_cols_pers_kernel)
    int mix_engine = 1;       // composite lengths with a compile-time plan on the register engine (fft_ce.h); 0: the general kernel everywhere
    int ce_rows_seqs = 0, ce_cols_seqs = 0, ce_log_g = -1;     // ... a built alternative of its rows / columns per workgroup (0: the default shape)
    int mix_stagger = 4;      // ... start-up stagger of the column kernel's workgroups in units of 512 cycles x 0 .. 7 where a CU holds one tile (fft_mixed.h MixShape::stagger); 0 = off
    int mix_ablate = 0;       // removed in round 5 (experiments/README.md; the value is refused): timing-only ablations of the mixed-radix kernels (fft_mixed.h MixShape::ablate; results are wrong)
    int two_units = 0;        // removed in round 5 (experiments/README.md; the value is refused): two units per workgroup in the passes of a 2048-point complex64 transform (bit 0 rows, bit 1 columns)
    int col_log_g = -1;       // sibling group of column-pass workgroups: 2^this adjacent tiles on one XCD back to back (-1: from the layout, capi.hip sibling_log_g)
    int stagger_group = 0;    // column kernels: the start-up stagger hashed per sibling group instead of per workgroup (fft_kernels.h engine_log_g)
    int batch_ws_mib = 128;   // batched transforms: fields per launch pair are chosen so their intermediates take
                             // at most this many MiB (measured best at 128; they should survive in the Infinity Cache between the passes)
};
Tuning& tuning();
// measured on MI355X (profiles/r01/sweep*.log): the half-LDS variant wins for complex128 rows of 2048 points (24.6 vs 30.5 us)
inline int row_variant(int dtype, int logn) {
    // (the knob row_var is accepted and ignored since round 5: one tiling per length and precision is built, fft_kernels.h built_row_var)
    // two rows per thread (variant 4) for complex64 from 4096 points: 53.5 -> 51.4 us at 4096^2, 226 -> 207 us at 8192^2;
    // it loses for complex128 (register pressure) and makes no difference at 2048
    if (dtype == PM_C64 && logn >= 12) return 4;
    // 2048 points: complex64 one row per 128-thread workgroup (twice the workgroups: 17.9 -> 16.5 us at 2048^2), complex128 the
    // half-LDS exchange (24.3 vs 28.3 us)
    if (logn == 11) return dtype == PM_C64 ? 5 : 1;
    return 0;
}

constexpr int kEngineMaxLog = 13;
inline int engine_log2(int64_t n) {  // log2(n) if n is a power of two the engine handles, else -1
    if (n < 2 || (n & (n - 1))) return -1;
    int l = 0;
    while ((int64_t(1) << l) < n) ++l;
    return l <= kEngineMaxLog ? l : -1;
}

// split n = R n' for the path of bigfft.hip (n' a power of two the engine transforms): R = 1 (native), 2 or 4 (powers of two above the
// engine's longest transform), 3 / 5 / 7 (mixed-radix lengths from 96: 1536, 2560, 3584 ...); 0 = not a length this path takes
// (`mix_owns` false: the OTHER axis needs this path -- 1536 x 16384, 6144 x 12288 -- so a length the composite-length kernel would
// take alone keeps its radix-R step here; without it such shapes fell to the both-axes Bluestein form at 16384 x 32768)
inline int big_split(int64_t n, bool mix_owns = true) {
    if (n < 2) return 0;
    const int64_t native = int64_t(1) << tuning().big_native_log;
    if ((n & (n - 1)) == 0) {
        if (n <= native) return 1;
        if (n == 2 * native) return 2;
        if (n == 4 * native) return 4;
        return 0;
    }
    if (!tuning().mixed_radix || n < 96) return 0;
    if (mix_owns && tuning().mix == 1 && n >= tuning().mix_min && mix_length(n)) return 0;    // the composite-length kernel takes these (up to 8192)
    for (int R = 3; R <= 7; R += 2) {
        if (n % R) continue;
        const int64_t q = n / R;
        if ((q & (q - 1)) == 0 && q >= 16 && q <= native && q <= (int64_t(1) << kEngineMaxLog)) return R;
    }
    return 0;
}

// ... and for the 2-D transform (capi.hip big2d_run) the sub-transforms of length n / R may also run on the mixed-radix kernel (round 4):
// composites ABOVE 8192 whose cofactor of 2, 3, 4, 5 or 7 is a length either kernel takes -- 10000 = 2 x 5000, 9000 = 2 x 4500,
// 12000 = 2 x 6000, 20000 = 4 x 5000 -- which used to convolve at 32768 points per axis (Bluestein); and, beside an axis that needs the
// step (`mix_owns` false), a composite length up to 8192 as it is (R = 1).  pm_fft1's big path keeps big_split (engine sub-transforms).
inline int big_split2d(int64_t n, bool mix_owns = true) {
    const int r = big_split(n, mix_owns);
    if (r) return r;
    if (!tuning().mix || !tuning().mixed_radix) return 0;
    if (n <= (int64_t(1) << kEngineMaxLog)) return (!mix_owns && n >= tuning().mix_min && mix_length(n)) ? 1 : 0;
    for (int R : {2, 3, 4, 5, 7}) {
        if (n % R) continue;
        const int64_t q = n / R;
        if (q > (int64_t(1) << kEngineMaxLog)) continue;
        if (engine_log2(q) >= 0 || (q >= tuning().mix_min && mix_length(q))) return R;
    }
    return 0;
}

// lengths the mixed-radix kernel takes: not a power of two (the engine's), not one of the radix-R lengths above when `big` says that
// path owns them
// ... its kernels address with 32-bit element offsets from a base that is uniform per workgroup and multiply indices by pitches with
// the 24-bit multiplier: pitches below 2^24 elements, and in column mode (offset = point x pitch) the whole n x pitch array below 4 GiB
inline bool mix_fits(int64_t n, int64_t pitch, size_t es, bool col) {
    if (pitch < 0 || pitch >= (int64_t(1) << 24)) return false;
    return !col || uint64_t(n) * uint64_t(pitch) * es < (uint64_t(1) << 32);
}
inline bool use_mix(int64_t n) {
    return tuning().mix && n >= tuning().mix_min && n >= 2 && (n & (n - 1)) != 0 && mix_length(n);
}

// lengths the Bluestein path takes (bluestein.h): not a power of two, at least blue_min, and a convolution length
// MB >= 2n - 1 that the engine runs as it is (use_blue: n <= 4096; the axis-by-axis form and pm_fft1_ws need that) or that
// the big power-of-two path runs (use_blue_long: n <= 16384; only the both-axes form, two big transforms around the multiply)
// (`mix` false: the caller found that the array does not fit the mixed-radix kernel's 32-bit offsets -- mix_fits -- and plans without it)
inline bool use_blue_long(int64_t n, bool mix = true) {
    const int lo = tuning().blue_min;
    return lo > 0 && n >= lo && n >= 2 && (n & (n - 1)) != 0 && n <= (int64_t(1) << 20) && !(mix && use_mix(n)) && big_split(blue_conv_len(n)) >= 1;
}
inline bool use_blue(int64_t n, bool mix = true) { return use_blue_long(n, mix) && big_split(blue_conv_len(n)) == 1; }
// MIXED shapes with one axis the paths above cannot take alone (a length in (4096, 16384] that is not a power of two, or a
// 16384-point axis beside a non power of two): the both-axes form runs them with the OTHER axis convolved as well -- any length
// from 2 whose convolution length the transforms reach (wasteful by the padding of that axis, but O(n log n))
inline bool blue_reach(int64_t n) { return tuning().blue_min > 0 && n >= 2 && n <= (int64_t(1) << 20) && big_split(blue_conv_len(n)) >= 1; }
inline bool blue_needs_both(int64_t n, bool mix = true) {
    const bool pow2 = (n & (n - 1)) == 0;
    if (!blue_reach(n) || (mix && use_mix(n))) return false;
    return pow2 ? big_split(n) > 1 : (n >= tuning().blue_min && !use_blue(n, mix));
}

}  // namespace pm
