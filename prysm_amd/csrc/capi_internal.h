// Internal interface between the translation units behind the extern "C" surface (round 6: capi.hip was one 2000-line file):
//   capi_core.hip  error text, twiddle / plan tables, tuning knobs, device queries   (+ pm_version, pm_set_tuning*, pm_last_error, pm_plan_prepare, pm_shutdown)
//   capi_plan.hip  host-only planning: which route a descriptor takes, workspace sizes, argument checks  (+ the workspace queries, pm_plan_explain)
//   capi_run.hip   the routes themselves: parameter blocks and kernel launches per plan
//   capi.hip       the transform entry points (pm_fft2, pm_fft2_spectral, pm_fft2_mul_ifft2, pm_fft1*, pm_czt_axis, pm_fft1_ramp, pm_fft2_time_passes)
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>
#include "bluestein.h"
#include "fft_mixed.h"
#include "pm_internal.h"
#include "fft_r2c_types.h"
#include "fft_hermt_types.h"
#include "fft_conv1_types.h"
#include "fft_spectral_types.h"
#include "fft_c2r_types.h"

namespace pm {

// ---------------------------------------------------------------- 2-D transform
struct Fft2Plan {
    int logn, logm;       // engine log2 sizes or -1 (direct)
    int tc;               // column-pass tile width when both passes run on the engine, else 0 (natural intermediate)
    bool r2c;             // real input on the Hermitian path (fft_r2c.h): N/2-point row transforms, N/2 + 1 columns, mirrored stores
    bool r2c_t;           // ... in its transposed form (fft_hermt.h, round 6): real-input COLUMN transforms into an M/2 x N natural intermediate,
                          // then full-length row transforms that store every row and its mirror image as whole lines
    int col_var;          // column-pass tiling (ColCfgSel): 2 = 128 B tiles for the planes of a folded 4096-row complex128 transform
    int log_k;            // layout tile width TL = tc << log_k
    size_t ws_bytes;      // total
    size_t ws_field;      // bytes of intermediate per field (256 B aligned)
    int64_t nbatch;       // fields
    int64_t chunk;        // fields per launch pair: the intermediates of one chunk stay resident in the 256 MiB
                          // Infinity Cache between the two passes, consecutive chunks reuse the same workspace
    bool fold;            // one radix-2 step of the column transform is taken in the row pass (RowStoreFold): the column
                          // pass then runs two planes of M/2-point tiles
    bool mix_n, mix_m;    // the row / column transforms take the mixed-radix kernel (composite lengths, fft_mixed.hip)
    bool mix_fold;        // ... with one radix-2 step of the column transform folded into the row pass (MixRowOut fold_h): half-length column tiles
    int64_t w_ld;         // row pitch of the NATURAL intermediate (tc == 0), in elements: N, or N rounded up to whole 128 B lines when the
                          // column pass is the mixed-radix kernel -- its 32 / 64 B pieces then share lines only inside one XCD group
                          // (3000 complex64 columns: rows of 24000 B put every other row half a line off and the pass read 1.52x its bytes)
    bool blue_n, blue_m;  // the row / column transforms take the Bluestein path (non-power-of-two lengths, bluestein.hip)
    size_t blue_off;      // its scratch sits behind the intermediates in the workspace (shared by the two passes)
    int big_rn, big_rm;   // power-of-two lengths above the engine's: radix of the extra step per axis (1 = none), 0 = not this path
    bool blue_big;        // blue2d whose convolution length exceeds the engine's: two big power-of-two transforms around the multiply
    bool blue2d;          // both axes: chirp multiply -> ONE fused fft2 x (B1 (x) B2) ifft2 chain of size MB1 x MB2 -> chirp multiply
                          // (blue2d_run); the workspace is then [a (M x N) | c (M x N) | workspace of the fused chain]
};

// ---------------------------------------------------------------- fused fft2 -> multiply -> ifft2
struct FusedPlan {
    int logn, logm, tc, log_k;
    size_t w1_bytes, w2_bytes;   // tiled buffers PER FIELD: stored input rows x N, and M x N (shared when rows == M)
    bool inplace;
    int64_t nbatch, chunk;       // fields, fields per launch triple
    size_t ws_bytes;             // total workspace
    bool fold;                   // radix-2 step of the column transforms folded into the first / last row pass
    bool mixmid;                 // composite column length: natural intermediates of pitch w_ld, the mixed-radix middle pass (fft_mixed.h)
    int64_t w_ld;
};

// ---- real object, real result: the chain on half spectra (fft_c2r.h)
struct HermConvPlan {
    int logn, logm, tc, log_k;
    bool fold;          // radix-2 step of the column transforms in the first / last row pass, as in the complex chain
    size_t ws_bytes;
};

// ---- capi_plan.hip / capi_run.hip
int sibling_log_g(int log_k);
inline AxisMap to_map(const pm_axis& a) { return AxisMap{int(a.n), int(a.len), int(a.off), int(a.shift)}; }
int64_t batch_chunk(int64_t nb, size_t ws_field);
Fft2Plan plan_fft2(const pm_fft2_desc* d, bool allow_r2c = true);
bool plan_fused_mix(const pm_fft2_desc* d, FusedPlan& p);
bool plan_fused(const pm_fft2_desc* d, FusedPlan& p);
void blue2d_desc(pm_fft2_desc& dd, int dtype, int64_t M, int64_t N);
size_t blue2d_fused_ws(int dtype, int64_t M, int64_t N);
int check_fft2(const pm_fft2_desc* d);
bool ce_rows_axis(const pm_fft2_desc* d, const Fft2Plan& p);
bool ce_cols_axis(const pm_fft2_desc* d, const Fft2Plan& p);
bool ce_both_axes_stack(const pm_fft2_desc* d, const Fft2Plan& p);
size_t fft1_big_scratch(size_t es, int axis, int64_t batch, int64_t n);
inline bool fft1_big_ok(const pm_axis* ti) { return big_split(ti->n) > 1 && ti->shift == 0; }
bool herm_conv_plan(const pm_fft2_desc* d, HermConvPlan& p);
bool spectral_fast(const pm_fft2_desc* d, const Fft2Plan& p);
int spectral_group(int32_t count);
int check_fft1(int32_t dtype, int32_t direction, int32_t axis, int64_t batch, const pm_axis* t_in, const pm_axis* t_out);
template <typename T>
int fft2_run(const pm_fft2_desc* d, const Fft2Plan& p, const void* in, void* out, void* ws, hipStream_t st);
template <typename T>
int fused_run(const pm_fft2_desc* d, const FusedPlan& p, const void* in, void* out, void* ws, hipStream_t st);
template <typename T>
int fft1_run(int direction, int axis, int64_t batch, const pm_axis* ti, const pm_axis* to, double scale,
                    const void* in, int64_t in_ld, void* out, int64_t out_ld, hipStream_t st, void* blue_ws = nullptr);
template <typename T>
int czt_axis_run(int32_t axis, int64_t nseq, int64_t K, int64_t in_len, int64_t in_off, int64_t out_len, int64_t out_off,
                        const void* pre, int pre_conj, const void* H, int h_conj, const void* post, int post_conj, double scale,
                        const void* in, int64_t in_ld, void* out, int64_t out_ld, hipStream_t st, int single = 0);
template <typename T>
int herm_conv_run(const pm_fft2_desc* d, const HermConvPlan& p, const void* in, void* out, void* ws, hipStream_t st);
template <typename T>
int fft2_spectral_group(const pm_fft2_desc* d, const Fft2Plan& p, const Spectral& w, const void* in, void* out, void* ws, hipStream_t st);

}  // namespace pm
