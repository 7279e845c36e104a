// extern "C" entry points of libprysm_amd.so: plan cache, 2-D / 1-D transform dispatch.
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

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// ---------------------------------------------------------------- plan cache
// Immutable twiddle tables keyed by (device, element size, n).  Built on the host in long double,
// rounded once, uploaded with a blocking copy at first use (or via pm_plan_prepare); the hot
// path afterwards only reads the map under a mutex.
static std::mutex g_mu;
static std::map<std::tuple<int, int, int64_t>, void*> g_tables;

template <typename T>
static const cx<T>* table_get(int64_t n, int* err) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        *err = int(e);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, int(sizeof(T)), n);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return reinterpret_cast<const cx<T>*>(it->second);
    std::vector<cx<T>> h(size_t(n > 0 ? n : 1));
    const long double pi = acosl(-1.0L);
    for (int64_t i = 0; i < n; ++i) {
        // octant symmetry is not needed for accuracy in long double; one rounding per entry
        const long double a = -2.0L * pi * (long double)i / (long double)n;
        h[size_t(i)] = {T(cosl(a)), T(sinl(a))};
    }
    void* d = nullptr;
    e = hipMalloc(&d, h.size() * sizeof(cx<T>));
    if (e != hipSuccess) {
        *err = int(e);
        return nullptr;
    }
    e = hipMemcpy(d, h.data(), h.size() * sizeof(cx<T>), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        *err = int(e);
        return nullptr;
    }
    g_tables[key] = d;
    return reinterpret_cast<const cx<T>*>(d);
}

template <> const cx<float>* twiddles<float>(int64_t n, int* err) { return table_get<float>(n, err); }
template <> const cx<double>* twiddles<double>(int64_t n, int* err) { return table_get<double>(n, err); }
const cx<double>* twiddles_f64(int64_t n, int* err) { return table_get<double>(n, err); }

// the mixed-radix plan of a composite length (fft_mixed.h) as the kernels read it, same cache, element-size keys 1002 .. 1020 (the planner's cap on the largest factor)
bool mix_plan_for(int n, size_t es, MixPlan& p);
const MixPlan* mix_plan_dev(int n, size_t es, int* err) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        *err = int(e);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, 1000 + tuning().mix_maxr + (es == 8 ? 100 : 0), int64_t(n));     // the plan follows the knob and the precision
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return reinterpret_cast<const MixPlan*>(it->second);
    MixPlan h;
    if (!mix_plan_for(n, es, h)) {
        *err = fail(PM_ERR_UNSUPPORTED, "mixed-radix path: length %d has no plan", n);
        return nullptr;
    }
    void* d = nullptr;
    e = hipMalloc(&d, sizeof(MixPlan));
    if (e == hipSuccess) e = hipMemcpy(d, &h, sizeof(MixPlan), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (d) (void)hipFree(d);
        *err = int(e);
        return nullptr;
    }
    g_tables[key] = d;
    return reinterpret_cast<const MixPlan*>(d);
}

// Bluestein tables [w (n) | B (MB)] of a non-power-of-two length n (bluestein.h), same cache, element-size key + 64
template <typename T>
static const cx<T>* blue_table_get(int64_t n, int* err) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        *err = int(e);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_tuple(dev, int(sizeof(T)) + 64, n);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return reinterpret_cast<const cx<T>*>(it->second);
    std::vector<cx<T>> h;
    blue_make_tables<T>(int(n), blue_conv_len(n), h);
    void* d = nullptr;
    e = hipMalloc(&d, h.size() * sizeof(cx<T>));
    if (e != hipSuccess) {
        *err = int(e);
        return nullptr;
    }
    e = hipMemcpy(d, h.data(), h.size() * sizeof(cx<T>), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        *err = int(e);
        return nullptr;
    }
    g_tables[key] = d;
    return reinterpret_cast<const cx<T>*>(d);
}
template <> const cx<float>* blue_tables<float>(int64_t n, int* err) { return blue_table_get<float>(n, err); }
template <> const cx<double>* blue_tables<double>(int64_t n, int* err) { return blue_table_get<double>(n, err); }

static void tune_set(Tuning& t, const char* key, size_t klen, int v) {
    auto is = [&](const char* k) { return strlen(k) == klen && !strncmp(key, k, klen); };
    if (is("col_var")) t.col_var = v;
    else if (is("log_k")) t.log_k = v > 12 ? 12 : v;   // < 0: auto; log2(N / tile width) makes the intermediate natural (row-major)
    else if (is("row_log_g")) t.row_log_g = v < 0 ? 0 : (v > 3 ? 3 : v);
    else if (is("row_var")) t.row_var = v;
    else if (is("stagger_group")) t.stagger_group = v ? 1 : 0;
    else if (is("col_log_g")) t.col_log_g = v;
    else if (is("gemm_bk")) t.gemm_bk = v;
    else if (is("gemm_bm")) t.gemm_bm = v;
    else if (is("gemm_dma")) t.gemm_dma = v ? 1 : 0;
    else if (is("gemm_dma_wgs")) t.gemm_dma_wgs = v < 1 ? 1 : v;
    else if (is("gemm_tile")) t.gemm_tile = (v == 64 || v == 128) ? v : 0;
    else if (is("gemm_3m")) t.gemm_3m = v ? 1 : 0;
    else if (is("gemm_wk")) t.gemm_wk = v & 7;
    else if (is("gemm_min_wgs")) t.gemm_min_wgs = v < 1 ? 1 : v;
    else if (is("nt_in")) t.nt_in = v;
    else if (is("nt_out")) t.nt_out = v;
    else if (is("fold")) t.fold = v;
    else if (is("mixed_radix")) t.mixed_radix = v ? 1 : 0;
    else if (is("mix")) t.mix = v < 0 ? 0 : (v > 2 ? 2 : v);
    else if (is("mix_min")) t.mix_min = v < 18 ? 18 : v;
    else if (is("mix_fused")) t.mix_fused = v ? 1 : 0;
    else if (is("mix_pad")) t.mix_pad = v ? 1 : 0;
    else if (is("two_units")) t.two_units = v & 3;
    else if (is("mix_ablate")) t.mix_ablate = v & 15;
    else if (is("fft_stagger")) t.fft_stagger = v < 0 ? -1 : (v > 164 ? 164 : v);
    else if (is("fft_stagger_r2c")) t.fft_stagger_r2c = v < 0 ? 0 : (v > 164 ? 164 : v);
    else if (is("fft_stagger_herm")) t.fft_stagger_herm = v < 0 ? -1 : (v > 164 ? 164 : v);
    else if (is("fft_stagger_mid")) t.fft_stagger_mid = v < 0 ? -1 : (v > 64 ? 64 : v);
    else if (is("fft_stagger_col")) t.fft_stagger_col = v < 0 ? -1 : (v > 164 ? 164 : v);
    else if (is("mix_fold")) t.mix_fold = v != 0;
    else if (is("mix_pers")) t.mix_pers = v != 0;
    else if (is("mix_engine")) t.mix_engine = v ? 1 : 0;
    else if (is("ce_rows_seqs")) t.ce_rows_seqs = v < 0 ? 0 : v;
    else if (is("ce_cols_seqs")) t.ce_cols_seqs = v < 0 ? 0 : v;
    else if (is("ce_log_g")) t.ce_log_g = v;
    else if (is("mix_stagger")) t.mix_stagger = v < 0 ? 0 : (v > 64 ? 64 : v);
    else if (is("engine_p8")) t.engine_p8 = v & 7;
    else if (is("mix_maxr")) t.mix_maxr = (v < 2 || v > 20) ? 20 : v;
    else if (is("mix_log_g")) t.mix_log_g = v;
    else if (is("mix_seqs")) t.mix_seqs = v < 0 ? 0 : v;
    else if (is("mix_tc")) t.mix_tc = v < 0 ? 0 : v;
    else if (is("mix_nt")) t.mix_nt = v < 0 ? 0 : v;
    else if (is("mix_ntc")) t.mix_ntc = v < 0 ? 0 : v;
    else if (is("r2c")) t.r2c = v < 0 ? -1 : (v > 1 ? 2 : v);
    else if (is("batch_ws_mib")) t.batch_ws_mib = v < 1 ? 1 : v;
    else if (is("colmul_mode")) t.colmul_mode = v;
    else if (is("herm_wide")) t.herm_wide = v;
    else if (is("herm_t")) t.herm_t = v < 0 ? -1 : (v ? 1 : 0);
    else if (is("herm_t_fold")) t.herm_t_fold = v < 0 ? -1 : (v ? 1 : 0);
    else if (is("herm_t_rowvar")) t.herm_t_rowvar = v;
    else if (is("spectral")) t.spectral = v;
    else if (is("spectral_mode")) t.spectral_mode = v & 3;
    else if (is("spectral_area_log")) t.spectral_area_log = v;
    else if (is("spectral2")) t.spectral2 = (v == 2 || v == 3 || v == 4) ? v : 0;
    else if (is("spectral2_keep")) t.spectral2_keep = v ? 1 : 0;
    else if (is("spectral2_min_log")) t.spectral2_min_log = v;
    else if (is("blue_min")) t.blue_min = v < 0 ? 0 : v;
    else if (is("blue_2d")) t.blue_2d = v ? 1 : 0;
    else if (is("blue_fuse")) t.blue_fuse = v ? 1 : 0;
    else if (is("big_native_log")) t.big_native_log = v < 1 ? 1 : (v > kEngineMaxLog ? kEngineMaxLog : v);
}

// The process-wide defaults (PM_TUNE, pm_set_tuning) and, per host thread, an optional private copy (pm_set_tuning_local): the
// reference's advice for several devices / pipelines is one pipeline per thread (GPU and Exascale Computing.ipynb, file line 66), and
// two threads that pick different routes must not race on one struct.  Every entry point reads the knobs through tuning(), on the
// calling thread.
static Tuning& tuning_global() {
    static Tuning t = [] {
        Tuning x;
        const char* e = getenv("PM_TUNE");   // e.g. PM_TUNE="nt_in=1,fold=0"
        while (e && *e) {
            const char* eq = strchr(e, '=');
            if (!eq) break;
            tune_set(x, e, size_t(eq - e), atoi(eq + 1));
            const char* c = strchr(eq, ',');
            e = c ? c + 1 : nullptr;
        }
        return x;
    }();
    return t;
}
static thread_local bool g_tune_local_on = false;
static thread_local Tuning g_tune_local;

Tuning& tuning() { return g_tune_local_on ? g_tune_local : tuning_global(); }

int pm_fft_stagger(int pass) {
    const Tuning& t = tuning();
    return pass == 2 ? t.fft_stagger_mid : (pass == 1 ? t.fft_stagger_col : (pass == 3 ? t.fft_stagger_r2c : (pass == 4 ? t.fft_stagger_herm : t.fft_stagger)));
}

int pm_stagger_group() { return tuning().stagger_group; }

int pm_num_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

// sibling group of column-pass workgroups: the tiles of one layout-tile row, at most 8
// (knob col_log_g >= 0 overrides: up to 2^5 = the 32 workgroups an XCD's CUs hold at one per CU -- experiments of round 5)
static int sibling_log_g(int log_k) {
    if (tuning().col_log_g >= 0) return tuning().col_log_g > 5 ? 5 : tuning().col_log_g;
    return log_k < 1 ? 1 : (log_k > 3 ? 3 : log_k);
}

static AxisMap to_map(const pm_axis& a) { return AxisMap{int(a.n), int(a.len), int(a.off), int(a.shift)}; }

static int check_axis(const pm_axis& a, const char* name) {
    if (a.n < 1 || a.n > (int64_t(1) << 30)) return fail(PM_ERR_ARG, "%s.n = %lld out of range", name, (long long)a.n);
    if (a.len < 0 || a.len > a.n) return fail(PM_ERR_ARG, "%s.len = %lld must be in [0, n]", name, (long long)a.len);
    if (a.off < 0 || a.off + a.len > a.n) return fail(PM_ERR_ARG, "%s window [off, off+len) must lie in [0, n]", name);
    if (a.shift < 0 || a.shift >= a.n) return fail(PM_ERR_ARG, "%s.shift must be in [0, n)", name);
    return 0;
}

// input mode of the row loader from the descriptor flags: complex, real, or pupil synthesis
template <typename T>
static void set_input_mode(RowLoadNat<T>& lp, const pm_fft2_desc* d) {
    lp.real = (d->flags & PM_FLAG_SYNTH_INPUT) ? ((d->flags & PM_FLAG_SYNTH_PACKED) ? 3 : 2) : ((d->flags & PM_FLAG_REAL_INPUT) ? 1 : 0);
    if (lp.real >= 2) {
        lp.amp = d->synth_amp;
        lp.amp_kind = !d->synth_amp ? 0 : (d->synth_amp_dtype == PM_F32 ? 1 : (d->synth_amp_dtype == PM_F64 ? 2 : 3));
        lp.amp_ld = d->synth_amp_ld;
        lp.k2 = d->synth_k / (2.0 * 3.14159265358979323846264338327950288);
        lp.conj = 0;   // the inverse transform's conj-in is folded into the sign of k by the caller; synthesis is forward only
    }
}

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
static size_t blue2d_fused_ws(int dtype, int64_t M, int64_t N);

// The fold needs every input row stored (pairs (i, i + M/2) are combined), rotations by 0 or M/2 and an even output
// window.  It pays from 4096-point columns: the M/2-point column tiles leave room for two workgroups per CU (their
// load / butterfly / store phases overlap), twice the register budget per thread (complex128) and 64 B instead of 32 B
// pieces at 8192.  Measured (profiles/r01/tune_fold.log): 4096^2 complex64 101.9 -> 98.1 us, complex128 229 -> 216 us,
// 8192^2 complex64 557 -> 497 us, complex128 1143 -> 1047 us; 2048-point columns gain nothing (complex128 loses).
static bool fold_legal(const pm_fft2_desc* d, int logn, int logm) {
    const int64_t M = d->in_y.n;
    return logn >= 11 && logm >= 3 && d->in_y.off == 0 && d->in_y.len == M && (d->in_y.shift == 0 || d->in_y.shift == M / 2) &&
           (d->out_y.off % 2) == 0 && (d->out_y.len % 2) == 0 && (d->out_y.shift % 2) == 0 && d->mul_kind == PM_MUL_NONE &&
           d->batch <= 1 && (d->out_ld % 2) == 0;
}

// Hermitian path (fft_r2c.h): a FORWARD transform of an unpadded real field, both lengths on the engine (rows of at least 32
// samples), rotations by 0 or half a length, an output that keeps every bin, no multiplier, one field.
static bool r2c_legal(const pm_fft2_desc* d, int logn, int logm) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    if (!(d->flags & PM_FLAG_REAL_INPUT) || (d->flags & (PM_FLAG_SYNTH_INPUT | PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY))) return false;
    if (tuning().r2c == 0 || d->direction != -1 || logn < 5 || logm < 5 || d->batch > 1 || d->mul_kind != PM_MUL_NONE) return false;
    if (d->epilogue == PM_EPI_ABS2_ACCUM) return false;
    // Measured (profiles/r02/exp_r2c.log): with a real epilogue or the centre normalisation the Hermitian path beats transform +
    // elementwise sweeps at every size (fp32 MTF: 4096^2 90 vs 266 us, 2048^2 42 vs 64 us); a plain complex spectrum gains from
    // 4096^2 (87 vs 104 us, 8192^2 421 vs 449 us) and loses below (2048^2: 38 vs 31 us -- the extra exchange phases are pure
    // latency there), so small plain transforms stay on the complex path that only READS the real array (knob r2c = 2 forces it).
    if (tuning().r2c < 2 && d->epilogue == PM_EPI_NONE && !(d->flags & PM_FLAG_NORM_DC) && M * N < (int64_t(1) << 24)) return false;
    if (d->dtype == PM_C128 && logn > 12) return false;     // complex128 rows of 4096 complex points exchange re / im separately
    if (d->in_y.len != M || d->in_x.len != N || d->out_y.len != M || d->out_x.len != N) return false;
    if (!(d->in_x.shift == 0 || d->in_x.shift == N / 2) || (d->in_ld % 2) != 0) return false;
    return true;
}

// Transposed Hermitian form (fft_hermt.h): what r2c_legal accepts, with every rotation 0 or half a length (the input's become signs) and
// lengths the two kernels exist for.  Auto (knob herm_t < 0), from profiles/r06/exp_herm_rule.log (mtf_from_psf, us, round-2 form /
// transposed): fp32 128^2 20.5 / 14.8, 1024^2 29.9 / 19.1, 2048^2 38.8 / 34.2, 4096 x 1024 51.8 / 31.3, 4096^2 72.4 / 68.1 -- and
// 2048 x 8192 65.2 / 67.5, 4096 x 8192 123.5 / 170.7, 8192 rows 82.3 / 82.4 .. 139 / 172: rows of 8192 samples and columns of 8192 stay
// on the round-2 form; fp64 1024^2 29.4 / 21.6, 2048^2 41.9 / 36.3, 4096 x 2048 68.1 / 56.1 -- and 2048 x 4096 64.2 / 68.4, 4096^2
// 117.4 / 155.5: rows of 4096 complex128 points stay too.
static bool hermt_legal(const pm_fft2_desc* d, int logn, int logm) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    const int ht = tuning().herm_t;
    if (ht == 0) return false;
    if (logm < 5 || logm > 13 || logn < 5 || logn > (d->dtype == PM_C64 ? 13 : 12)) return false;
    if (logm == 13 && tuning().herm_t_fold == 0) return false;     // 8192-point columns exist as planes of 4096-point tiles only
    if (ht < 0 && (logm > 12 || logn > (d->dtype == PM_C64 ? 12 : 11))) return false;
    if (!(d->in_y.shift == 0 || d->in_y.shift == M / 2) || !(d->out_y.shift == 0 || d->out_y.shift == M / 2) ||
        !(d->out_x.shift == 0 || d->out_x.shift == N / 2))
        return false;
    return d->in_y.off == 0 && d->out_y.off == 0 && d->out_x.off == 0;
}

static int64_t batch_chunk(int64_t nb, size_t ws_field) {
    const size_t budget = size_t(tuning().batch_ws_mib) << 20;
    int64_t c = int64_t(budget / (ws_field ? ws_field : 1));
    if (c < 1) c = 1;
    return c < nb ? c : nb;
}

static Fft2Plan plan_fft2(const pm_fft2_desc* d, bool allow_r2c = true) {
    Fft2Plan p;
    const int64_t M = d->in_y.n, N = d->in_x.n;
    p.logn = engine_log2(N);
    p.logm = engine_log2(M);
    const size_t es = d->dtype == PM_C64 ? 8 : 16;
    const int64_t rows = d->in_y.len;   // only stored input rows are transformed in pass 1
    p.fold = false;
    p.w_ld = N;
    p.r2c = allow_r2c && p.logn >= 0 && p.logm >= 0 && r2c_legal(d, p.logn, p.logm);
    p.r2c_t = p.r2c && hermt_legal(d, p.logn, p.logm);
    if (p.r2c_t) {
        p.col_var = 0;
        p.tc = 0;
        p.log_k = 0;
        p.ws_bytes = size_t(M / 2) * size_t(N) * es;      // rows u < M/2 of the column spectra, row-major (row 0 carries u = 0 and u = M/2)
    } else if (p.r2c) {
        p.col_var = 0;
        p.tc = col_tile_width_for(d->dtype, p.logm, 0);
        // layout tiles of 8 column tiles: mtf_from_psf 4096^2 fp32 73.6 -> 72.0 us against 4 (profiles/r05/exp_layout_sweep.log)
        p.log_k = tuning().log_k >= 0 ? tuning().log_k : 3;
        // fold (one radix-2 step of the column transform in the row pass, as in the complex path): half-length column tiles, two
        // workgroups per CU whose load / transform / store phases overlap -- here from 1024-point columns, because the Hermitian
        // column pass has only half the tiles to fill the chip with
        const int f = tuning().fold;
        p.fold = (f > 0 || (f < 0 && p.logm >= 10)) && p.logm >= 5 && (d->in_y.shift == 0 || d->in_y.shift == M / 2) &&
                 (d->out_y.shift == 0 || d->out_y.shift == M / 2);
        if (p.fold) p.tc = col_tile_width_for(d->dtype, p.logm - 1, 0);
        const int tlog = p.fold ? p.logm - 1 : p.logm;
        if (tuning().herm_wide && tlog == 11 && (N / 2) % col_tile_width_for(d->dtype, 11, 2) == 0) {
            p.col_var = 2;
            p.tc = col_tile_width_for(d->dtype, 11, 2);
        }
        while (p.log_k > 0 && ((N / 2) % (int64_t(p.tc) << p.log_k)) != 0) --p.log_k;
        const int64_t nc = N / 2, tl = int64_t(p.tc) << p.log_k;
        p.ws_bytes = size_t((nc + tl - 1) / tl) * size_t(M) * size_t(tl) * es;
    } else if (p.logn >= 0 && p.logm >= 0) {
        if (fold_legal(d, p.logn, p.logm)) {
            const int f = tuning().fold;
            p.fold = f > 0 || (f < 0 && p.logm >= 12);
        }
        {   // 128 B tiles exist for 2048-point complex128 tiles only (fft_kernels.h launch_fft): the knob can switch them off or, for an
            // unfolded 2048-row transform, on -- nothing else
            const bool want2 = tuning().col_var >= 0 ? tuning().col_var == 2 : (p.fold && p.logm == 12);
            p.col_var = (want2 && d->dtype == PM_C128 && (p.fold ? p.logm - 1 : p.logm) == 11) ? 2 : 0;
        }
        p.tc = col_tile_width_for(d->dtype, p.fold ? p.logm - 1 : p.logm, p.col_var);
        p.log_k = tuning().log_k >= 0 ? tuning().log_k : (N >= 8192 ? 3 : (N >= 4096 ? 2 : 1));   // auto: >= 256 B pieces from 4096 columns
        // folded 4096^2 complex64 (intermediate = 128 MiB, inside the Infinity Cache): 8 KiB row pieces measured 95.0 vs 97.8 us
        // (profiles/r01/tune_log_k.log); every other size / precision measured best with the narrow tiles above
        if (tuning().log_k < 0 && p.fold && d->dtype == PM_C64 && N == 4096 && M == 4096) p.log_k = 7;
        while (p.log_k > 0 && (N % (int64_t(p.tc) << p.log_k)) != 0) --p.log_k;
        const int64_t tl = int64_t(p.tc) << p.log_k;
        const int64_t ntl = (N + tl - 1) / tl;
        p.ws_bytes = size_t(ntl) * size_t(rows) * size_t(tl) * es;
    } else {
        p.tc = 0;
        p.log_k = 0;
        if (p.logm < 0 && use_mix(M)) {
            const int64_t line = int64_t(128 / es);
            p.w_ld = (N + line - 1) / line * line;
        }
        p.ws_bytes = size_t(rows) * size_t(p.w_ld) * es;
    }
    if (p.ws_bytes == 0) p.ws_bytes = es;
    p.nbatch = d->batch > 1 ? d->batch : 1;
    p.ws_field = (p.ws_bytes + 255) & ~size_t(255);
    p.chunk = batch_chunk(p.nbatch, p.ws_field);
    if (p.nbatch > 1) p.ws_bytes = p.ws_field * size_t(p.chunk);
    // powers of two above the engine's longest transform: both axes powers of two, at least one split (big2d_run)
    p.big_rn = big_split2d(N);
    p.big_rm = big_split2d(M);
    if (p.big_rn > 1 && !p.big_rm) p.big_rm = big_split2d(M, false);     // a composite length beside one that needs the split: both take it
    if (p.big_rm > 1 && !p.big_rn) p.big_rn = big_split2d(N, false);
    if (p.big_rn && p.big_rm) {     // sub-transforms on the mixed-radix kernel address with 32-bit offsets
        const int64_t np_ = N / p.big_rn, mp_ = M / p.big_rm;
        const bool mixn = engine_log2(np_) < 0, mixm = engine_log2(mp_) < 0;
        if ((mixn && !(mix_fits(np_, np_, es, false) && (p.big_rn > 1 || mix_fits(np_, d->in_ld, es, false)))) ||
            (mixm && !mix_fits(mp_, int64_t(p.big_rm) * np_, es, true)))
            p.big_rn = p.big_rm = 0;
    }
    if (!(p.big_rn && p.big_rm && (p.big_rn > 1 || p.big_rm > 1)) || (d->flags & (PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY | PM_FLAG_SYNTH_INPUT)))
        p.big_rn = p.big_rm = 0;
    p.mix_n = p.mix_m = p.mix_fold = false;
    if (p.big_rn) {   // [Z: R_n planes of M x N/R_n | F (and the pre-processed rows before it): the same size]
        p.tc = 0;
        p.fold = false;
        p.blue_n = p.blue_m = p.blue2d = p.blue_big = false;
        p.blue_off = 0;
        const size_t arr = (size_t(M) * size_t(N) * es + 255) & ~size_t(255);
        p.ws_bytes = 2 * arr;
        return p;
    }
    // the mixed-radix kernel addresses with 32-bit offsets: arrays of 4 GiB and more plan without it
    const bool mixfit = mix_fits(N, d->in_ld, es, false) && mix_fits(M, p.w_ld, es, true) && mix_fits(M, d->out_ld, es, true);
    p.mix_n = mixfit && p.logn < 0 && use_mix(N);
    p.mix_m = mixfit && p.logm < 0 && use_mix(M);
    // FOLD on composite grids (round 4; knob mix_fold, experiment builds only -- measured slower, fft_mixed.h MixRowOut): where four columns of M points fill a CU's LDS the column kernel runs one workgroup per CU, whose
    // load / butterfly / store phases nothing overlaps (fft_mixed_kernels.h).  With the radix-2 step of the column transform taken by the
    // row pass (rows in pairs (g, g + M/2)) the column tiles are half as tall and two or three workgroups share a CU.  Needs every row
    // stored, rotations of 0 or M/2 on the way in and an even one on the way out, no multiplier, one field.
    {
        const int64_t H = M / 2;
        // the tile the unfolded column pass would take (mix_cols_impl: four columns, eight of mid-size complex64, fewer when they do not fit)
        const size_t per = size_t(M) * es, hard = size_t(156) * 1024;
        size_t tc0 = (es == 8 && per > size_t(10) * 1024 && 8 * per <= hard) ? 8 : 4;
        while (tc0 > 1 && tc0 * per > hard) tc0 /= 2;
        p.mix_fold = tuning().mix_fold && p.mix_n && p.mix_m && (M % 2) == 0 && use_mix(H) && tc0 * per > size_t(80) * 1024 &&
                     d->in_y.len == M && d->in_y.off == 0 && (d->in_y.shift == 0 || d->in_y.shift == H) && d->out_y.len == M && d->out_y.off == 0 &&
                     (d->out_y.shift % 2) == 0 && d->mul_kind == PM_MUL_NONE &&
                     !(d->flags & (PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY)) && mix_fits(N, H * d->in_ld, es, false) &&
                     (!(d->flags & PM_FLAG_SYNTH_INPUT) || !d->synth_amp || mix_fits(N, H * d->synth_amp_ld, es, false));
    }
    p.blue_n = p.logn < 0 && use_blue(N, mixfit);
    p.blue_m = p.logm < 0 && use_blue(M, mixfit);
    p.blue_off = (p.ws_bytes + 255) & ~size_t(255);
    const bool noflags = !(d->flags & (PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY | PM_FLAG_SYNTH_INPUT));
    p.blue2d = p.blue_n && p.blue_m && tuning().blue_2d && noflags;
    p.blue_big = !p.blue2d && tuning().blue_2d && noflags && blue_reach(N) && blue_reach(M) && (blue_needs_both(N, mixfit) || blue_needs_both(M, mixfit)) &&
                 (big_split(blue_conv_len(N)) > 1 || big_split(blue_conv_len(M)) > 1);
    if (p.blue_big) {   // [a (M x N) | c (M x N) | spectrum (MB1 x MB2) | workspace of the big transforms]
        p.blue2d = true;
        p.blue_n = p.blue_m = false;
        const size_t arr = (size_t(M) * size_t(N) * es + 255) & ~size_t(255);
        const int64_t mb1 = blue_conv_len(M), mb2 = blue_conv_len(N);
        const size_t spec = (size_t(mb1) * size_t(mb2) * es + 255) & ~size_t(255);
        p.ws_bytes = 2 * arr + spec + 2 * spec;
    } else if (p.blue2d) {
        const size_t arr = (size_t(M) * size_t(N) * es + 255) & ~size_t(255);
        p.ws_bytes = 2 * arr + blue2d_fused_ws(d->dtype, M, N);
    } else if (p.blue_n || p.blue_m) {
        const size_t a = p.blue_n ? blue_rows_scratch(es, rows, N) : 0, b = p.blue_m ? blue_cols_scratch(es, N, M) : 0;
        p.ws_bytes = p.blue_off + (a > b ? a : b);
    }
    return p;
}

template <typename T>
static ColStoreNat<T> make_colstore(const pm_fft2_desc* d, void* out, int logm_tile = -1) {
    ColStoreNat<T> cs{};
    cs.dst = out;
    cs.ld = d->out_ld;
    cs.ay = to_map(d->out_y);
    cs.ax = to_map(d->out_x);
    cs.conj = d->direction > 0 ? 1 : 0;
    cs.epilogue = d->epilogue;
    cs.scale = T(d->scale);
    cs.weight = T(d->weight);
    cs.mul_kind = d->mul_kind;
    cs.mul_conj = d->mul_conj;
    cs.mul = reinterpret_cast<const cx<T>*>(d->mul);
    cs.mul_x = reinterpret_cast<const cx<T>*>(d->mul_x);
    cs.mul_ld = d->mul_ld;
    cs.bstride = d->out_bstride;
    cs.mul_bstride = d->mul_bstride;
    cs.mul_bstride_x = d->mul_x_bstride;
    bool vec = true;
    if (d->epilogue == PM_EPI_NONE && sizeof(T) == 4)
        vec = (d->out_ld % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0) && (d->out_bstride % 2 == 0);
    cs.vec_ok = vec ? 1 : 0;
    // bit 1: pairs of REAL outputs (the |.|^2 epilogues of the two-column complex64 threads) may go out as 8-byte accesses
    if (d->epilogue != PM_EPI_NONE && (d->out_ld % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % (2 * sizeof(T)) == 0) &&
        (d->out_bstride % 2 == 0))
        cs.vec_ok |= 2;
    const size_t out_bytes = size_t(d->batch > 1 ? d->batch : 1) * size_t(d->out_y.len) * size_t(d->out_x.len) *
                             (d->epilogue == PM_EPI_NONE ? sizeof(cx<T>) : sizeof(T));
    // streaming stores only help when a workgroup writes whole 64 B pieces; on the 32 B pieces of 8192-point
    // columns they defeat the L2 write combining of sibling workgroups (measured: 977 -> 428 us without)
    if (logm_tile < 0) logm_tile = engine_log2(d->out_y.n) >= 0 ? engine_log2(d->out_y.n) : 12;
    const size_t piece = size_t(col_tile_width_for(d->dtype, logm_tile, 0)) *
                         (d->epilogue == PM_EPI_NONE ? sizeof(cx<T>) : sizeof(T));
    // ... and only while the output is about the size of the 256 MiB Infinity Cache: measured +25 % at 256 MiB (4096^2
    // complex128), -10 % at 512 MiB and 1 GiB (8192^2), -8 % at 128 MiB
    cs.nt = tuning().nt_out >= 0 ? tuning().nt_out
                                 : ((out_bytes >= (size_t(192) << 20) && out_bytes < (size_t(384) << 20) && piece >= 64) ? 1 : 0);
    return cs;
}

template <typename T>
static int blue2d_run(const pm_fft2_desc* d, const Fft2Plan& p, const void* in, void* out, void* ws, hipStream_t st);
template <typename T>
static int big2d_run(const pm_fft2_desc* d, const Fft2Plan& p, const void* in, void* out, void* ws, hipStream_t st);

// one launch pair over `nb` fields (nb > 1 only when both passes run on the engine)
template <typename T>
static int fft2_run_chunk(const pm_fft2_desc* d, const Fft2Plan& p, const void* in, void* out, void* ws, hipStream_t st, int nb) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    const int64_t wstride = int64_t(p.ws_field / sizeof(cx<T>));
    const int rows = int(d->in_y.len);
    const int conj = d->direction > 0 ? 1 : 0;
    int err = 0;
    cx<T>* W = reinterpret_cast<cx<T>*>(ws);
    const bool run1 = !(d->flags & PM_FLAG_PASS2_ONLY), run2 = !(d->flags & PM_FLAG_PASS1_ONLY);
    if (p.big_rn) return big2d_run<T>(d, p, in, out, ws, st);
    if (p.blue2d) return blue2d_run<T>(d, p, in, out, ws, st);
    if (p.r2c_t) {
        // pass A: M-point transforms down the N/2 packed columns of the real array, separated into the N column spectra (rows u < M/2)
        const cx<T>* twm = twiddles<T>(M, &err);
        if (!twm) return err;
        const cx<T>* twn = twiddles<T>(N, &err);
        if (!twn) return err;
        const int64_t n2 = N / 2;
        const int tc = col_tile_width_for(d->dtype, p.logm, 0);
        const int ntiles = int((n2 + tc - 1) / tc);
        const int64_t ld2 = d->in_ld / 2;
        ColLoadNat<T> cl{reinterpret_cast<const cx<T>*>(in), ld2, AxisMap{int(M), int(M), 0, 0}, int(n2), 0,
                         (ld2 % 2 == 0 && reinterpret_cast<uintptr_t>(in) % 16 == 0) ? 1 : 0, 0};
        // fold (a radix-2 step of the column transform in the load, planes of M/2-point tiles: two workgroups per CU): from 4096 rows
        const int hf = tuning().herm_t_fold;
        const bool fold = (hf > 0 && p.logm >= 11) || (hf < 0 && p.logm >= 12);
        HermTColStore<T> cs{W, N, int(n2), d->in_y.shift == M / 2 ? 1 : 0, fold ? 1 : 0, twm, 0};
        const cx<T>* twa = twm;
        int tiles = ntiles;
        if (fold) {
            twa = twiddles<T>(M / 2, &err);
            if (!twa) return err;
            const int tcf = col_tile_width_for(d->dtype, p.logm - 1, 0);
            tiles = int((n2 + tcf - 1) / tcf);
        }
        cs.ntiles = tiles;
        // adjacent tiles read the two halves of the input's 128 B lines and write adjacent lines of the intermediate: siblings on one XCD
        // (profiles/r06/exp_herm_t_log_g.log, mtf_from_psf us at col_log_g 0 .. 5: 4096^2 fp32 folded 75.9 68.5 69.7 69.3 67.9 68.1 -- the pass reads
        // 64 B pieces of a row-major array, neighbours share its 128 B lines --; 2048^2 (128 tiles, half the CUs) 31.9 33.6 33.7 33.8 34.2 34.4)
        int lg = tuning().col_log_g >= 0 ? tuning().col_log_g : (p.logm >= 12 ? 4 : 0);
        while (lg > 0 && ((fold ? 2 * tiles : tiles) % (8 << (lg + (fold ? 1 : 0)))) != 0) --lg;
        int rc = launch_col_hermt<T>(p.logm, cl, cs, twa, tiles, lg, st);
        if (rc) return rc < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2: internal: no transposed Hermitian column kernel for %lld points", (long long)M) : rc;
        // pass B: N-point transforms of the M/2 rows, each stored with its mirror image
        RowLoadNat<T> lp{W, N, AxisMap{int(N), int(N), 0, 0}, int(M / 2), 0, 0, 0};
        HermTRowStore<T> rs{out, d->out_ld, int(M), int(N), int(d->out_y.shift), int(d->out_x.shift), d->epilogue, T(d->scale),
                            (d->flags & PM_FLAG_NORM_DC) ? 1 : 0, W, d->in_x.shift == N / 2 ? 1 : 0, int(M / 2)};
        rc = launch_row_hermt<T>(p.logn, tuning().herm_t_rowvar >= 0 ? tuning().herm_t_rowvar : row_variant(d->dtype, p.logn), lp, rs, twn, tuning().row_log_g, st);
        if (rc) return rc < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2: internal: no transposed Hermitian row kernel for %lld points", (long long)N) : rc;
        return 0;
    }
    if (p.r2c) {
        // rows: the real array read as N/2 complex points per row -> N/2 columns of the tiled intermediate (column 0 = X[0] + i X[N/2])
        const int64_t n2 = N / 2, tl = int64_t(p.tc) << p.log_k;
        int ltl = 0;
        while ((int64_t(1) << ltl) < tl) ++ltl;
        const cx<T>* tw2 = twiddles<T>(n2, &err);
        if (!tw2) return err;
        const cx<T>* twn = twiddles<T>(N, &err);
        if (!twn) return err;
        const cx<T>* twm = twiddles<T>(M, &err);
        if (!twm) return err;
        const size_t in_bytes = size_t(M) * size_t(N) * sizeof(T);
        RowLoadNat<T> lp{reinterpret_cast<const cx<T>*>(in), d->in_ld / 2, AxisMap{int(n2), int(n2), 0, int(d->in_x.shift / 2)}, int(M), 0,
                         tuning().nt_in >= 0 ? tuning().nt_in : (in_bytes >= (size_t(96) << 20) ? 1 : 0), 0};
        const int H = int(M / 2);
        const int64_t ntl = (n2 + tl - 1) / tl, plane = ntl * H * tl;
        R2CRowStore<T> rs{W, int(M), ltl, twn, 0, 0, nullptr, 0};
        if (p.fold) {
            lp.eoff = H;
            rs.nseq = H;
            rs.fold = 1;
            rs.plane_stride = plane;
            rs.twm = twm;
            rs.swap = d->in_y.shift == M / 2 ? 1 : 0;
        }
        int rc = launch_row_r2c<T>(p.logn - 1, lp, rs, tw2, p.fold ? H : int(M), p.fold ? 0 : tuning().row_log_g, st);
        if (rc) return rc < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2: internal: no Hermitian row kernel for %lld points", (long long)N) : rc;
        // columns: M-point transforms of the N/2 columns, each bin stored at (u, k) and conjugated at (-u, -k)
        const int ntiles = int((n2 + p.tc - 1) / p.tc);
        const size_t oes = d->epilogue == PM_EPI_NONE ? sizeof(cx<T>) : sizeof(T);
        const int fast = ((d->out_y.shift == 0 || d->out_y.shift == M / 2) && (d->out_x.shift == 0 || d->out_x.shift == N / 2) &&
                          (N % (2 * p.tc)) == 0 && (d->out_ld % 2) == 0 && reinterpret_cast<uintptr_t>(out) % (2 * oes) == 0) ? 1 : 0;
        if (p.fold) {
            // two planes of M/2-point column transforms; plane b holds the bins 2 u' + b = output rows of that parity: the output is
            // seen with a doubled leading dimension, plane 1 one row further
            const cx<T>* twh = twiddles<T>(H, &err);
            if (!twh) return err;
            ColLoadTiled<T> cl{W, H, AxisMap{H, H, 0, 0}, ntiles, p.log_k, plane};
            HermStore<T> hs{out, 2 * d->out_ld, AxisMap{H, H, 0, int(d->out_y.shift / 2)}, to_map(d->out_x), H, int(N), d->epilogue,
                            T(d->scale), T(d->weight), (d->flags & PM_FLAG_NORM_DC) ? 1 : 0, W, tl, H, 0, d->out_ld, fast, p.col_var == 2 ? 1 : 0};
            // several rounds of one-workgroup-per-CU tiles (8192^2: 1024 of them): ALL 32 workgroups an XCD holds take adjacent tiles,
            // so a row of the output is written 2 KiB at a time -- mtf_from_psf 8192^2 fp32 391 -> 373 us (exp_layout_sweep.log)
            int lg = sibling_log_g(p.log_k);
            if (tuning().col_log_g < 0 && 2 * ntiles > 2 * pm_num_cus())
                for (lg = 5; lg > 3 && ntiles % (8 << lg); --lg) {}
            return launch_col_herm<T>(p.logm - 1, cl, hs, twh, ntiles, lg, st);
        }
        ColLoadTiled<T> cl{W, int(M), to_map(d->in_y), ntiles, p.log_k, 0};
        HermStore<T> hs{out, d->out_ld, to_map(d->out_y), to_map(d->out_x), int(M), int(N), d->epilogue, T(d->scale), T(d->weight),
                        (d->flags & PM_FLAG_NORM_DC) ? 1 : 0, W, tl, int(M), -1, 0, fast, p.col_var == 2 ? 1 : 0};
        return launch_col_herm<T>(p.logm, cl, hs, twm, ntiles, sibling_log_g(p.log_k), st);
    }

    // ---- pass 1: one transform of length N per STORED input row (all-zero padded rows are skipped)
    if (run1 && rows > 0) {
        if (p.logn >= 0) {
            const cx<T>* tw = twiddles<T>(N, &err);
            if (!tw) return err;
            const size_t in_bytes = size_t(p.nbatch) * size_t(rows) * size_t(d->in_x.len) * sizeof(cx<T>);
            const int nt_in = tuning().nt_in >= 0 ? tuning().nt_in : (in_bytes >= (size_t(96) << 20) ? 1 : 0);
            RowLoadNat<T> lp{reinterpret_cast<const cx<T>*>(in), d->in_ld, to_map(d->in_x), rows, conj, nt_in, d->in_bstride};
            set_input_mode(lp, d);
            int rc;
            if (p.fold) {
                int ltc = 0;
                while ((1 << ltc) < (p.tc << p.log_k)) ++ltc;
                const cx<T>* twm = twiddles<T>(M, &err);
                if (!twm) return err;
                lp.eoff = int(M / 2);
                const int64_t tl = int64_t(1) << ltc, ntl = (N + tl - 1) / tl;
                RowStoreFold<T> sp{W, ntl * (M / 2) * tl, int(M / 2), ltc, twm, d->in_y.shift == M / 2 ? 1 : 0, 0};
                rc = launch_row_fold<T>(p.logn, lp, sp, tw, int(M / 2), 0, st, 1);   // pairs are not siblings: no XCD grouping
            } else if (p.tc) {
                int ltc = 0;
                while ((1 << ltc) < (p.tc << p.log_k)) ++ltc;
                RowStoreTiled<T> sp{W, rows, ltc, wstride};
                rc = launch_row_tiled<T>(p.logn, row_variant(d->dtype, p.logn), lp, sp, tw, rows, tuning().row_log_g, st, nb);
            } else {
                RowStoreNat<T> sp{W, p.w_ld, AxisMap{int(N), int(N), 0, 0}, rows, 0, T(1), 0, AxisMap{1, 1, 0, 0}};
                rc = launch_row_nat<T>(p.logn, row_variant(d->dtype, p.logn), lp, sp, tw, rows, 0, st);
            }
            if (rc) return rc;
        } else {
            DirectIn<T> di{reinterpret_cast<const cx<T>*>(in), d->in_ld, 1, to_map(d->in_x), rows, conj,
                           (d->flags & PM_FLAG_REAL_INPUT) ? 1 : 0};
            if (d->flags & PM_FLAG_SYNTH_INPUT) {     // the mixed-radix row kernel synthesises the pupil in its first stage's loads
                if (!p.mix_n)
                    return fail(PM_ERR_UNSUPPORTED, "pm_fft2: PM_FLAG_SYNTH_INPUT: rows of %lld samples do not run on a kernel that synthesises the "
                                "pupil while loading (synthesise it with pm_pupil_synth first)", (long long)N);
                di.synth = (d->flags & PM_FLAG_SYNTH_PACKED) ? 3 : 2;
                di.real = 0;
                di.conj = 0;
                di.amp = d->synth_amp;
                di.amp_kind = !d->synth_amp ? 0 : (d->synth_amp_dtype == PM_F32 ? 1 : (d->synth_amp_dtype == PM_F64 ? 2 : 3));
                di.amp_ld = d->synth_amp_ld;
                di.k2 = d->synth_k / (2.0 * 3.14159265358979323846264338327950288);
                if (di.amp && !mix_fits(N, di.amp_ld, sizeof(cx<T>), false))
                    return fail(PM_ERR_UNSUPPORTED, "pm_fft2: PM_FLAG_SYNTH_INPUT: amplitude pitch beyond 2^24 elements");
            }
            int rc;
            if (p.mix_n && p.mix_fold) {
                const cx<T>* twm = twiddles<T>(M, &err);
                if (!twm) return err;
                const MixFold<T> mf{int(M / 2), d->in_y.shift == M / 2 ? 1 : 0, twm};
                rc = mix_rows<T>(di, W, p.w_ld, st, nullptr, &mf);
            } else if (p.mix_n) {
                di.nb = nb;                 // a stack (fft2_run: only where both passes are mixed-radix and the view is plain)
                di.bstride = d->in_bstride;
                rc = mix_rows<T>(di, W, p.w_ld, st, nullptr, nullptr, wstride);
            } else if (p.blue_n) {
                rc = blue_rows<T>(di, W, p.w_ld, static_cast<char*>(ws) + p.blue_off, st);
            } else {
                const cx<double>* tw = twiddles_f64(N, &err);
                if (!tw) return err;
                rc = direct_rows<T>(di, W, p.w_ld, tw, st);
            }
            if (rc) return rc;
        }
    }
    if (!run2) return 0;

    // ---- pass 2: transforms of length M down the columns, epilogue fused into the store
    ColStoreNat<T> cs = make_colstore<T>(d, out, p.fold ? p.logm - 1 : -1);
    if (p.fold) {
        // two planes of M/2-point column transforms: plane b holds output rows 2k + b -> output view with doubled
        // leading dimension, plane b offset by one row (the batch stride of the store)
        const cx<T>* tw = twiddles<T>(M / 2, &err);
        if (!tw) return err;
        const int ntiles = int((N + p.tc - 1) / p.tc);
        const int64_t tl = int64_t(p.tc) << p.log_k, ntl = (N + tl - 1) / tl;
        const int H = int(M / 2);
        ColLoadTiled<T> cl{W, H, AxisMap{H, H, 0, 0}, ntiles, p.log_k, ntl * H * tl};
        cs.ay = AxisMap{H, int(d->out_y.len / 2), int(d->out_y.off / 2), int(d->out_y.shift / 2)};
        cs.bstride = d->out_ld;
        cs.ld = 2 * d->out_ld;
        return launch_col_tiled<T>(p.logm - 1, p.col_var, cl, cs, tw, ntiles, sibling_log_g(p.log_k), st, 2);
    }
    if (p.logm >= 0) {
        const cx<T>* tw = twiddles<T>(M, &err);
        if (!tw) return err;
        if (p.tc) {
            const int ntiles = int((N + p.tc - 1) / p.tc);
            ColLoadTiled<T> cl{W, rows, to_map(d->in_y), ntiles, p.log_k, wstride};
            return launch_col_tiled<T>(p.logm, p.col_var, cl, cs, tw, ntiles, sibling_log_g(p.log_k), st, nb);
        }
        const int tc = col_tile_width_for(d->dtype, p.logm, 0);
        const int ntiles = int((N + tc - 1) / tc);
        ColLoadNat<T> cl{W, p.w_ld, to_map(d->in_y), int(N), 0, (p.w_ld % 2 == 0) ? 1 : 0};
        return launch_col_nat<T>(p.logm, 0, cl, cs, tw, ntiles, 1, st);
    }
    if (p.mix_m && p.mix_fold) {
        // two planes of M/2-point column transforms: plane b holds the output rows 2 k + b -- the output seen with a doubled leading
        // dimension, plane 1 one row further (as the engine's fold above)
        const int H = int(M / 2);
        const size_t oes = d->epilogue == PM_EPI_NONE ? sizeof(cx<T>) : sizeof(T);
        for (int b = 0; b < 2; ++b) {
            DirectIn<T> dp{W + int64_t(b) * H * p.w_ld, 1, p.w_ld, AxisMap{H, H, 0, 0}, int(N), 0};
            ColStoreNat<T> cp = cs;
            cp.dst = static_cast<char*>(cs.dst) + size_t(b) * size_t(d->out_ld) * oes;
            cp.ld = 2 * d->out_ld;
            cp.ay = AxisMap{H, H, 0, int(d->out_y.shift / 2)};
            const int rc = mix_cols<T>(dp, cp, st);
            if (rc) return rc;
        }
        return 0;
    }
    DirectIn<T> di{W, 1, p.w_ld, to_map(d->in_y), int(N), 0};   // sequence = column c at W[c], element stride = the pitch of the intermediate
    if (p.mix_m) {
        di.nb = nb;
        di.bstride = wstride;
        return mix_cols<T>(di, cs, st);
    }
    if (p.blue_m) return blue_cols<T>(di, cs, static_cast<char*>(ws) + p.blue_off, st);
    const cx<double>* tw = twiddles_f64(M, &err);
    if (!tw) return err;
    return direct_cols<T>(di, cs, tw, st);
}

static const void* offset_elems(const void* p, int64_t elems, size_t es) {
    return p ? static_cast<const void*>(static_cast<const char*>(p) + elems * int64_t(es)) : nullptr;
}

// Batch driver: chunks of fields whose intermediates fit the Infinity Cache go out as one launch pair each
// (grid.y = fields); sizes that need the direct-DFT kernels run field by field.
template <typename T>
static int fft2_run(const pm_fft2_desc* d, const Fft2Plan& p, const void* in, void* out, void* ws, hipStream_t st) {
    if (p.nbatch <= 1) return fft2_run_chunk<T>(d, p, in, out, ws, st, 1);
    const bool engine = p.tc != 0;
    // ... and composite grids whose two passes both run on the composite register engine (fft_ce.h: grid.y = fields; round 5)
    const bool f32 = d->dtype == PM_C64;
    const bool ce_stack = p.mix_n && p.mix_m && !p.mix_fold && !p.big_rn && !p.blue2d && tuning().mix_engine &&
                          !(d->flags & (PM_FLAG_REAL_INPUT | PM_FLAG_SYNTH_INPUT | PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY)) && d->mul_kind == PM_MUL_NONE &&
                          d->out_y.off == 0 && d->out_y.len == d->out_y.n && d->out_x.off == 0 && d->out_x.len == d->out_x.n &&
                          d->epilogue <= PM_EPI_ABS2_ACCUM &&
                          (f32 ? ce_has_plan<float>(int(d->in_x.n)) && ce_has_plan<float>(int(d->in_y.n))
                               : ce_has_plan<double>(int(d->in_x.n)) && ce_has_plan<double>(int(d->in_y.n)));
    const int64_t step = (engine || ce_stack) ? p.chunk : 1;
    const size_t oes = d->epilogue == PM_EPI_NONE ? sizeof(cx<T>) : sizeof(T);
    for (int64_t b0 = 0; b0 < p.nbatch; b0 += step) {
        const int nb = int(p.nbatch - b0 < step ? p.nbatch - b0 : step);
        pm_fft2_desc dd = *d;
        dd.mul = offset_elems(d->mul, b0 * d->mul_bstride, sizeof(cx<T>));
        dd.mul_x = offset_elems(d->mul_x, b0 * d->mul_x_bstride, sizeof(cx<T>));
        const void* inb = offset_elems(in, b0 * d->in_bstride, (d->flags & PM_FLAG_REAL_INPUT) ? sizeof(T) : sizeof(cx<T>));
        void* outb = const_cast<void*>(offset_elems(out, b0 * d->out_bstride, oes));
        int rc = fft2_run_chunk<T>(&dd, p, inb, outb, ws, st, nb);
        if (rc) return rc;
    }
    return 0;
}

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

// Composite grids (round 4): the column length runs on its own factors with the column resident in LDS through forward stages, multiplier
// and transposed stages (mix_cols_mul); the row passes are the engine's (a power-of-two row length) or the mixed-radix row kernel, on
// NATURAL intermediates.  Three passes / 6 N^2 s bytes where two pm_fft2 calls move 8 N^2 s -- the reference takes any size through one
// code path (prysm/propagation/angular_spectrum.py:9-42, prysm/convolution.py:9-31).  One field, complex output.
static bool plan_fused_mix(const pm_fft2_desc* d, FusedPlan& p) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    const size_t es = d->dtype == PM_C64 ? 8 : 16;
    if (!tuning().mix_fused || !use_mix(M) || !(p.logn >= 0 || use_mix(N)) || d->batch > 1) return false;
    if (d->flags & (PM_FLAG_SYNTH_INPUT | PM_FLAG_SYNTH_PACKED | PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY | PM_FLAG_NORM_DC)) return false;
    if (d->epilogue != PM_EPI_NONE || d->in_y.len <= 0) return false;
    const int64_t line = int64_t(128 / es);
    p.w_ld = (N + line - 1) / line * line;
    if (!mix_fits(N, d->in_ld, es, false) || !mix_fits(M, p.w_ld, es, true) || !mix_fits(N, d->out_ld, es, false) || !mix_fits(N, p.w_ld, es, false))
        return false;
    {   // pass A is the first pass of pm_fft2 on this shape (fused_mix_run): its planner must take the same natural intermediate, or the
        // query below would promise a chain the run then refuses
        pm_fft2_desc da = *d;
        da.flags = (d->flags & PM_FLAG_REAL_INPUT) | PM_FLAG_PASS1_ONLY;
        da.mul_kind = PM_MUL_NONE;
        da.direction = -1;
        da.batch = 0;
        const Fft2Plan pa = plan_fft2(&da);
        if (pa.tc != 0 || pa.w_ld != p.w_ld || pa.blue_n || pa.blue2d || pa.big_rn) return false;
    }
    p.mixmid = true;
    p.fold = false;
    p.tc = 0;
    p.log_k = 0;
    p.inplace = d->in_y.len == M;
    p.w1_bytes = (size_t(d->in_y.len) * size_t(p.w_ld) * es + 255) & ~size_t(255);
    p.w2_bytes = p.inplace ? 0 : ((size_t(M) * size_t(p.w_ld) * es + 255) & ~size_t(255));
    p.nbatch = p.chunk = 1;
    p.ws_bytes = p.w1_bytes + p.w2_bytes;
    return true;
}

static bool plan_fused(const pm_fft2_desc* d, FusedPlan& p) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    p.logn = engine_log2(N);
    p.logm = engine_log2(M);
    p.mixmid = false;
    p.w_ld = N;
    if (p.logm < 0) return plan_fused_mix(d, p);
    if (p.logn < 0) return false;
    const size_t es = d->dtype == PM_C64 ? 8 : 16;
    // fold (see fold_legal): here the output window is unconstrained -- the last row pass rebuilds whole rows
    p.fold = false;
    if (p.logn >= 11 && p.logm >= 3 && d->in_y.off == 0 && d->in_y.len == M && (d->in_y.shift == 0 || d->in_y.shift == M / 2) &&
        d->batch <= 1) {
        const int f = tuning().fold;
        p.fold = f > 0 || (f < 0 && p.logm >= 12);
    }
    p.tc = col_tile_width_for(d->dtype, p.fold ? p.logm - 1 : p.logm, 0);
    p.log_k = tuning().log_k >= 0 ? tuning().log_k : (N >= 8192 ? 3 : (N >= 4096 ? 2 : 1));
    // folded 4096^2 complex64: 16 KiB row pieces measured 182.7 vs 188.7 us for the chain (see plan_fft2)
    if (tuning().log_k < 0 && p.fold && d->dtype == PM_C64 && N == 4096 && M == 4096) p.log_k = 8;
    while (p.log_k > 0 && (N % (int64_t(p.tc) << p.log_k)) != 0) --p.log_k;
    const int64_t tl = int64_t(p.tc) << p.log_k, ntl = (N + tl - 1) / tl;
    p.inplace = d->in_y.len == M;
    p.w1_bytes = size_t(ntl) * size_t(d->in_y.len > 0 ? d->in_y.len : 1) * size_t(tl) * es;
    p.w2_bytes = p.inplace ? 0 : size_t(ntl) * size_t(M) * size_t(tl) * es;
    p.w1_bytes = (p.w1_bytes + 255) & ~size_t(255);
    p.w2_bytes = (p.w2_bytes + 255) & ~size_t(255);
    p.nbatch = d->batch > 1 ? d->batch : 1;
    p.chunk = batch_chunk(p.nbatch, p.w1_bytes + p.w2_bytes);
    p.ws_bytes = (p.w1_bytes + p.w2_bytes) * size_t(p.chunk);
    return true;
}

// Last row pass of the fused chains: streaming (non-temporal) stores -- the output is written once, in whole rows, and every line it does
// not leave in the caches is a line of the intermediate that stays.  Measured (profiles/r03/exp_nt_rows.log, chain us without / with):
// 4096^2 complex128 (256 MiB out) 344-347 / 320-321, padded 2048^2 -> 4096^2 complex128 343 / 326, 4096^2 complex64 (128 MiB) 167.7 /
// 163.3, 2048^2 complex64 54.6 / 52.7, 2048^2 complex128 83.0 / 82.8.  Not beyond the Infinity Cache's size class (the two-pass
// transform's column store lost 10 % with streaming stores at 512 MiB and 1 GiB, make_colstore).
static int row_store_nt(size_t out_bytes) {
    return tuning().nt_out >= 0 ? tuning().nt_out : ((out_bytes >= (size_t(24) << 20) && out_bytes < (size_t(384) << 20)) ? 1 : 0);
}

template <typename T>
static int fused_run_chunk(const pm_fft2_desc* d, const FusedPlan& p, const void* in, void* out, void* ws, hipStream_t st, int nb) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    const int rows = int(d->in_y.len);
    int err = 0;
    // per field: W1 at ws + b*(w1 + w2), W2 right behind it (or the same block when the transform is in place)
    const int64_t wstride = int64_t((p.w1_bytes + p.w2_bytes) / sizeof(cx<T>));
    cx<T>* W1 = reinterpret_cast<cx<T>*>(ws);
    cx<T>* W2 = p.inplace ? W1 : reinterpret_cast<cx<T>*>(reinterpret_cast<char*>(ws) + p.w1_bytes);
    const cx<T>* twN = twiddles<T>(N, &err);
    if (!twN) return err;
    const cx<T>* twM = twiddles<T>(M, &err);
    if (!twM) return err;
    const int tl = p.tc << p.log_k;
    int ltl = 0;
    while ((1 << ltl) < tl) ++ltl;
    if (p.fold) {
        // folded chain: row FFT + radix-2 DIF step -> two planes of M/2 rows; column FFT x H x IFFT per plane on M/2
        // points (in place); radix-2 DIT step + inverse row FFT -> natural output
        const int H = int(M / 2);
        const int64_t ntl = (N + tl - 1) / tl, plane = ntl * H * tl;
        const cx<T>* twH = twiddles<T>(H, &err);
        if (!twH) return err;
        const size_t in_bytes = size_t(M) * size_t(d->in_x.len) * sizeof(cx<T>);
        const int nt_in = tuning().nt_in >= 0 ? tuning().nt_in : (in_bytes >= (size_t(96) << 20) ? 1 : 0);
        RowLoadNat<T> lp{reinterpret_cast<const cx<T>*>(in), d->in_ld, to_map(d->in_x), int(M), 0, nt_in, 0, 0, H};
        set_input_mode(lp, d);
        RowStoreFold<T> sp{W1, plane, H, ltl, twM, d->in_y.shift == M / 2 ? 1 : 0, 0};
        int rc = launch_row_fold<T>(p.logn, lp, sp, twN, H, 0, st, 1);
        if (rc) return rc;
        const int ntiles = int((N + p.tc - 1) / p.tc);
        ColLoadTiled<T> cl{W1, H, AxisMap{H, H, 0, 0}, ntiles, p.log_k, plane};
        MidMul<T> mm{d->mul_kind, d->mul_conj, reinterpret_cast<const cx<T>*>(d->mul), reinterpret_cast<const cx<T>*>(d->mul_x),
                     2 * d->mul_ld, int(N), d->mul_kind == PM_MUL_FULL ? d->mul_ld : 1, 0, 2, 0};
        mm.vec_ok = (d->mul_kind == PM_MUL_FULL && sizeof(T) == 4 && d->mul_ld % 2 == 0 &&
                     reinterpret_cast<uintptr_t>(d->mul) % 16 == 0) ? 1 : 0;
        ColStoreTiled<T> cst{W1, H, ntiles, p.log_k, plane};
        rc = launch_col_mul<T>(p.logm - 1, cl, mm, cst, twH, ntiles, sibling_log_g(p.log_k), st, 2, tuning().colmul_mode);
        if (rc) return rc;
        RowLoadFold<T> rl{W1, plane, H, ltl, twM, 1, 0};
        RowStoreNat<T> rs{reinterpret_cast<cx<T>*>(out), d->out_ld, to_map(d->out_x), int(M), 1, T(d->scale), 1, to_map(d->out_y), 0, H};
        rs.nt = row_store_nt(size_t(d->out_y.len) * size_t(d->out_x.len) * sizeof(cx<T>));
        return launch_row_unfold<T>(p.logn, rl, rs, twN, H, st, 1);
    }
    // pass A: forward row transforms of the stored input rows -> tiled W1
    if (rows > 0) {
        const size_t in_bytes = size_t(p.nbatch) * size_t(rows) * size_t(d->in_x.len) * sizeof(cx<T>);
        const int nt_in = tuning().nt_in >= 0 ? tuning().nt_in : (in_bytes >= (size_t(96) << 20) ? 1 : 0);
        RowLoadNat<T> lp{reinterpret_cast<const cx<T>*>(in), d->in_ld, to_map(d->in_x), rows, 0, nt_in, d->in_bstride};
        set_input_mode(lp, d);
        RowStoreTiled<T> sp{W1, rows, ltl, wstride};
        int rc = launch_row_tiled<T>(p.logn, row_variant(d->dtype, p.logn), lp, sp, twN, rows, tuning().row_log_g, st, nb);
        if (rc) return rc;
    }
    // pass B: column FFT, x H, column IFFT (unnormalised) -> tiled W2 (all M rows)
    const int ntiles = int((N + p.tc - 1) / p.tc);
    ColLoadTiled<T> cl{W1, rows, to_map(d->in_y), ntiles, p.log_k, wstride};
    MidMul<T> mm{d->mul_kind, d->mul_conj, reinterpret_cast<const cx<T>*>(d->mul), reinterpret_cast<const cx<T>*>(d->mul_x),
                 d->mul_ld, int(N), d->mul_bstride, d->mul_x_bstride, 0,
                 (d->mul_kind == PM_MUL_FULL && sizeof(T) == 4 && d->mul_ld % 2 == 0 && d->mul_bstride % 2 == 0 &&
                  reinterpret_cast<uintptr_t>(d->mul) % 16 == 0) ? 1 : 0};
    ColStoreTiled<T> cst{W2, int(M), ntiles, p.log_k, wstride};
    int rc = launch_col_mul<T>(p.logm, cl, mm, cst, twM, ntiles, sibling_log_g(p.log_k), st, nb, tuning().colmul_mode);
    if (rc) return rc;
    // pass C: inverse row transforms of the rows inside the output window -> natural output, scale applied here.
    // Sequence s is stored row s of W2 (= logical row s); the output row map rotates / crops it.
    // An unrotated row window (crops: adjoints, the Bluestein convolution) only transforms its own rows [off, off + len).
    int row0 = 0, nrun = int(M);
    AxisMap oy = to_map(d->out_y);
    if (d->out_y.shift == 0 && d->out_y.len < M) {
        row0 = int(d->out_y.off);
        nrun = int(d->out_y.len);
        oy = AxisMap{nrun, nrun, 0, 0};
    }
    RowLoadTiled<T> rl{W2, int(M), ltl, row0, nrun, 1, wstride};
    RowStoreNat<T> rs{reinterpret_cast<cx<T>*>(out), d->out_ld, to_map(d->out_x), nrun, 1, T(d->scale), 1, oy, d->out_bstride};
    rs.nt = row_store_nt(size_t(p.nbatch) * size_t(d->out_y.len) * size_t(d->out_x.len) * sizeof(cx<T>));
    return launch_row_from_tiled<T>(p.logn, row_variant(d->dtype, p.logn), rl, rs, twN, nrun, st, nb);
}

// the composite-grid chain (plan_fused_mix)
template <typename T>
static int fused_mix_run(const pm_fft2_desc* d, const FusedPlan& p, const void* in, void* out, void* ws, hipStream_t st) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    cx<T>* W1 = reinterpret_cast<cx<T>*>(ws);
    cx<T>* W2 = p.inplace ? W1 : reinterpret_cast<cx<T>*>(reinterpret_cast<char*>(ws) + p.w1_bytes);
    // pass A: forward row transforms of the stored input rows -> natural W1 (the first pass of pm_fft2 on this shape)
    pm_fft2_desc da = *d;
    da.flags = (d->flags & PM_FLAG_REAL_INPUT) | PM_FLAG_PASS1_ONLY;
    da.mul_kind = PM_MUL_NONE;
    da.direction = -1;
    da.batch = 0;
    const Fft2Plan pa = plan_fft2(&da);
    if (pa.tc != 0 || pa.w_ld != p.w_ld || pa.blue_n || pa.blue2d || pa.big_rn)
        return fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: internal: the composite-grid chain and the transform planner disagree on %lld x %lld", (long long)M, (long long)N);
    int rc = fft2_run_chunk<T>(&da, pa, in, out, W1, st, 1);
    if (rc) return rc;
    // pass B: column transform, x H, inverse column transform (unnormalised) with the columns resident in LDS -> natural W2, all M rows
    DirectIn<T> di{W1, 1, p.w_ld, to_map(d->in_y), int(N), 0, 0};
    MidMul<T> mm{d->mul_kind, d->mul_conj, reinterpret_cast<const cx<T>*>(d->mul), reinterpret_cast<const cx<T>*>(d->mul_x), d->mul_ld, int(N), 0, 0, 0, 0};
    if ((rc = mix_cols_mul<T>(di, mm, W2, p.w_ld, st))) return rc;
    // pass C: inverse row transforms (conj in, conj out) of the rows the output window keeps, scale applied here
    int err = 0;
    if (p.logn >= 0) {
        const cx<T>* twN = twiddles<T>(N, &err);
        if (!twN) return err;
        RowLoadNat<T> lp{W2, p.w_ld, AxisMap{int(N), int(N), 0, 0}, int(M), 1, 0, 0};
        RowStoreNat<T> rs{reinterpret_cast<cx<T>*>(out), d->out_ld, to_map(d->out_x), int(M), 1, T(d->scale), 1, to_map(d->out_y), 0};
        rs.nt = row_store_nt(size_t(d->out_y.len) * size_t(d->out_x.len) * sizeof(cx<T>));
        return launch_row_nat<T>(p.logn, row_variant(d->dtype, p.logn), lp, rs, twN, int(M), 0, st);
    }
    // the mixed-radix row kernel writes sequence s to memory row s: the kept positions [off, off + len) of the rotated rows are at most
    // two runs of consecutive logical rows
    const int64_t off = d->out_y.off, len = d->out_y.len, sh = d->out_y.shift;
    const int64_t cut = sh > off ? (sh < off + len ? sh : off + len) : off;      // positions [off, cut) are logical rows p - sh + M
    const int64_t runs[2][3] = {{off - sh + M, 0, cut - off}, {cut - sh, cut - off, off + len - cut}};     // first logical row, first memory row, count
    for (const auto& r : runs) {
        if (r[2] <= 0) continue;
        DirectIn<T> ri{W2 + r[0] * p.w_ld, p.w_ld, 1, AxisMap{int(N), int(N), 0, 0}, int(r[2]), 1, 0};
        RowStoreNat<T> rs{reinterpret_cast<cx<T>*>(out) + r[1] * d->out_ld, d->out_ld, to_map(d->out_x), int(r[2]), 1, T(d->scale), 0, AxisMap{1, 1, 0, 0}, 0};
        if ((rc = mix_rows<T>(ri, nullptr, 0, st, &rs))) return rc;
    }
    return 0;
}

// ---------------------------------------------------------------- both axes not powers of two: 2-D Bluestein
// The 2-D cyclic convolution with the separable chirp IS the fused chain: window(ifft2(fft2(pad(a)) * (B1 (x) B2))) with the
// pad window [0, n) of MB on the way in and the same crop on the way out.
static void blue2d_desc(pm_fft2_desc& dd, int dtype, int64_t M, int64_t N) {
    memset(&dd, 0, sizeof dd);
    const int64_t mb1 = blue_conv_len(M), mb2 = blue_conv_len(N);
    dd.dtype = dtype;
    dd.direction = -1;
    dd.in_y = dd.out_y = pm_axis{mb1, M, 0, 0};
    dd.in_x = dd.out_x = pm_axis{mb2, N, 0, 0};
    dd.in_ld = dd.out_ld = N;
    dd.scale = 1.0;   // 1 / (MB1 MB2) lives in the tables
    dd.weight = 1.0;
    dd.mul_kind = PM_MUL_SEPARABLE;
}
static size_t blue2d_fused_ws(int dtype, int64_t M, int64_t N) {
    pm_fft2_desc dd;
    blue2d_desc(dd, dtype, M, N);
    FusedPlan fp;
    return plan_fused(&dd, fp) ? fp.ws_bytes : 0;
}

template <typename T>
static int blue2d_fused_run(const pm_fft2_desc* d, const void* in, void* out, void* ws, hipStream_t st);

template <typename T>
static int blue2d_run(const pm_fft2_desc* d, const Fft2Plan& p, const void* in, void* out, void* ws, hipStream_t st) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    int err = 0;
    const cx<T>* t1 = blue_tables<T>(M, &err);
    if (!t1) return err;
    const cx<T>* t2 = blue_tables<T>(N, &err);
    if (!t2) return err;
    const size_t arr = (size_t(M) * size_t(N) * sizeof(cx<T>) + 255) & ~size_t(255);
    cx<T>* a = reinterpret_cast<cx<T>*>(ws);
    cx<T>* c = reinterpret_cast<cx<T>*>(static_cast<char*>(ws) + arr);
    void* fws = static_cast<char*>(ws) + 2 * arr;
    Blue2dIn<T> bi{in, d->in_ld, to_map(d->in_y), to_map(d->in_x), d->direction > 0 ? 1 : 0, (d->flags & PM_FLAG_REAL_INPUT) ? 1 : 0};
    if (!p.blue_big && tuning().blue_fuse) return blue2d_fused_run<T>(d, in, out, fws, st);
    int rc = blue_pre2d<T>(bi, a, t1, t2, st);
    if (rc) return rc;
    pm_fft2_desc dd;
    blue2d_desc(dd, d->dtype, M, N);
    dd.mul = t1 + M;
    dd.mul_x = t2 + N;
    if (p.blue_big) {
        // convolution lengths above the engine's (n in (4096, 16384]): spectrum = fft2(pad(a)) x (B1 (x) B2) by one big transform
        // with the multiplier in its epilogue, then the cropped inverse by a second one (bigfft.hip)
        const int64_t mb1 = dd.in_y.n, mb2 = dd.in_x.n;
        const size_t spec = (size_t(mb1) * size_t(mb2) * sizeof(cx<T>) + 255) & ~size_t(255);
        cx<T>* S = reinterpret_cast<cx<T>*>(fws);
        void* bws = static_cast<char*>(fws) + spec;
        pm_fft2_desc d1 = dd;
        d1.out_y = pm_axis{mb1, mb1, 0, 0};
        d1.out_x = pm_axis{mb2, mb2, 0, 0};
        d1.out_ld = mb2;
        const Fft2Plan p1 = plan_fft2(&d1);
        if (!p1.big_rn) return fail(PM_ERR_UNSUPPORTED, "pm_fft2: internal: no big plan for the Bluestein convolution");
        if ((rc = big2d_run<T>(&d1, p1, a, S, bws, st))) return rc;
        pm_fft2_desc d2 = dd;
        d2.direction = +1;
        d2.in_y = pm_axis{mb1, mb1, 0, 0};
        d2.in_x = pm_axis{mb2, mb2, 0, 0};
        d2.in_ld = mb2;
        d2.mul_kind = PM_MUL_NONE;
        d2.mul = d2.mul_x = nullptr;
        const Fft2Plan p2 = plan_fft2(&d2);
        if (!p2.big_rn) return fail(PM_ERR_UNSUPPORTED, "pm_fft2: internal: no big plan for the Bluestein convolution");
        if ((rc = big2d_run<T>(&d2, p2, S, c, bws, st))) return rc;
        const ColStoreNat<T> cs = make_colstore<T>(d, out);
        return blue_post2d<T>(c, int(M), int(N), t1, t2, cs, st);
    }
    FusedPlan fp;
    if (!plan_fused(&dd, fp)) return fail(PM_ERR_UNSUPPORTED, "pm_fft2: internal: no fused plan for the Bluestein convolution");
    rc = fused_run_chunk<T>(&dd, fp, a, c, fws, st, 1);
    if (rc) return rc;
    const ColStoreNat<T> cs = make_colstore<T>(d, out);
    return blue_post2d<T>(c, int(M), int(N), t1, t2, cs, st);
}

// The same on engine lengths with the two chirp multiplies inside the chain: the first row pass loads the caller's view times
// w1 (x) w2 (RowLoadChirp), the last one stores conj(.) w1 (x) w2 through the caller's epilogue (RowStoreChirp).  Three launches,
// no n1 x n2 temporaries.  (The unfolded passes of fused_run_chunk with those two ends.)
template <typename T>
static int blue2d_fused_run(const pm_fft2_desc* d, const void* in, void* out, void* ws, hipStream_t st) {
    const int64_t n1 = d->in_y.n, n2 = d->in_x.n;
    int err = 0;
    const cx<T>* t1 = blue_tables<T>(n1, &err);
    if (!t1) return err;
    const cx<T>* t2 = blue_tables<T>(n2, &err);
    if (!t2) return err;
    pm_fft2_desc dd;
    blue2d_desc(dd, d->dtype, n1, n2);
    FusedPlan fp;
    if (!plan_fused(&dd, fp) || fp.fold) return fail(PM_ERR_UNSUPPORTED, "pm_fft2: internal: no fused plan for the Bluestein convolution");
    const int64_t M = dd.in_y.n, N = dd.in_x.n;   // convolution lengths
    const int rows = int(n1);
    cx<T>* W1 = reinterpret_cast<cx<T>*>(ws);
    cx<T>* W2 = fp.inplace ? W1 : reinterpret_cast<cx<T>*>(reinterpret_cast<char*>(ws) + fp.w1_bytes);
    const cx<T>* twN = twiddles<T>(N, &err);
    if (!twN) return err;
    const cx<T>* twM = twiddles<T>(M, &err);
    if (!twM) return err;
    const int tl = fp.tc << fp.log_k;
    int ltl = 0;
    while ((1 << ltl) < tl) ++ltl;
    // pass A: rows x(i, .) w1[i] w2[.] padded to N, forward transform -> tiled W1 (n1 rows)
    RowLoadChirp<T> lp{Blue2dIn<T>{in, d->in_ld, to_map(d->in_y), to_map(d->in_x), d->direction > 0 ? 1 : 0,
                                   (d->flags & PM_FLAG_REAL_INPUT) ? 1 : 0},
                       t1, t2, rows};
    RowStoreTiled<T> sp{W1, rows, ltl, 0};
    int rc = launch_row_chirp_tiled<T>(fp.logn, row_variant(d->dtype, fp.logn), lp, sp, twN, rows, tuning().row_log_g, st);
    if (rc) return rc;
    // pass B: column FFT x (B1 (x) B2) x column IFFT -> tiled W2
    const int ntiles = int((N + fp.tc - 1) / fp.tc);
    ColLoadTiled<T> cl{W1, rows, AxisMap{int(M), rows, 0, 0}, ntiles, fp.log_k, 0};
    MidMul<T> mm{MUL_SEPARABLE, 0, t1 + n1, t2 + n2, 0, int(N), 0, 0, 0, 0};
    ColStoreTiledCrop<T> cst{W2, rows, ntiles, fp.log_k};   // only the n1 rows the crop keeps are stored
    rc = launch_col_mul_crop<T>(fp.logm, cl, mm, cst, twM, ntiles, sibling_log_g(fp.log_k), st);
    if (rc) return rc;
    // pass C: inverse row transforms of the first n1 rows, bins [0, n2) x chirp through the caller's epilogue
    RowLoadTiled<T> rl{W2, rows, ltl, 0, rows, 1, 0};
    RowStoreChirp<T> rs{make_colstore<T>(d, out), t1, t2, int(n1), int(n2), 1};
    return launch_row_tiled_chirp<T>(fp.logn, row_variant(d->dtype, fp.logn), rl, rs, twN, rows, st);
}

// ---------------------------------------------------------------- powers of two above the engine's longest transform
// (bigfft.hip): rows by a decimation-in-frequency step in front of ONE engine row pass over R_n planes, columns by engine
// passes over the R_m row sub-lattices and a combining epilogue kernel.
template <typename T>
static int big2d_run(const pm_fft2_desc* d, const Fft2Plan& p, const void* in, void* out, void* ws, hipStream_t st) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    const int Rn = p.big_rn, Rm = p.big_rm;
    const int np = int(N / Rn), mp = int(M / Rm);
    const int lgn = engine_log2(np), lgm = engine_log2(mp);      // < 0: that sub-transform runs on the mixed-radix kernel (big_split2d)
    if ((lgn < 0 && !use_mix(np)) || (lgm < 0 && !use_mix(mp)))
        return fail(PM_ERR_UNSUPPORTED, "pm_fft2: internal: big split %d x %d of %lld x %lld", Rm, Rn, (long long)M, (long long)N);
    const int dt = d->dtype;
    const int conj = d->direction > 0 ? 1 : 0;
    int err = 0;
    const size_t arr = (size_t(M) * size_t(N) * sizeof(cx<T>) + 255) & ~size_t(255);
    cx<T>* Z = reinterpret_cast<cx<T>*>(ws);
    cx<T>* F = reinterpret_cast<cx<T>*>(static_cast<char*>(ws) + arr);
    const cx<T>* twn = lgn >= 0 ? twiddles<T>(np, &err) : nullptr;
    if (lgn >= 0 && !twn) return err;
    const cx<T>* twm = twiddles<T>(mp, &err);
    if (!twm) return err;
    int rc;
    // ---- rows -> Z[m][i][k] = X_row_i[R_n k + m], every LOGICAL row i present
    if (Rn > 1) {
        const cx<T>* twN = twiddles<T>(N, &err);
        if (!twN) return err;
        Blue2dIn<T> bi{in, d->in_ld, to_map(d->in_y), to_map(d->in_x), conj, (d->flags & PM_FLAG_REAL_INPUT) ? 1 : 0};
        cx<T>* Y = F;   // dead before the column stage writes F
        if ((rc = big_pre_rows<T>(bi, int(M), np, Rn, Y, twN, st))) return rc;
        const int nseq = Rn * int(M);
        if (lgn < 0) {
            DirectIn<T> ri{Y, np, 1, AxisMap{np, np, 0, 0}, nseq, 0, 0};
            if ((rc = mix_rows<T>(ri, Z, np, st))) return rc;
        } else {
            RowLoadNat<T> lp{Y, np, AxisMap{np, np, 0, 0}, nseq, 0, 0};
            RowStoreNat<T> sp{Z, np, AxisMap{np, np, 0, 0}, nseq, 0, T(1), 0, AxisMap{1, 1, 0, 0}};
            if ((rc = launch_row_nat<T>(lgn, row_variant(dt, lgn), lp, sp, twn, nseq, 0, st))) return rc;
        }
    } else if (lgn < 0) {
        // a composite row length as it is (beside a column length that needs the step): the mixed-radix row kernel writes sequence s to
        // memory row s, so the stored rows go out as the (at most two) runs of consecutive LOGICAL rows they are
        const int rows = int(d->in_y.len);
        if (rows < M) {
            hipError_t e = hipMemsetAsync(Z, 0, size_t(M) * size_t(N) * sizeof(cx<T>), st);
            if (e != hipSuccess) return int(e);
        }
        const int64_t c = ((d->in_y.off - d->in_y.shift) % M + M) % M;      // stored row q is logical row (q + c) mod M
        const int64_t n1 = rows < M - c ? rows : M - c;
        const int64_t runs[2][3] = {{0, c, n1}, {n1, 0, rows - n1}};       // first stored row, first logical row, count
        const size_t ies = (d->flags & PM_FLAG_REAL_INPUT) ? sizeof(T) : sizeof(cx<T>);
        for (const auto& r : runs) {
            if (r[2] <= 0) continue;
            DirectIn<T> ri{reinterpret_cast<const cx<T>*>(static_cast<const char*>(in) + size_t(r[0]) * size_t(d->in_ld) * ies), d->in_ld, 1, to_map(d->in_x),
                           int(r[2]), conj, (d->flags & PM_FLAG_REAL_INPUT) ? 1 : 0};
            if ((rc = mix_rows<T>(ri, Z + r[1] * N, N, st))) return rc;
        }
    } else {
        const int rows = int(d->in_y.len);
        if (rows < M) {
            hipError_t e = hipMemsetAsync(Z, 0, size_t(M) * size_t(N) * sizeof(cx<T>), st);
            if (e != hipSuccess) return int(e);
        }
        if (rows > 0) {
            // stored row q is logical row (q + off - shift) mod M: the row map of the store puts it there
            const int sh = int(((d->in_y.off - d->in_y.shift) % M + M) % M);
            RowLoadNat<T> lp{reinterpret_cast<const cx<T>*>(in), d->in_ld, to_map(d->in_x), rows, conj, 0};
            set_input_mode(lp, d);
            RowStoreNat<T> sp{Z, N, AxisMap{int(N), int(N), 0, 0}, rows, 0, T(1), 1, AxisMap{int(M), int(M), 0, sh}};
            if ((rc = launch_row_nat<T>(lgn, row_variant(dt, lgn), lp, sp, twn, rows, 0, st))) return rc;
        }
    }
    // ---- columns: plane m is an M x np matrix; F[(m R_m + r)] = FFT_{mp} down the columns of its rows r, r + R_m, ...
    const int tc = col_tile_width_for(dt, lgm, tuning().col_var);
    const int ntiles = (np + tc - 1) / tc;
    const int vec = (sizeof(T) != 4 || np % 2 == 0) ? 1 : 0;
    const int64_t plane = int64_t(mp) * np;
    for (int m = 0; m < Rn; ++m) {
        if (lgm < 0) {      // the sub-lattices r, r + R_m, ... one launch each on the mixed-radix column kernel
            for (int r = 0; r < Rm; ++r) {
                DirectIn<T> ci{Z + int64_t(m) * M * np + int64_t(r) * np, 1, int64_t(Rm) * np, AxisMap{mp, mp, 0, 0}, np, 0, 0};
                ColStoreNat<T> cs{};
                cs.dst = F + (int64_t(m) * Rm + r) * plane;
                cs.ld = np;
                cs.ay = AxisMap{mp, mp, 0, 0};
                cs.ax = AxisMap{np, np, 0, 0};
                cs.epilogue = EPI_NONE;
                cs.scale = T(1);
                cs.weight = T(1);
                cs.mul_kind = MUL_NONE;
                cs.vec_ok = vec;
                if ((rc = mix_cols<T>(ci, cs, st))) return rc;
            }
            continue;
        }
        ColLoadNat<T> cl{Z + int64_t(m) * M * np, int64_t(Rm) * np, AxisMap{mp, mp, 0, 0}, np, 0, vec, int64_t(np)};
        ColStoreNat<T> cs{};
        cs.dst = F + int64_t(m) * Rm * plane;
        cs.ld = np;
        cs.ay = AxisMap{mp, mp, 0, 0};
        cs.ax = AxisMap{np, np, 0, 0};
        cs.epilogue = EPI_NONE;
        cs.scale = T(1);
        cs.weight = T(1);
        cs.mul_kind = MUL_NONE;
        cs.vec_ok = vec;
        cs.bstride = plane;
        if ((rc = launch_col_nat<T>(lgm, 0, cl, cs, twm, ntiles, 1, st, Rm))) return rc;
    }
    // ---- combine the sub-lattices, un-interleave the row split, common epilogue
    const cx<T>* twM = twm;
    if (Rm > 1) {
        twM = twiddles<T>(M, &err);
        if (!twM) return err;
    }
    ColStoreNat<T> ep = make_colstore<T>(d, out);
    ep.bstride = 0;
    return big_finish<T>(F, mp, np, Rm, Rn, twM, ep, st);
}

template <typename T>
static int fused_run(const pm_fft2_desc* d, const FusedPlan& p, const void* in, void* out, void* ws, hipStream_t st) {
    if (p.mixmid) return fused_mix_run<T>(d, p, in, out, ws, st);
    for (int64_t b0 = 0; b0 < p.nbatch; b0 += p.chunk) {
        const int nb = int(p.nbatch - b0 < p.chunk ? p.nbatch - b0 : p.chunk);
        pm_fft2_desc dd = *d;
        dd.mul = offset_elems(d->mul, b0 * d->mul_bstride, sizeof(cx<T>));
        dd.mul_x = offset_elems(d->mul_x, b0 * d->mul_x_bstride, sizeof(cx<T>));
        int rc = fused_run_chunk<T>(&dd, p, offset_elems(in, b0 * d->in_bstride, (d->flags & PM_FLAG_REAL_INPUT) ? sizeof(T) : sizeof(cx<T>)),
                                    const_cast<void*>(offset_elems(out, b0 * d->out_bstride, sizeof(cx<T>))), ws, st, nb);
        if (rc) return rc;
    }
    return 0;
}

static int check_fft2(const pm_fft2_desc* d) {
    if (!d) return fail(PM_ERR_ARG, "pm_fft2: null descriptor");
    if (d->dtype != PM_C64 && d->dtype != PM_C128) return fail(PM_ERR_ARG, "pm_fft2: dtype must be PM_C64 or PM_C128");
    if (d->direction != 1 && d->direction != -1) return fail(PM_ERR_ARG, "pm_fft2: direction must be -1 or +1");
    if (d->epilogue < PM_EPI_NONE || d->epilogue > PM_EPI_ARG) return fail(PM_ERR_ARG, "pm_fft2: bad epilogue");
    if ((d->epilogue > PM_EPI_ABS2_ACCUM || (d->flags & PM_FLAG_NORM_DC)) &&
        !r2c_legal(d, engine_log2(d->in_x.n), engine_log2(d->in_y.n)))
        return fail(PM_ERR_UNSUPPORTED, "pm_fft2: PM_EPI_ABS / PM_EPI_ARG / PM_FLAG_NORM_DC exist on the Hermitian path only (a forward "
                    "transform of an unpadded real field with power-of-two lengths and an unwindowed output)");
    if (d->mul_kind < PM_MUL_NONE || d->mul_kind > PM_MUL_SEPARABLE) return fail(PM_ERR_ARG, "pm_fft2: bad mul_kind");
    if (d->mul_kind != PM_MUL_NONE && !d->mul) return fail(PM_ERR_ARG, "pm_fft2: mul is null");
    if (d->mul_kind == PM_MUL_SEPARABLE && !d->mul_x) return fail(PM_ERR_ARG, "pm_fft2: mul_x is null");
    int rc;
    if ((rc = check_axis(d->in_y, "in_y")) || (rc = check_axis(d->in_x, "in_x")) ||
        (rc = check_axis(d->out_y, "out_y")) || (rc = check_axis(d->out_x, "out_x")))
        return rc;
    if (d->in_y.n != d->out_y.n || d->in_x.n != d->out_x.n)
        return fail(PM_ERR_ARG, "pm_fft2: input and output views must share the transform size");
    if (d->in_ld < d->in_x.len || d->out_ld < d->out_x.len) return fail(PM_ERR_ARG, "pm_fft2: leading dimension < row length");
    if ((d->flags & PM_FLAG_SYNTH_PACKED) && !(d->flags & PM_FLAG_SYNTH_INPUT))
        return fail(PM_ERR_ARG, "pm_fft2: PM_FLAG_SYNTH_PACKED qualifies PM_FLAG_SYNTH_INPUT");
    if (d->flags & PM_FLAG_SYNTH_INPUT) {
        if (d->direction != -1) return fail(PM_ERR_ARG, "pm_fft2: PM_FLAG_SYNTH_INPUT is a forward transform");
        if ((engine_log2(d->in_x.n) < 0 && !use_mix(d->in_x.n)) || d->batch > 1)
            return fail(PM_ERR_UNSUPPORTED, "pm_fft2: PM_FLAG_SYNTH_INPUT needs a row length that is a power of two or a composite with primes "
                        "<= 19 (the kernels whose loaders synthesise the pupil), and no batch");
        if (d->synth_amp && d->synth_amp_dtype != PM_F32 && d->synth_amp_dtype != PM_F64 && d->synth_amp_dtype != PM_BOOL)
            return fail(PM_ERR_ARG, "pm_fft2: synth_amp_dtype");
        if (d->synth_amp && d->synth_amp_ld < d->in_x.len) return fail(PM_ERR_ARG, "pm_fft2: synth_amp_ld < row length");
    }
    if (d->batch < 0 || d->batch > 65535) return fail(PM_ERR_ARG, "pm_fft2: batch = %lld must be in [0, 65535]", (long long)d->batch);
    if (d->batch > 1) {
        if (d->in_bstride < 0 || d->out_bstride < 0 || d->mul_bstride < 0 || d->mul_x_bstride < 0)
            return fail(PM_ERR_ARG, "pm_fft2: batch strides must be >= 0");
        if (d->out_bstride < (d->out_y.len > 0 ? (d->out_y.len - 1) * d->out_ld + d->out_x.len : 0))
            return fail(PM_ERR_ARG, "pm_fft2: out_bstride = %lld makes the outputs of a batch overlap", (long long)d->out_bstride);
    }
    const int64_t lim = int64_t(1) << 15;
    if ((engine_log2(d->in_x.n) < 0 && d->in_x.n > lim) || (engine_log2(d->in_y.n) < 0 && d->in_y.n > lim))
        return fail(PM_ERR_UNSUPPORTED, "pm_fft2: length %lld x %lld: powers of two up to 8192 run on the FFT engine, other "
                    "lengths up to 32768 on the direct DFT",
                    (long long)d->in_y.n, (long long)d->in_x.n);
    return 0;
}

// 1-D transforms of the lengths of bigfft.hip, n = R n' (16384, 32768; 3 / 5 / 7 x 2^k): one radix-R step around engine transforms
// of length n', the pieces of big2d_run with the other axis left alone.
//   axis 1 (rows)     decimation in frequency: big_pre_rows -> ONE engine row pass over the R planes -> big_finish with a unit column
//                     radix un-interleaves the bins (X[R k + m] = plane m, bin k) through the output view;  workspace 2 batch n
//   axis 0 (columns)  decimation in time: the sub-sequence r is rows r, r + R, ... of the caller's array -- a leading dimension of R
//                     rows and the stored window cut to the rows of that residue -- R engine column passes -> big_finish combines;
//                     workspace batch n
static size_t fft1_big_scratch(size_t es, int axis, int64_t batch, int64_t n) {
    const size_t arr = (size_t(batch) * size_t(n) * es + 255) & ~size_t(255);
    return axis == 1 ? 2 * arr : arr;
}
static bool fft1_big_ok(const pm_axis* ti) { return big_split(ti->n) > 1 && ti->shift == 0; }

template <typename T>
static int fft1_big(int conj, int axis, int64_t batch, const pm_axis* ti, const pm_axis* to, double scale, const void* in, int64_t in_ld,
                    void* out, int64_t out_ld, hipStream_t st, void* ws) {
    const int64_t n = ti->n;
    const int R = big_split(n), np = int(n / R), lg = engine_log2(np);
    const int dt = sizeof(T) == 4 ? PM_C64 : PM_C128;
    int err = 0, rc;
    const cx<T>* twp = twiddles<T>(np, &err);
    if (!twp) return err;
    const cx<T>* twN = twiddles<T>(n, &err);
    if (!twN) return err;
    ColStoreNat<T> o{};
    o.dst = out;
    o.ld = out_ld;
    o.conj = conj;
    o.epilogue = EPI_NONE;
    o.scale = T(scale);
    o.weight = T(1);
    o.mul_kind = MUL_NONE;
    const int nb = int(batch);
    if (axis == 1) {
        o.ay = AxisMap{nb, nb, 0, 0};
        o.ax = to_map(*to);
        const size_t arr = (size_t(batch) * size_t(n) * sizeof(cx<T>) + 255) & ~size_t(255);
        cx<T>* Y = reinterpret_cast<cx<T>*>(ws);
        cx<T>* Z = reinterpret_cast<cx<T>*>(static_cast<char*>(ws) + arr);
        Blue2dIn<T> bi{in, in_ld, AxisMap{nb, nb, 0, 0}, to_map(*ti), conj, 0};
        if ((rc = big_pre_rows<T>(bi, nb, np, R, Y, twN, st))) return rc;
        const int nseq = R * nb;
        RowLoadNat<T> lp{Y, np, AxisMap{np, np, 0, 0}, nseq, 0, 0};
        RowStoreNat<T> sp{Z, np, AxisMap{np, np, 0, 0}, nseq, 0, T(1), 0, AxisMap{1, 1, 0, 0}};
        if ((rc = launch_row_nat<T>(lg, row_variant(dt, lg), lp, sp, twp, nseq, 0, st))) return rc;
        return big_finish<T>(Z, nb, np, 1, R, twp, o, st);
    }
    o.ay = to_map(*to);
    o.ax = AxisMap{nb, nb, 0, 0};
    cx<T>* F = reinterpret_cast<cx<T>*>(ws);
    const int64_t plane = int64_t(np) * batch;
    const int tc = col_tile_width_for(dt, lg, tuning().col_var);
    const int ntiles = int((batch + tc - 1) / tc);
    const int vec_in = (sizeof(T) != 4 || ((in_ld % 2 == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0))) ? 1 : 0;
    const int vec_f = (sizeof(T) != 4 || batch % 2 == 0) ? 1 : 0;
    const int64_t off = ti->off, end = ti->off + ti->len;
    for (int r = 0; r < R; ++r) {
        // logical rows r + R i (i < np) of the zero-padded sequence; stored: off <= r + R i < off + len
        const int64_t ilo = off > r ? (off - r + R - 1) / R : 0;
        int64_t ihi = end > r ? (end - r + R - 1) / R : 0;
        if (ihi > np) ihi = np;
        cx<T>* Fr = F + int64_t(r) * plane;
        if (ihi <= ilo) {
            hipError_t e = hipMemsetAsync(Fr, 0, size_t(plane) * sizeof(cx<T>), st);
            if (e != hipSuccess) return int(e);
            continue;
        }
        const cx<T>* base = reinterpret_cast<const cx<T>*>(in) + (int64_t(R) * ilo + r - off) * in_ld;
        ColLoadNat<T> cl{base, int64_t(R) * in_ld, AxisMap{np, int(ihi - ilo), int(ilo), 0}, nb, conj, vec_in, 0};
        ColStoreNat<T> cs{};
        cs.dst = Fr;
        cs.ld = batch;
        cs.ay = AxisMap{np, np, 0, 0};
        cs.ax = AxisMap{nb, nb, 0, 0};
        cs.epilogue = EPI_NONE;
        cs.scale = T(1);
        cs.weight = T(1);
        cs.mul_kind = MUL_NONE;
        cs.vec_ok = vec_f;
        if ((rc = launch_col_nat<T>(lg, 0, cl, cs, twp, ntiles, 1, st, 1))) return rc;
    }
    return big_finish<T>(F, np, nb, R, 1, twN, o, st);
}

template <typename T>
static int fft1_run(int direction, int axis, int64_t batch, const pm_axis* ti, const pm_axis* to, double scale,
                    const void* in, int64_t in_ld, void* out, int64_t out_ld, hipStream_t st, void* blue_ws = nullptr) {
    const int64_t n = ti->n;
    const int lg = engine_log2(n);
    const int conj = direction > 0 ? 1 : 0;
    int err = 0;
    if (blue_ws && fft1_big_ok(ti)) return fft1_big<T>(conj, axis, batch, ti, to, scale, in, in_ld, out, out_ld, st, blue_ws);
    if (big_split(n) > 1) blue_ws = nullptr;     // the workspace was sized for the radix-R path
    if (axis == 1) {
        RowStoreNat<T> sp{reinterpret_cast<cx<T>*>(out), out_ld, to_map(*to), int(batch), conj, T(scale), 0, AxisMap{1, 1, 0, 0}};
        if (lg >= 0) {
            const cx<T>* tw = twiddles<T>(n, &err);
            if (!tw) return err;
            RowLoadNat<T> lp{reinterpret_cast<const cx<T>*>(in), in_ld, to_map(*ti), int(batch), conj, 0};
            return launch_row_nat<T>(lg, row_variant(sizeof(T) == 4 ? PM_C64 : PM_C128, lg), lp, sp, tw, int(batch), 0, st);
        }
        DirectIn<T> di{reinterpret_cast<const cx<T>*>(in), in_ld, 1, to_map(*ti), int(batch), conj};
        if (use_mix(n) && mix_fits(n, in_ld, sizeof(cx<T>), false) && mix_fits(n, out_ld, sizeof(cx<T>), false)) return mix_rows<T>(di, nullptr, 0, st, &sp);
        if (blue_ws) return blue_rows<T>(di, nullptr, 0, blue_ws, st, &sp);
        const cx<double>* tw = twiddles_f64(n, &err);
        if (!tw) return err;
        return direct_rows_out<T>(di, sp, tw, st);
    }
    // axis == 0: sequences are the `batch` columns
    ColStoreNat<T> cs{};
    cs.dst = out;
    cs.ld = out_ld;
    cs.ay = to_map(*to);
    cs.ax = AxisMap{int(batch), int(batch), 0, 0};
    cs.conj = conj;
    cs.epilogue = EPI_NONE;
    cs.scale = T(scale);
    cs.weight = T(1);
    cs.mul_kind = MUL_NONE;
    cs.vec_ok = (sizeof(T) != 4 || ((out_ld % 2 == 0) && (reinterpret_cast<uintptr_t>(out) % 16 == 0))) ? 1 : 0;
    if (lg >= 0) {
        const cx<T>* tw = twiddles<T>(n, &err);
        if (!tw) return err;
        const int tc = col_tile_width_for(sizeof(T) == 4 ? PM_C64 : PM_C128, lg, tuning().col_var);
        const int ntiles = int((batch + tc - 1) / tc);
        const int vec = (sizeof(T) != 4 || ((in_ld % 2 == 0) && (reinterpret_cast<uintptr_t>(in) % 16 == 0))) ? 1 : 0;
        ColLoadNat<T> cl{reinterpret_cast<const cx<T>*>(in), in_ld, to_map(*ti), int(batch), conj, vec};
        return launch_col_nat<T>(lg, 0, cl, cs, tw, ntiles, 1, st);
    }
    DirectIn<T> di{reinterpret_cast<const cx<T>*>(in), 1, in_ld, to_map(*ti), int(batch), conj};
    if (use_mix(n) && mix_fits(n, in_ld, sizeof(cx<T>), true) && mix_fits(to->n, out_ld, sizeof(cx<T>), true)) return mix_cols<T>(di, cs, st);
    if (blue_ws) return blue_cols<T>(di, cs, blue_ws, st);
    const cx<double>* tw = twiddles_f64(n, &err);
    if (!tw) return err;
    return direct_cols<T>(di, cs, tw, st);
}

template <typename T>
static int czt_axis_run(int32_t axis, int64_t nseq, int64_t K, int64_t in_len, int64_t in_off, int64_t out_len, int64_t out_off,
                        const void* pre, int pre_conj, const void* H, int h_conj, const void* post, int post_conj, double scale,
                        const void* in, int64_t in_ld, void* out, int64_t out_ld, hipStream_t st, int single = 0) {
    int err = 0;
    const cx<T>* tw = twiddles<T>(K, &err);
    if (!tw) return err;
    // 1 / K of the inverse transform rides on the scale
    Conv1<T> p{reinterpret_cast<const cx<T>*>(in), in_ld, reinterpret_cast<cx<T>*>(out), out_ld, int(nseq), int(in_len), int(in_off),
               int(out_len), int(out_off), reinterpret_cast<const cx<T>*>(pre), reinterpret_cast<const cx<T>*>(H),
               reinterpret_cast<const cx<T>*>(post), pre_conj ? 1 : 0, h_conj ? 1 : 0, post_conj ? 1 : 0,
               T(single ? scale : scale / double(K)), single};
    const int lg = engine_log2(K);
    return axis == 1 ? launch_conv1_rows<T>(lg, p, tw, st) : launch_conv1_cols<T>(lg, p, tw, st);
}

// ---- real object, real result: the chain on half spectra (fft_c2r.h)
struct HermConvPlan {
    int logn, logm, tc, log_k;
    bool fold;          // radix-2 step of the column transforms in the first / last row pass, as in the complex chain
    size_t ws_bytes;
};
static bool herm_conv_plan(const pm_fft2_desc* d, HermConvPlan& p) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    p.logn = engine_log2(N);
    p.logm = engine_log2(M);
    if (!(d->flags & PM_FLAG_REAL_INPUT) || (d->flags & (PM_FLAG_SYNTH_INPUT | PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY))) return false;
    if (p.logn < 6 || p.logm < 1 || d->batch > 1 || d->mul_kind != PM_MUL_FULL) return false;
    if (d->dtype == PM_C128 && p.logn > 12) return false;
    // Measured (profiles/r02/exp_conv.log, us, half spectra against the complex chain): 1024^2 fp32 39 vs 35 (three launch-bound passes
    // with an extra exchange each), 2048^2 59 vs 59 / fp64 74 vs 93, 4096^2 147 vs 214 / fp64 302 vs 419, 8192^2 589 vs 882: from 2048^2
    // (knob r2c = 2: always, 0: never)
    if (tuning().r2c == 0 || (tuning().r2c < 2 && M * N < (int64_t(1) << 22))) return false;
    if (d->in_y.len != M || d->in_x.len != N || d->out_y.len != M || d->out_x.len != N) return false;
    if (!(d->in_x.shift == 0 || d->in_x.shift == N / 2) || !(d->out_x.shift == 0 || d->out_x.shift == N / 2)) return false;
    if ((d->in_ld % 2) != 0 || (d->out_ld % 2) != 0) return false;
    // fold: rows of 4096 / 8192 samples (the two-rows-per-thread kernels exist for 2048 / 4096 complex points), rotations by 0 or M/2
    const int f = tuning().fold;
    // (automatic where it measured faster: 8192-row objects 589 vs 806 us, 4096-row fp64 302 vs 318; 4096-row fp32 is 151 vs 147)
    p.fold = (f > 0 || (f < 0 && (p.logm >= 13 || (p.logm == 12 && d->dtype == PM_C128)))) && p.logm >= 2 &&
             (p.logn == 12 || (p.logn == 13 && d->dtype == PM_C64)) &&
             (d->in_y.shift == 0 || d->in_y.shift == M / 2);
    p.tc = col_tile_width_for(d->dtype, p.fold ? p.logm - 1 : p.logm, 0);
    p.log_k = tuning().log_k >= 0 ? tuning().log_k : 2;
    while (p.log_k > 0 && ((N / 2) % (int64_t(p.tc) << p.log_k)) != 0) --p.log_k;
    if ((N / 2) % p.tc) return false;
    const size_t es = d->dtype == PM_C64 ? 8 : 16;
    p.ws_bytes = size_t(M) * size_t(N / 2) * es;
    return true;
}

template <typename T>
static int herm_conv_run(const pm_fft2_desc* d, const HermConvPlan& p, const void* in, void* out, void* ws, hipStream_t st) {
    const int64_t M = d->in_y.n, N = d->in_x.n, n2 = N / 2;
    int err = 0;
    cx<T>* W = reinterpret_cast<cx<T>*>(ws);
    const cx<T>* tw2 = twiddles<T>(n2, &err);
    if (!tw2) return err;
    const cx<T>* twn = twiddles<T>(N, &err);
    if (!twn) return err;
    const cx<T>* twm = twiddles<T>(M, &err);
    if (!twm) return err;
    const int64_t tl = int64_t(p.tc) << p.log_k;
    int ltl = 0;
    while ((int64_t(1) << ltl) < tl) ++ltl;
    // rows: the real array as N/2 complex points per row -> N/2 columns, column 0 = X[0] + i X[N/2]
    RowLoadNat<T> lp{reinterpret_cast<const cx<T>*>(in), d->in_ld / 2, AxisMap{int(n2), int(n2), 0, int(d->in_x.shift / 2)}, int(M), 0, 0, 0};
    R2CRowStore<T> rs{W, int(M), ltl, twn, 0, 0, nullptr, 0};
    if (p.fold) {
        // folded: two planes of M/2 rows (even / odd bins of the column transform), M/2-point column tiles, the last pass rebuilds row pairs
        const int H = int(M / 2);
        const int64_t plane = (n2 / tl) * H * tl;
        const cx<T>* twh = twiddles<T>(H, &err);
        if (!twh) return err;
        lp.eoff = H;
        rs.nseq = H;
        rs.fold = 1;
        rs.plane_stride = plane;
        rs.twm = twm;
        rs.swap = d->in_y.shift == M / 2 ? 1 : 0;
        int rcf = launch_row_r2c<T>(p.logn - 1, lp, rs, tw2, H, 0, st);
        if (rcf) return rcf < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: internal: no folded Hermitian row kernel for %lld points", (long long)N) : rcf;
        const int ntf = int(n2 / p.tc);
        ColLoadTiled<T> clf{W, H, AxisMap{H, H, 0, 0}, ntf, p.log_k, plane};
        HermMul<T> hmf{reinterpret_cast<const cx<T>*>(d->mul), d->mul_ld, int(M), int(N), d->mul_conj ? 1 : 0, 1};
        ColStoreTiled<T> csf{W, H, ntf, p.log_k, plane};
        rcf = launch_col_mul_herm<T>(p.logm - 1, clf, hmf, csf, twh, ntf, sibling_log_g(p.log_k), st);
        if (rcf) return rcf < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: internal: no Hermitian column kernel for %lld points", (long long)H) : rcf;
        RowLoadFold<T> rlf{W, plane, H, ltl, twm, 0, 0};
        RowStoreNat<T> rof{reinterpret_cast<cx<T>*>(out), d->out_ld / 2, AxisMap{int(n2), int(n2), 0, int(d->out_x.shift / 2)}, int(M), 1,
                           T(d->scale), 1, to_map(d->out_y), 0, H};
        rof.nt = row_store_nt(size_t(M) * size_t(N) * sizeof(T));
        rcf = launch_row_c2r_fold<T>(p.logn - 1, rlf, rof, tw2, twn, H, st);
        return rcf < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: internal: no folded half-spectrum row kernel for %lld points", (long long)N) : rcf;
    }
    int rc = launch_row_r2c<T>(p.logn - 1, lp, rs, tw2, int(M), tuning().row_log_g, st);
    if (rc) return rc < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: internal: no Hermitian row kernel for %lld points", (long long)N) : rc;
    // columns: transform, x the Hermitian part of H, inverse transform, in place
    const int ntiles = int(n2 / p.tc);
    ColLoadTiled<T> cl{W, int(M), to_map(d->in_y), ntiles, p.log_k, 0};
    HermMul<T> hm{reinterpret_cast<const cx<T>*>(d->mul), d->mul_ld, int(M), int(N), d->mul_conj ? 1 : 0, 0};
    ColStoreTiled<T> cst{W, int(M), ntiles, p.log_k, 0};
    rc = launch_col_mul_herm<T>(p.logm, cl, hm, cst, twm, ntiles, sibling_log_g(p.log_k), st);
    if (rc) return rc < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: internal: no Hermitian column kernel for %lld points", (long long)M) : rc;
    // rows back: half spectra -> N real samples per row = N/2 complex elements of the output seen as complex
    RowLoadTiled<T> rl{W, int(M), ltl, 0, int(M), 0, 0};
    RowStoreNat<T> ro{reinterpret_cast<cx<T>*>(out), d->out_ld / 2, AxisMap{int(n2), int(n2), 0, int(d->out_x.shift / 2)}, int(M), 1,
                      T(d->scale), 1, to_map(d->out_y), 0, 0};
    ro.nt = row_store_nt(size_t(M) * size_t(N) * sizeof(T));
    rc = launch_row_c2r<T>(p.logn - 1, rl, ro, tw2, twn, int(M), st);
    return rc < 0 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: internal: no half-spectrum row kernel for %lld points", (long long)N) : rc;
}

// ---- the wavelength loop as launch pairs over groups of wavelengths (fft_spectral.h)
// fast form: complex64 packed synthesis, |.|^2 accumulation, both lengths on the engine with a tiled intermediate, fewer than 4096^2
// bins.  Measured (profiles/r02/exp_spectral.log, us per wavelength, loop -> groups of 8): 1024^2 24.7 -> 11.0, 2048^2 40.6 -> 22.6,
// 1024^2 padded to 2048^2 34.0 -> 13.3; at 4096^2 the loop's passes already run at 84 % of copy speed with their intermediate in the
// Infinity Cache, and the grouped kernels pay for their registers with occupancy (rocprofv3: 47.6 + 50.7 us per wavelength against
// 45.0 + 55.9 in groups of 8), so those sizes keep the loop.
static bool spectral_fast(const pm_fft2_desc* d, const Fft2Plan& p) {
    const int want = PM_FLAG_SYNTH_INPUT | PM_FLAG_SYNTH_PACKED;
    // complex128: rows of up to 2048 samples, unfolded (the grouped double-precision kernels exist for those; profiles/r02/exp_spectral_c128.log)
    if (d->dtype == PM_C128 && (p.logn > 11 || p.fold)) return false;
    return tuning().spectral > 1 && (d->flags & want) == want && !(d->flags & (PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY)) &&
           d->epilogue == PM_EPI_ABS2_ACCUM && d->batch <= 1 && d->mul_kind == PM_MUL_NONE && !p.r2c && !p.big_rn && !p.blue2d &&
           p.logn >= 5 && p.logn <= 12 /* its row kernel spills hundreds of registers at 8192-point rows (two rows per thread + the packed map) */ &&
           p.logm >= (p.fold ? 6 : 5) && p.logm - (p.fold ? 1 : 0) <= 11 /* the accumulating column kernel spills beyond 2048-point tiles */ &&
           p.tc != 0 && p.col_var == 0 && d->in_y.len > 0 && p.logn + p.logm < tuning().spectral_area_log;
}
static int spectral_group(int32_t count) {
    int g = tuning().spectral;
    if (g > kSpectralMax) g = kSpectralMax;
    return g < count ? g : count;
}

template <typename T>
static int fft2_spectral_group(const pm_fft2_desc* d, const Fft2Plan& p, const Spectral& w, const void* in, void* out, void* ws, hipStream_t st) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    const int rows = int(d->in_y.len);
    int err = 0;
    cx<T>* W = reinterpret_cast<cx<T>*>(ws);
    const cx<T>* tw = twiddles<T>(N, &err);
    if (!tw) return err;
    const size_t in_bytes = size_t(rows) * size_t(d->in_x.len) * sizeof(cx<T>);
    const int nt_in = tuning().nt_in >= 0 ? tuning().nt_in : (in_bytes >= (size_t(96) << 20) ? 1 : 0);
    RowLoadNat<T> lp{reinterpret_cast<const cx<T>*>(in), d->in_ld, to_map(d->in_x), rows, 0, nt_in, 0};
    int ltc = 0;
    while ((1 << ltc) < (p.tc << p.log_k)) ++ltc;
    const int64_t tl = int64_t(1) << ltc, ntl = (N + tl - 1) / tl;
    ColStoreNat<T> cs = make_colstore<T>(d, out, p.fold ? p.logm - 1 : -1);
    const int ntiles = int((N + p.tc - 1) / p.tc);
    int rc;
    if (p.fold) {
        const int H = int(M / 2);
        const cx<T>* twm = twiddles<T>(M, &err);
        if (!twm) return err;
        const cx<T>* twh = twiddles<T>(H, &err);
        if (!twh) return err;
        lp.eoff = H;
        RowStoreFold<T> sp{W, ntl * H * tl, H, ltc, twm, d->in_y.shift == M / 2 ? 1 : 0, 0};
        if ((rc = launch_row_spectral_fold<T>(p.logn, lp, sp, tw, H, w, st))) return rc;
        ColLoadTiled<T> cl{W, H, AxisMap{H, H, 0, 0}, ntiles, p.log_k, ntl * H * tl};
        cs.ay = AxisMap{H, int(d->out_y.len / 2), int(d->out_y.off / 2), int(d->out_y.shift / 2)};
        cs.bstride = d->out_ld;
        cs.ld = 2 * d->out_ld;
        return launch_col_spectral<T>(p.logm - 1, cl, cs, twh, ntiles, sibling_log_g(p.log_k), w, st, 2);
    }
    const cx<T>* twm = twiddles<T>(M, &err);
    if (!twm) return err;
    RowStoreTiled<T> sp{W, rows, ltc, 0};
    if ((rc = launch_row_spectral<T>(p.logn, row_variant(d->dtype, p.logn), lp, sp, tw, rows, tuning().row_log_g, w, st))) return rc;
    ColLoadTiled<T> cl{W, rows, to_map(d->in_y), ntiles, p.log_k, 0};
    return launch_col_spectral<T>(p.logm, cl, cs, twm, ntiles, sibling_log_g(p.log_k), w, st, 1);
}


}  // namespace pm

using namespace pm;

extern "C" {

int pm_version(void) { return PM_VERSION; }

// Variants that measured slower were built behind -DPM_EXPERIMENTS through round 4 and left the sources in round 5 (experiments/README.md
// has the list, the logs and the patch that brings them back): the library refuses the knob values that selected them instead of
// silently running something else.
static bool experiment_only(const char* key, int v) {
    auto is = [&](const char* k) { return !strcmp(key, k); };
    return (is("spectral_mode") && (v & 3) != 3) || (is("gemm_3m") && !v) || (is("gemm_bm") && v == 128) || (is("gemm_bk") && v == 32) ||
           (is("colmul_mode") && (v == 1 || v == 2)) || (is("gemm_wk") && (v & 6)) || (is("spectral2") && v != 0) || (is("two_units") && v != 0) || (is("engine_p8") && v != 0) ||
           (is("mix_ablate") && v != 0) || (is("mix_pers") && v != 0) || (is("mix_fold") && v != 0);
}

int pm_set_tuning(const char* key, int32_t value) {
    if (!key) return fail(PM_ERR_ARG, "pm_set_tuning: null key");
    if (experiment_only(key, value))
        return fail(PM_ERR_UNSUPPORTED, "pm_set_tuning: %s = %d selects a variant that lost its measurement and is no longer in the library (experiments/README.md)", key,
                    int(value));
    static std::mutex mu;      // writers of the process-wide defaults are serialised; a thread that needs its own values while others
    std::lock_guard<std::mutex> lk(mu);     // run takes pm_set_tuning_local
    tune_set(tuning_global(), key, strlen(key), value);
    return 0;
}
int pm_set_tuning_local(const char* key, int32_t value) {
    if (!key) return fail(PM_ERR_ARG, "pm_set_tuning_local: null key");
    if (experiment_only(key, value))
        return fail(PM_ERR_UNSUPPORTED, "pm_set_tuning_local: %s = %d selects a variant that lost its measurement and is no longer in the library (experiments/README.md)",
                    key, int(value));
    if (!g_tune_local_on) {
        g_tune_local = tuning_global();     // the thread's copy starts from the defaults of this moment
        g_tune_local_on = true;
    }
    tune_set(g_tune_local, key, strlen(key), value);
    return 0;
}
void pm_reset_tuning_local(void) { g_tune_local_on = false; }
const char* pm_last_error(void) { return g_err; }

int pm_plan_prepare(int32_t dtype, int64_t n) {
    if (n < 1) return fail(PM_ERR_ARG, "pm_plan_prepare: n < 1");
    int err = 0;
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_plan_prepare: dtype");
    if (big_split(n) > 1) {   // one radix-R step around engine transforms (bigfft.hip): the tables of n and of n / R
        const int64_t part = n / big_split(n);
        const bool ok = dtype == PM_C64 ? (twiddles<float>(n, &err) && twiddles<float>(part, &err))
                                        : (twiddles<double>(n, &err) && twiddles<double>(part, &err));
        if (!ok) return err;
        if (engine_log2(n) >= 0 || !use_blue_long(n)) return 0;     // a mixed-radix length beside an awkward one still takes Bluestein
    }
    if (big_split(n) == 0 && big_split2d(n) > 1) {     // a composite above 8192 (2-D transforms): the tables of n and of its mixed-radix cofactor
        const int64_t part = n / big_split2d(n);
        const bool ok = dtype == PM_C64 ? (twiddles<float>(n, &err) && twiddles<float>(part, &err)) : (twiddles<double>(n, &err) && twiddles<double>(part, &err));
        if (!ok) return err;
        if (engine_log2(part) < 0 && !mix_plan_dev(int(part), dtype == PM_C64 ? 8 : 16, &err)) return err;
        return 0;
    }
    if (use_mix(n)) {
        if (!mix_plan_dev(int(n), dtype == PM_C64 ? 8 : 16, &err)) return err;
        return (dtype == PM_C64 ? (const void*)twiddles<float>(n, &err) : (const void*)twiddles<double>(n, &err)) ? 0 : err;
    }
    if (use_blue_long(n)) {   // Bluestein tables of n and the twiddles of the convolution length (of its engine part when it is split)
        const int64_t mb = blue_conv_len(n), part = mb / big_split(mb);
        if (dtype == PM_C64) return (blue_tables<float>(n, &err) && twiddles<float>(mb, &err) && twiddles<float>(part, &err)) ? 0 : err;
        return (blue_tables<double>(n, &err) && twiddles<double>(mb, &err) && twiddles<double>(part, &err)) ? 0 : err;
    }
    if (engine_log2(n) < 0) return twiddles_f64(n, &err) ? 0 : err;
    if (dtype == PM_C64) return twiddles<float>(n, &err) ? 0 : err;
    if (dtype == PM_C128) return twiddles<double>(n, &err) ? 0 : err;
    return fail(PM_ERR_ARG, "pm_plan_prepare: dtype");
}

void pm_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_tables) (void)hipFree(kv.second);
    g_tables.clear();
}

size_t pm_fft2_workspace(const pm_fft2_desc* d) {
    if (check_fft2(d)) return 0;
    const Fft2Plan p = plan_fft2(d);
    if (!p.r2c) return p.ws_bytes;
    // the Hermitian path reads the real array as complex pairs: a base address that is not aligned like a complex element sends the
    // call down the complex path instead (pm_fft2 below), whose intermediate is larger -- the query covers both
    const size_t other = plan_fft2(d, false).ws_bytes;
    return other > p.ws_bytes ? other : p.ws_bytes;
}

int pm_fft2(const pm_fft2_desc* d, const void* in, void* out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_fft2(d);
    if (rc) return rc;
    if (!in || !out) return fail(PM_ERR_ARG, "pm_fft2: null buffer");
    Fft2Plan p = plan_fft2(d);
    if (p.r2c && reinterpret_cast<uintptr_t>(in) % (d->dtype == PM_C64 ? 8 : 16) != 0) {
        // the Hermitian path reads the real array as pairs; the complex path that takes over has neither the centre normalisation nor
        // the |.| / angle epilogues (check_fft2 accepted them on the strength of r2c_legal): refuse, do not run something else
        if ((d->flags & PM_FLAG_NORM_DC) || d->epilogue == PM_EPI_ABS || d->epilogue == PM_EPI_ARG)
            return fail(PM_ERR_UNSUPPORTED, "pm_fft2: PM_FLAG_NORM_DC / PM_EPI_ABS / PM_EPI_ARG read the real array as pairs; its base address "
                        "must be aligned like a complex element (%d bytes)", d->dtype == PM_C64 ? 8 : 16);
        p = plan_fft2(d, false);
    }
    if (!workspace || workspace_bytes < p.ws_bytes)
        return fail(PM_ERR_WORKSPACE, "pm_fft2: workspace of %zu bytes required, %zu given", p.ws_bytes, workspace_bytes);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == PM_C64) return fft2_run<float>(d, p, in, out, workspace, st);
    return fft2_run<double>(d, p, in, out, workspace, st);
}

// One line that says which route a descriptor takes -- the planner's decisions (plan_fft2 / plan_fused / herm_conv_plan under the
// calling thread's tuning knobs) in words.  Host logic only: callable without a GPU, so the routes of a table of shapes are pinned by
// a CPU test (tests/test_host_logic.py) and a shape that falls to a slow route shows up there and not as a timing.
static const char* axis_route(bool engine, bool mix, bool blue) { return engine ? "stockham" : (mix ? "mixed-radix" : (blue ? "bluestein" : "direct")); }
int pm_plan_explain(const pm_fft2_desc* d, int32_t op, char* buf, size_t n) {
    if (!buf || n < 64) return fail(PM_ERR_ARG, "pm_plan_explain: a buffer of at least 64 bytes is required");
    buf[0] = 0;
    int rc = check_fft2(d);
    if (rc) return rc;
    const long long M = d->in_y.n, N = d->in_x.n;
    const char* dt = d->dtype == PM_C64 ? "c64" : "c128";
    if (op == 1) {
        if (d->flags & PM_FLAG_REAL_OUTPUT) {
            HermConvPlan hp;
            if (!herm_conv_plan(d, hp)) { snprintf(buf, n, "fft2_mul_ifft2 %lldx%lld %s: route=unsupported (real output needs the Hermitian chain)", M, N, dt); return 0; }
            snprintf(buf, n, "fft2_mul_ifft2 %lldx%lld %s: route=hermitian-chain passes=3 ws=%zu", M, N, dt, hp.ws_bytes);
            return 0;
        }
        FusedPlan p;
        if (!plan_fused(d, p)) { snprintf(buf, n, "fft2_mul_ifft2 %lldx%lld %s: route=composed (two fft2 calls)", M, N, dt); return 0; }
        snprintf(buf, n, "fft2_mul_ifft2 %lldx%lld %s: route=%s passes=3 rows=%s mid=%s%s ws=%zu", M, N, dt, p.mixmid ? "fused-composite" : "fused",
                 axis_route(p.logn >= 0, !(p.logn >= 0), false), p.mixmid ? "mixed-radix-resident" : "stockham-pair", p.fold ? " fold" : "", p.ws_bytes);
        return 0;
    }
    if (op != 0) return fail(PM_ERR_ARG, "pm_plan_explain: op must be 0 (pm_fft2) or 1 (pm_fft2_mul_ifft2)");
    const Fft2Plan p = plan_fft2(d);
    if (p.big_rn) {
        const long long np_ = N / p.big_rn, mp_ = M / p.big_rm;
        snprintf(buf, n, "fft2 %lldx%lld %s: route=radix-step rows=%dx%s(%lld) cols=%dx%s(%lld) ws=%zu", M, N, dt, p.big_rn,
                 engine_log2(np_) >= 0 ? "stockham" : "mixed-radix", np_, p.big_rm, engine_log2(mp_) >= 0 ? "stockham" : "mixed-radix", mp_, p.ws_bytes);
    } else if (p.blue2d) {
        snprintf(buf, n, "fft2 %lldx%lld %s: route=%s conv=%lldx%lld ws=%zu", M, N, dt, p.blue_big ? "bluestein-2d-big" : "bluestein-2d",
                 (long long)blue_conv_len(M), (long long)blue_conv_len(N), p.ws_bytes);
    } else if (p.r2c_t) {
        snprintf(buf, n, "fft2 %lldx%lld %s: route=hermitian-transposed cols=stockham-r2c(%lld) rows=stockham(%lld)x%lld ws=%zu", M, N, dt, M, N, M / 2, p.ws_bytes);
    } else if (p.r2c) {
        snprintf(buf, n, "fft2 %lldx%lld %s: route=hermitian%s rows=stockham-r2c(%lld) cols=stockham(%lld%s) tile=%d log_k=%d ws=%zu", M, N, dt,
                 p.fold ? "-fold" : "", N / 2, p.fold ? M / 2 : M, p.fold ? "x2" : "", p.tc, p.log_k, p.ws_bytes);
    } else {
        const bool en = p.logn >= 0, em = p.logm >= 0;
        // a composite axis whose length has a compile-time plan runs on the register engine (fft_ce.h) when the view is plain
        const bool f32 = d->dtype == PM_C64;
        const size_t es = f32 ? 8 : 16;
        const bool ce_n = p.mix_n && !p.mix_fold && tuning().mix_engine && !(d->flags & PM_FLAG_REAL_INPUT) && (f32 || !(d->flags & PM_FLAG_SYNTH_INPUT)) &&
                          (f32 ? ce_has_plan<float>(int(N)) : ce_has_plan<double>(int(N))) && ce_fits32(kCeMaxSeqs * d->in_ld + 2 * N, es) &&
                          ce_fits32(kCeMaxSeqs * p.w_ld + 2 * N, es);
        const bool whole_out = d->out_y.off == 0 && d->out_y.len == M && d->out_x.off == 0 && d->out_x.len == N;
        const bool ce_m = p.mix_m && !p.mix_fold && tuning().mix_engine && whole_out && !d->mul && d->epilogue <= PM_EPI_ABS2_ACCUM &&
                          (f32 ? ce_has_plan<float>(int(M)) : ce_has_plan<double>(int(M))) && ce_fits32(2 * M * p.w_ld + kCeMaxSeqs, es) &&
                          ce_fits32(2 * M * d->out_ld + N, es);
        snprintf(buf, n, "fft2 %lldx%lld %s: route=%s rows=%s(%lld) cols=%s(%lld%s) tile=%d log_k=%d chunk=%lld ws=%zu", M, N, dt,
                 (en && em) ? (p.fold ? "engine-fold" : "engine") : ((p.mix_n || !p.blue_n) && (p.mix_m || !p.blue_m) && (p.mix_n || p.mix_m) ? "natural-mixed" : "natural"),
                 ce_n ? "mixed-radix-registers" : axis_route(en, p.mix_n, p.blue_n), N, ce_m ? "mixed-radix-registers" : axis_route(em, p.mix_m, p.blue_m),
                 p.fold ? M / 2 : M, p.fold ? "x2" : "", p.tc, p.log_k, (long long)p.chunk, p.ws_bytes);
    }
    return 0;
}

size_t pm_fft2_spectral_workspace(const pm_fft2_desc* d, int32_t count) {
    if (check_fft2(d) || count <= 0) return 0;
    const Fft2Plan p = plan_fft2(d);
    size_t need = spectral_fast(d, p) ? p.ws_field * size_t(spectral_group(count)) : p.ws_bytes;
    return need;
}

int pm_fft2_spectral(const pm_fft2_desc* d, int32_t count, const double* k, const double* weight, const void* in, void* out, void* workspace,
                     size_t workspace_bytes, void* stream) {
    int rc = check_fft2(d);
    if (rc) return rc;
    if (count < 0) return fail(PM_ERR_ARG, "pm_fft2_spectral: count < 0");
    if (count == 0) return 0;
    if (!in || !out || !k || !weight) return fail(PM_ERR_ARG, "pm_fft2_spectral: null buffer");
    if (!(d->flags & PM_FLAG_SYNTH_INPUT) || d->epilogue != PM_EPI_ABS2_ACCUM)
        return fail(PM_ERR_ARG, "pm_fft2_spectral: the descriptor must ask for PM_FLAG_SYNTH_INPUT and PM_EPI_ABS2_ACCUM");
    const Fft2Plan p = plan_fft2(d);
    const size_t need = pm_fft2_spectral_workspace(d, count);
    if (!workspace || workspace_bytes < need)
        return fail(PM_ERR_WORKSPACE, "pm_fft2_spectral: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!spectral_fast(d, p)) {     // the loop itself: one transform pair per wavelength
        pm_fft2_desc dd = *d;
        for (int32_t b = 0; b < count; ++b) {
            dd.synth_k = k[b];
            dd.weight = weight[b];
            rc = d->dtype == PM_C64 ? fft2_run<float>(&dd, p, in, out, workspace, st) : fft2_run<double>(&dd, p, in, out, workspace, st);
            if (rc) return rc;
        }
        return 0;
    }
    const int g = spectral_group(count);
    const double two_pi = 2.0 * 3.14159265358979323846264338327950288;
    for (int32_t b0 = 0; b0 < count; b0 += g) {
        Spectral w{};
        w.nb = count - b0 < g ? count - b0 : g;
        w.fstride = int64_t(p.ws_field / (d->dtype == PM_C64 ? sizeof(cx<float>) : sizeof(cx<double>)));
        w.mode = tuning().spectral_mode;
        for (int i = 0; i < w.nb; ++i) {
            w.w[i] = weight[b0 + i];
            w.k2[i] = k[b0 + i] / two_pi;
        }
        rc = d->dtype == PM_C64 ? fft2_spectral_group<float>(d, p, w, in, out, workspace, st)
                                : fft2_spectral_group<double>(d, p, w, in, out, workspace, st);
        if (rc)
            return rc == -2 ? fail(PM_ERR_UNSUPPORTED, "pm_fft2_spectral: internal: no grouped kernel for %lld x %lld", (long long)d->in_y.n, (long long)d->in_x.n) : rc;
    }
    return 0;
}

size_t pm_fft2_mul_ifft2_workspace(const pm_fft2_desc* d) {
    if (check_fft2(d)) return 0;
    if (d->flags & PM_FLAG_REAL_OUTPUT) {
        HermConvPlan hp;
        return herm_conv_plan(d, hp) ? hp.ws_bytes : 0;
    }
    FusedPlan p;
    if (!plan_fused(d, p)) return 0;
    return p.ws_bytes;
}

int pm_fft2_mul_ifft2(const pm_fft2_desc* d, const void* in, void* out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_fft2(d);
    if (rc) return rc;
    if (!in || !out) return fail(PM_ERR_ARG, "pm_fft2_mul_ifft2: null buffer");
    if (d->mul_kind == PM_MUL_NONE) return fail(PM_ERR_ARG, "pm_fft2_mul_ifft2: a multiplier is required");
    if (d->flags & PM_FLAG_REAL_OUTPUT) {
        HermConvPlan hp;
        if (!herm_conv_plan(d, hp))
            return fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: PM_FLAG_REAL_OUTPUT needs a real unpadded power-of-two field (rows of 64 .. 8192 "
                        "samples), a full multiplier, x rotations by 0 or N/2 and an unwindowed output; take the real part of the complex "
                        "chain instead");
        if (reinterpret_cast<uintptr_t>(in) % (d->dtype == PM_C64 ? 8 : 16) != 0 || reinterpret_cast<uintptr_t>(out) % (d->dtype == PM_C64 ? 8 : 16) != 0)
            return fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: PM_FLAG_REAL_OUTPUT reads and writes the real arrays as pairs; their base "
                        "addresses must be aligned like a complex element (take the real part of the complex chain instead)");
        if (!workspace || workspace_bytes < hp.ws_bytes)
            return fail(PM_ERR_WORKSPACE, "pm_fft2_mul_ifft2: workspace of %zu bytes required, %zu given", hp.ws_bytes, workspace_bytes);
        hipStream_t hst = reinterpret_cast<hipStream_t>(stream);
        if (d->dtype == PM_C64) return herm_conv_run<float>(d, hp, in, out, workspace, hst);
        return herm_conv_run<double>(d, hp, in, out, workspace, hst);
    }
    FusedPlan p;
    if (!plan_fused(d, p))
        return fail(PM_ERR_UNSUPPORTED, "pm_fft2_mul_ifft2: takes powers of two <= 8192 on both axes, or (one field, complex output) a composite "
                    "column length with primes <= 19 beside a row length of either kind (got %lld x %lld); compose two pm_fft2 calls instead",
                    (long long)d->in_y.n, (long long)d->in_x.n);
    const size_t need = p.ws_bytes;
    if (!workspace || workspace_bytes < need)
        return fail(PM_ERR_WORKSPACE, "pm_fft2_mul_ifft2: workspace of %zu bytes required, %zu given", need, workspace_bytes);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d->dtype == PM_C64) return fused_run<float>(d, p, in, out, workspace, st);
    return fused_run<double>(d, p, in, out, workspace, st);
}

int pm_fft2_time_passes(const pm_fft2_desc* d, const void* in, void* out, void* workspace, size_t workspace_bytes,
                        int reps, double* ms, void* stream) {
    // Average duration of each of the two kernels of a propagation, HIP events on the launch stream.  Each pass is timed as ITS OWN
    // back-to-back loop of `reps` launches between one pair of events (round 4; rounds 1 - 3 recorded an event between the two
    // kernels of every propagation, which serialises them: row + column exceeded the step time by 7 %).  A loop of one
    // kernel keeps the launch pipeline full exactly as the real alternating sequence does, so ms[0] + ms[1] = the step time within a
    // few per cent; what it does not reproduce is the other pass's footprint in the caches (the row pass re-reads its input and
    // rewrites the intermediate every launch, the column pass re-reads the intermediate) -- bench.py prints the ratio to the step time.
    // ms[0] = row pass, ms[1] = column pass.
    if (!ms || reps < 1 || reps > 4096) return fail(PM_ERR_ARG, "pm_fft2_time_passes: reps must be in [1, 4096]");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipEvent_t ev[4];
    for (auto& e : ev) {
        hipError_t he = hipEventCreate(&e);
        if (he != hipSuccess) return int(he);
    }
    pm_fft2_desc dd = *d;
    const int32_t keep = d->flags & (PM_FLAG_REAL_INPUT | PM_FLAG_SYNTH_INPUT | PM_FLAG_SYNTH_PACKED);
    dd.flags = keep;
    int rc = pm_fft2(&dd, in, out, workspace, workspace_bytes, stream);   // warm (also builds the plan)
    for (int pass = 0; pass < 2 && !rc; ++pass) {
        dd.flags = keep | (pass == 0 ? PM_FLAG_PASS1_ONLY : PM_FLAG_PASS2_ONLY);
        for (int i = 0; i < 3 && !rc; ++i) rc = pm_fft2(&dd, in, out, workspace, workspace_bytes, stream);
        (void)hipEventRecord(ev[2 * pass], st);
        for (int i = 0; i < reps && !rc; ++i) rc = pm_fft2(&dd, in, out, workspace, workspace_bytes, stream);
        (void)hipEventRecord(ev[2 * pass + 1], st);
    }
    ms[0] = ms[1] = 0.0;
    if (!rc) {
        (void)hipEventSynchronize(ev[3]);
        for (int pass = 0; pass < 2; ++pass) {
            float a = 0.f;
            (void)hipEventElapsedTime(&a, ev[2 * pass], ev[2 * pass + 1]);
            ms[pass] = double(a) / reps;
        }
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

static int check_fft1(int32_t dtype, int32_t direction, int32_t axis, int64_t batch, const pm_axis* t_in, const pm_axis* t_out) {
    if (!t_in || !t_out) return fail(PM_ERR_ARG, "pm_fft1: null argument");
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_fft1: dtype must be PM_C64 or PM_C128");
    if (direction != 1 && direction != -1) return fail(PM_ERR_ARG, "pm_fft1: direction must be -1 or +1");
    if (axis != 0 && axis != 1) return fail(PM_ERR_ARG, "pm_fft1: axis must be 0 or 1");
    if (batch < 0) return fail(PM_ERR_ARG, "pm_fft1: batch < 0");
    int rc;
    if ((rc = check_axis(*t_in, "t_in")) || (rc = check_axis(*t_out, "t_out"))) return rc;
    if (t_in->n != t_out->n) return fail(PM_ERR_ARG, "pm_fft1: t_in.n != t_out.n");
    if (engine_log2(t_in->n) < 0 && t_in->n > (int64_t(1) << 15))
        return fail(PM_ERR_UNSUPPORTED, "pm_fft1: length %lld not supported", (long long)t_in->n);
    return 0;
}

int pm_czt_axis(int32_t dtype, int32_t axis, int64_t nseq, int64_t K, int64_t in_len, int64_t in_off, int64_t out_len, int64_t out_off,
                const void* pre, int32_t pre_conj, const void* H, int32_t h_conj, const void* post, int32_t post_conj, double scale,
                const void* in, int64_t in_ld, void* out, int64_t out_ld, void* stream) {
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_czt_axis: dtype must be PM_C64 or PM_C128");
    if (axis != 0 && axis != 1) return fail(PM_ERR_ARG, "pm_czt_axis: axis must be 0 or 1");
    if (!in || !out || !H) return fail(PM_ERR_ARG, "pm_czt_axis: null argument");
    if (nseq < 0 || in_len < 0 || out_len < 0 || in_off < 0 || out_off < 0 || in_off + in_len > K || out_off + out_len > K)
        return fail(PM_ERR_ARG, "pm_czt_axis: the input and output windows must lie inside [0, K)");
    const int lg = engine_log2(K);
    if (lg < 4) return fail(PM_ERR_UNSUPPORTED, "pm_czt_axis: K = %lld must be a power of two from 16 to 8192 (compose pm_fft1 and "
                            "pm_scale_sep otherwise)", (long long)K);
    if (nseq == 0 || out_len == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int rc = dtype == PM_C64
                 ? czt_axis_run<float>(axis, nseq, K, in_len, in_off, out_len, out_off, pre, pre_conj, H, h_conj, post, post_conj, scale, in,
                                       in_ld, out, out_ld, st)
                 : czt_axis_run<double>(axis, nseq, K, in_len, in_off, out_len, out_off, pre, pre_conj, H, h_conj, post, post_conj, scale,
                                        in, in_ld, out, out_ld, st);
    return rc == -2 ? fail(PM_ERR_UNSUPPORTED, "pm_czt_axis: no kernel for K = %lld", (long long)K) : rc;
}

int pm_fft1_ramp(int32_t dtype, int32_t direction, int32_t axis, int64_t nseq, int64_t K, int64_t in_len, int64_t in_off, int64_t out_len,
                 int64_t out_off, const void* pre, int32_t pre_conj, const void* post, int32_t post_conj, double scale, const void* in,
                 int64_t in_ld, void* out, int64_t out_ld, void* stream) {
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_fft1_ramp: dtype must be PM_C64 or PM_C128");
    if (axis != 0 && axis != 1) return fail(PM_ERR_ARG, "pm_fft1_ramp: axis must be 0 or 1");
    if (direction != -1 && direction != 1) return fail(PM_ERR_ARG, "pm_fft1_ramp: direction must be -1 or +1");
    if (!in || !out) return fail(PM_ERR_ARG, "pm_fft1_ramp: null argument");
    if (nseq < 0 || in_len < 0 || out_len < 0 || in_off < 0 || out_off < 0 || in_off + in_len > K || out_off + out_len > K)
        return fail(PM_ERR_ARG, "pm_fft1_ramp: the input and output windows must lie inside [0, K)");
    const int lg = engine_log2(K);
    if (lg < 4) return fail(PM_ERR_UNSUPPORTED, "pm_fft1_ramp: K = %lld must be a power of two from 16 to 8192 (compose pm_fft1 and "
                            "pm_scale_sep otherwise)", (long long)K);
    if (nseq == 0 || out_len == 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int single = direction < 0 ? 1 : 2;
    int rc = dtype == PM_C64
                 ? czt_axis_run<float>(axis, nseq, K, in_len, in_off, out_len, out_off, pre, pre_conj, nullptr, 0, post, post_conj, scale, in,
                                       in_ld, out, out_ld, st, single)
                 : czt_axis_run<double>(axis, nseq, K, in_len, in_off, out_len, out_off, pre, pre_conj, nullptr, 0, post, post_conj, scale,
                                        in, in_ld, out, out_ld, st, single);
    return rc == -2 ? fail(PM_ERR_UNSUPPORTED, "pm_fft1_ramp: no kernel for K = %lld", (long long)K) : rc;
}

int pm_fft1(int32_t dtype, int32_t direction, int32_t axis, int64_t batch, const pm_axis* t_in, const pm_axis* t_out,
            double scale, const void* in, int64_t in_ld, void* out, int64_t out_ld, void* stream) {
    return pm_fft1_ws(dtype, direction, axis, batch, t_in, t_out, scale, in, in_ld, out, out_ld, nullptr, 0, stream);
}

size_t pm_fft1_workspace(int32_t dtype, int32_t axis, int64_t batch, int64_t n) {
    if ((dtype != PM_C64 && dtype != PM_C128) || batch <= 0) return 0;
    const size_t es = dtype == PM_C64 ? 8 : 16;
    if (big_split(n) > 1) return fft1_big_scratch(es, axis, batch, n);     // 16384 / 32768 and 3 / 5 / 7 x 2^k: a radix-R step
    if (!use_blue(n)) return 0;
    return axis == 1 ? blue_rows_scratch(es, batch, n) : blue_cols_scratch(es, batch, n);
}

int pm_fft1_ws(int32_t dtype, int32_t direction, int32_t axis, int64_t batch, const pm_axis* t_in, const pm_axis* t_out,
               double scale, const void* in, int64_t in_ld, void* out, int64_t out_ld, void* workspace, size_t workspace_bytes,
               void* stream) {
    int rc = check_fft1(dtype, direction, axis, batch, t_in, t_out);
    if (rc) return rc;
    if (!in || !out) return fail(PM_ERR_ARG, "pm_fft1: null argument");
    if (batch == 0) return 0;
    // a workspace of pm_fft1_workspace() bytes puts lengths above 8192 and mixed-radix lengths on the radix-R path, other
    // non-power-of-two lengths on the Bluestein path; without one they run on the direct O(n^2) kernel.  (A rotated input view of a
    // radix-R length has no Bluestein scratch in that workspace either: it runs direct.)
    const size_t need = pm_fft1_workspace(dtype, axis, batch, t_in->n);
    void* bws = (need && workspace && workspace_bytes >= need && reinterpret_cast<uintptr_t>(workspace) % 256 == 0) ? workspace : nullptr;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == PM_C64) return fft1_run<float>(direction, axis, batch, t_in, t_out, scale, in, in_ld, out, out_ld, st, bws);
    return fft1_run<double>(direction, axis, batch, t_in, t_out, scale, in, in_ld, out, out_ld, st, bws);
}

}  // extern "C"
