// Error text, twiddle / plan tables behind a mutex, tuning knobs (process-wide and per thread), device queries -- and the entry points that only touch those.
#include "capi_internal.h"

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

}  // extern "C"
