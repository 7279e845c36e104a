// extern "C" transform entry points of libprysm_amd.so: check, plan (capi_plan.hip), run (capi_run.hip).
#include "capi_internal.h"

using namespace pm;

extern "C" {

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
