// Host-only planning (no GPU work, callable on a box without one): legality of each route, the plan of a 2-D transform / fused chain / Hermitian chain, workspace sizes, argument checks, pm_plan_explain.
#include "capi_internal.h"

namespace pm {

// sibling group of column-pass workgroups: the tiles of one layout-tile row, at most 8
// (knob col_log_g >= 0 overrides: up to 2^5 = the 32 workgroups an XCD's CUs hold at one per CU -- experiments of round 5)
int sibling_log_g(int log_k) {
    if (tuning().col_log_g >= 0) return tuning().col_log_g > 5 ? 5 : tuning().col_log_g;
    return log_k < 1 ? 1 : (log_k > 3 ? 3 : log_k);
}


static int check_axis(const pm_axis& a, const char* name) {
    if (a.n < 1 || a.n > (int64_t(1) << 30)) return fail(PM_ERR_ARG, "%s.n = %lld out of range", name, (long long)a.n);
    if (a.len < 0 || a.len > a.n) return fail(PM_ERR_ARG, "%s.len = %lld must be in [0, n]", name, (long long)a.len);
    if (a.off < 0 || a.off + a.len > a.n) return fail(PM_ERR_ARG, "%s window [off, off+len) must lie in [0, n]", name);
    if (a.shift < 0 || a.shift >= a.n) return fail(PM_ERR_ARG, "%s.shift must be in [0, n)", name);
    return 0;
}

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
// transposed): fp32 128^2 21.1 / 17.0, 1024^2 30.3 / 19.8, 2048^2 38.8 / 32.5, 4096 x 1024 52.2 / 31.7, 4096^2 71.9 / 59.7,
// 8192 x 2048 80.7 / 69.1 -- and 2048 x 8192 64.9 / 70.5, 4096 x 8192 122.9 / 155.4, 8192 x 4096 140.0 / 158.5: rows of 8192 samples
// stay on the round-2 form, columns of 8192 beyond 2048 rows' width too; fp64 1024^2 29.4 / 22.3, 2048^2 41.8 / 40.1, 4096 x 2048
// 67.8 / 58.4 -- and 2048 x 4096 64.7 / 76.5, 4096^2 117.8 / 156.1, 8192 x 2048 138.2 / 155.7: rows of 4096 complex128 points stay too.
static bool hermt_legal(const pm_fft2_desc* d, int logn, int logm) {
    const int64_t M = d->in_y.n, N = d->in_x.n;
    const int ht = tuning().herm_t;
    if (ht == 0) return false;
    if (logm < 5 || logm > 13 || logn < 5 || logn > (d->dtype == PM_C64 ? 13 : 12)) return false;
    if (logm == 13 && tuning().herm_t_fold == 0) return false;     // 8192-point columns exist as planes of 4096-point tiles only
    if (ht < 0 && (logn > (d->dtype == PM_C64 ? 12 : 11) || logm > (d->dtype == PM_C64 && logn <= 11 ? 13 : 12))) return false;
    if (!(d->in_y.shift == 0 || d->in_y.shift == M / 2) || !(d->out_y.shift == 0 || d->out_y.shift == M / 2) ||
        !(d->out_x.shift == 0 || d->out_x.shift == N / 2))
        return false;
    return d->in_y.off == 0 && d->out_y.off == 0 && d->out_x.off == 0;
}

int64_t batch_chunk(int64_t nb, size_t ws_field) {
    const size_t budget = size_t(tuning().batch_ws_mib) << 20;
    int64_t c = int64_t(budget / (ws_field ? ws_field : 1));
    if (c < 1) c = 1;
    return c < nb ? c : nb;
}

Fft2Plan plan_fft2(const pm_fft2_desc* d, bool allow_r2c) {
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

// Composite grids (round 4): the column length runs on its own factors with the column resident in LDS through forward stages, multiplier
// and transposed stages (mix_cols_mul); the row passes are the engine's (a power-of-two row length) or the mixed-radix row kernel, on
// NATURAL intermediates.  Three passes / 6 N^2 s bytes where two pm_fft2 calls move 8 N^2 s -- the reference takes any size through one
// code path (prysm/propagation/angular_spectrum.py:9-42, prysm/convolution.py:9-31).  One field, complex output.
bool plan_fused_mix(const pm_fft2_desc* d, FusedPlan& p) {
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

bool plan_fused(const pm_fft2_desc* d, FusedPlan& p) {
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

// ---------------------------------------------------------------- both axes not powers of two: 2-D Bluestein
// The 2-D cyclic convolution with the separable chirp IS the fused chain: window(ifft2(fft2(pad(a)) * (B1 (x) B2))) with the
// pad window [0, n) of MB on the way in and the same crop on the way out.
void blue2d_desc(pm_fft2_desc& dd, int dtype, int64_t M, int64_t N) {
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

size_t blue2d_fused_ws(int dtype, int64_t M, int64_t N) {
    pm_fft2_desc dd;
    blue2d_desc(dd, dtype, M, N);
    FusedPlan fp;
    return plan_fused(&dd, fp) ? fp.ws_bytes : 0;
}

int check_fft2(const pm_fft2_desc* d) {
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
size_t fft1_big_scratch(size_t es, int axis, int64_t batch, int64_t n) {
    const size_t arr = (size_t(batch) * size_t(n) * es + 255) & ~size_t(255);
    return axis == 1 ? 2 * arr : arr;
}


bool herm_conv_plan(const pm_fft2_desc* d, HermConvPlan& p) {
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
    // layout tiles of 8 column tiles for complex64 (profiles/r06/exp_conv_fold.log: 4096^2 fp32 126.3 -> 122.3 us, 2048^2 55.5 -> 54.8; fp64 within noise: 4)
    p.log_k = tuning().log_k >= 0 ? tuning().log_k : (d->dtype == PM_C64 ? 3 : 2);
    while (p.log_k > 0 && ((N / 2) % (int64_t(p.tc) << p.log_k)) != 0) --p.log_k;
    if ((N / 2) % p.tc) return false;
    const size_t es = d->dtype == PM_C64 ? 8 : 16;
    p.ws_bytes = size_t(M) * size_t(N / 2) * es;
    return true;
}

// ---- the wavelength loop as launch pairs over groups of wavelengths (fft_spectral.h)
// fast form: complex64 packed synthesis, |.|^2 accumulation, both lengths on the engine with a tiled intermediate, fewer than 4096^2
// bins.  Measured (profiles/r02/exp_spectral.log, us per wavelength, loop -> groups of 8): 1024^2 24.7 -> 11.0, 2048^2 40.6 -> 22.6,
// 1024^2 padded to 2048^2 34.0 -> 13.3; at 4096^2 the loop's passes already run at 84 % of copy speed with their intermediate in the
// Infinity Cache, and the grouped kernels pay for their registers with occupancy (rocprofv3: 47.6 + 50.7 us per wavelength against
// 45.0 + 55.9 in groups of 8), so those sizes keep the loop.
bool spectral_fast(const pm_fft2_desc* d, const Fft2Plan& p) {
    const int want = PM_FLAG_SYNTH_INPUT | PM_FLAG_SYNTH_PACKED;
    // complex128: rows of up to 2048 samples, unfolded (the grouped double-precision kernels exist for those; profiles/r02/exp_spectral_c128.log)
    if (d->dtype == PM_C128 && (p.logn > 11 || p.fold)) return false;
    return tuning().spectral > 1 && (d->flags & want) == want && !(d->flags & (PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY)) &&
           d->epilogue == PM_EPI_ABS2_ACCUM && d->batch <= 1 && d->mul_kind == PM_MUL_NONE && !p.r2c && !p.big_rn && !p.blue2d &&
           p.logn >= 5 && p.logn <= 12 /* its row kernel spills hundreds of registers at 8192-point rows (two rows per thread + the packed map) */ &&
           p.logm >= (p.fold ? 6 : 5) && p.logm - (p.fold ? 1 : 0) <= 11 /* the accumulating column kernel spills beyond 2048-point tiles */ &&
           p.tc != 0 && p.col_var == 0 && d->in_y.len > 0 && p.logn + p.logm < tuning().spectral_area_log;
}

int spectral_group(int32_t count) {
    int g = tuning().spectral;
    if (g > kSpectralMax) g = kSpectralMax;
    return g < count ? g : count;
}

// One line that says which route a descriptor takes -- the planner's decisions (plan_fft2 / plan_fused / herm_conv_plan under the
// calling thread's tuning knobs) in words.  Host logic only: callable without a GPU, so the routes of a table of shapes are pinned by
// a CPU test (tests/test_host_logic.py) and a shape that falls to a slow route shows up there and not as a timing.
// Does the row / column pass of this plan run on the composite register engine (fft_ce.h)?  The descriptor-level statement of what
// ce_rows_view / ce_cols_view (fft_ce_kernels.h) accept, in ONE place (it was written out in pm_plan_explain, in fft2_run's
// stack test and in the views): pm_plan_explain reports it, fft2_run sends (B, m, n) stacks out as grid.y by it.
bool ce_rows_axis(const pm_fft2_desc* d, const Fft2Plan& p) {
    const bool f32 = d->dtype == PM_C64;
    const size_t es = f32 ? 8 : 16;
    const int64_t N = d->in_x.n;
    return p.mix_n && !p.mix_fold && tuning().mix_engine && !(d->flags & PM_FLAG_REAL_INPUT) && (f32 || !(d->flags & PM_FLAG_SYNTH_INPUT)) &&
           (f32 ? ce_has_plan<float>(int(N)) : ce_has_plan<double>(int(N))) && ce_fits32(kCeMaxSeqs * d->in_ld + 2 * N, es) &&
           ce_fits32(kCeMaxSeqs * p.w_ld + 2 * N, es);
}
bool ce_cols_axis(const pm_fft2_desc* d, const Fft2Plan& p) {
    const bool f32 = d->dtype == PM_C64;
    const size_t es = f32 ? 8 : 16;
    const int64_t M = d->in_y.n, N = d->in_x.n;
    const bool whole_out = d->out_y.off == 0 && d->out_y.len == M && d->out_x.off == 0 && d->out_x.len == N;
    return p.mix_m && !p.mix_fold && tuning().mix_engine && whole_out && d->mul_kind == PM_MUL_NONE && d->epilogue <= PM_EPI_ABS2_ACCUM &&
           (f32 ? ce_has_plan<float>(int(M)) : ce_has_plan<double>(int(M))) && ce_fits32(2 * M * p.w_ld + kCeMaxSeqs, es) &&
           ce_fits32(2 * M * d->out_ld + N, es);
}
// ... both, for a stack of plain complex fields (no synthesis in the loads, whole transform in one call)
bool ce_both_axes_stack(const pm_fft2_desc* d, const Fft2Plan& p) {
    return !p.big_rn && !p.blue2d && !(d->flags & (PM_FLAG_SYNTH_INPUT | PM_FLAG_PASS1_ONLY | PM_FLAG_PASS2_ONLY)) && ce_rows_axis(d, p) && ce_cols_axis(d, p);
}

static const char* axis_route(bool engine, bool mix, bool blue) { return engine ? "stockham" : (mix ? "mixed-radix" : (blue ? "bluestein" : "direct")); }

int check_fft1(int32_t dtype, int32_t direction, int32_t axis, int64_t batch, const pm_axis* t_in, const pm_axis* t_out) {
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

}  // namespace pm

using namespace pm;

extern "C" {

size_t pm_fft2_workspace(const pm_fft2_desc* d) {
    if (check_fft2(d)) return 0;
    const Fft2Plan p = plan_fft2(d);
    if (!p.r2c) return p.ws_bytes;
    // the Hermitian path reads the real array as complex pairs: a base address that is not aligned like a complex element sends the
    // call down the complex path instead (pm_fft2 below), whose intermediate is larger -- the query covers both
    const size_t other = plan_fft2(d, false).ws_bytes;
    return other > p.ws_bytes ? other : p.ws_bytes;
}

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
        const bool ce_n = ce_rows_axis(d, p), ce_m = ce_cols_axis(d, p);
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

size_t pm_fft1_workspace(int32_t dtype, int32_t axis, int64_t batch, int64_t n) {
    if ((dtype != PM_C64 && dtype != PM_C128) || batch <= 0) return 0;
    const size_t es = dtype == PM_C64 ? 8 : 16;
    if (big_split(n) > 1) return fft1_big_scratch(es, axis, batch, n);     // 16384 / 32768 and 3 / 5 / 7 x 2^k: a radix-R step
    if (!use_blue(n)) return 0;
    return axis == 1 ? blue_rows_scratch(es, batch, n) : blue_cols_scratch(es, batch, n);
}

}  // extern "C"
