// The routes: parameter blocks and kernel launches for a planned 2-D transform, fused chain, Hermitian chain, 1-D transform, chirp-Z axis, wavelength group.
#include "capi_internal.h"

namespace pm {

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
int fft2_run(const pm_fft2_desc* d, const Fft2Plan& p, const void* in, void* out, void* ws, hipStream_t st) {
    if (p.nbatch <= 1) return fft2_run_chunk<T>(d, p, in, out, ws, st, 1);
    const bool engine = p.tc != 0;
    // ... and composite grids whose two passes both run on the composite register engine (fft_ce.h: grid.y = fields; round 5)
    const bool ce_stack = ce_both_axes_stack(d, p);
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
int fused_run(const pm_fft2_desc* d, const FusedPlan& p, const void* in, void* out, void* ws, hipStream_t st) {
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
int fft1_run(int direction, int axis, int64_t batch, const pm_axis* ti, const pm_axis* to, double scale,
                    const void* in, int64_t in_ld, void* out, int64_t out_ld, hipStream_t st, void* blue_ws) {
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
int czt_axis_run(int32_t axis, int64_t nseq, int64_t K, int64_t in_len, int64_t in_off, int64_t out_len, int64_t out_off,
                        const void* pre, int pre_conj, const void* H, int h_conj, const void* post, int post_conj, double scale,
                        const void* in, int64_t in_ld, void* out, int64_t out_ld, hipStream_t st, int single) {
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

template <typename T>
int herm_conv_run(const pm_fft2_desc* d, const HermConvPlan& p, const void* in, void* out, void* ws, hipStream_t st) {
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

template <typename T>
int fft2_spectral_group(const pm_fft2_desc* d, const Fft2Plan& p, const Spectral& w, const void* in, void* out, void* ws, hipStream_t st) {
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

// the runners the entry points call (capi.hip), both precisions
#define PM_INST(T)                                                                                                                              \
    template int fft2_run<T>(const pm_fft2_desc*, const Fft2Plan&, const void*, void*, void*, hipStream_t);                                     \
    template int fused_run<T>(const pm_fft2_desc*, const FusedPlan&, const void*, void*, void*, hipStream_t);                                   \
    template int herm_conv_run<T>(const pm_fft2_desc*, const HermConvPlan&, const void*, void*, void*, hipStream_t);                            \
    template int fft2_spectral_group<T>(const pm_fft2_desc*, const Fft2Plan&, const Spectral&, const void*, void*, void*, hipStream_t);                 \
    template int fft1_run<T>(int, int, int64_t, const pm_axis*, const pm_axis*, double, const void*, int64_t, void*, int64_t, hipStream_t, void*);    \
    template int czt_axis_run<T>(int32_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, const void*, int, const void*, int, const void*, int,  \
                                 double, const void*, int64_t, void*, int64_t, hipStream_t, int);
PM_INST(float)
PM_INST(double)
#undef PM_INST

}  // namespace pm
