// Round 5, VERDICT r4 item 1: what could a workgroup whose waves split into movers and transformers gain on the kernels that hold ONE
// workgroup per CU?  This standalone benchmark (engine headers only, no library, no Python) answers with an ABLATION of the shipped column
// kernels: the same kernel with its global loads, its global stores or its transform switched off, per configuration --
//     T(all)                         what ships
//     T(no transform)                memory only: loads + stores of every tile, nothing in between
//     T(no loads, no stores)         transform only (LDS exchanges, butterflies, twiddle loads)
//     T(no loads) / T(no stores)     one memory phase missing
// Perfect overlap of the phases of different tiles ON ONE CU (what wave specialisation is for) cannot beat max(T(memory only),
// T(transform only)); the stagger already overlaps the phases of different CUs.  If T(all) is within a few per cent of that maximum the
// split has nothing left to win, whatever it costs -- and it costs LDS the kernels do not have (see LDS / register budget printed below).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I prysm_amd/csrc -I include experiments/wave_spec/ws_bench.hip -o experiments/wave_spec/ws_bench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#define PM_PACKED_F32
#include "fft_r2c.h"

namespace pm {
static int g_stagger_group = 0, g_stagger = -1;
int pm_fft_stagger(int) { return g_stagger; }
int pm_stagger_group() { return g_stagger_group; }
int pm_num_cus() { return 256; }

// ABL bit 0: no global loads (registers from the thread index), bit 1: no global stores (one guarded store keeps the results alive),
// bit 2: no transform
template <typename C, int ABL, typename L, typename S>
__global__ void __launch_bounds__(C::NT, (fft_kernel_min_waves<C, true, 0, S>())) abl_kernel(const L lp, const S sp, const cx<typename C::T>* __restrict__ tw,
                                                                                              const int log_g_packed, const float poison) {
    using T = typename C::T;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int log_g = engine_stagger(log_g_packed);
    int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    unit = unit * C::BO + pos.bo;
    cx<T> v[C::E][C::P];
    const L lpb = at_batch(lp, blockIdx.y);
    const S spb = at_batch(sp, blockIdx.y);
    if constexpr (ABL & 1) {
#pragma unroll
        for (int e = 0; e < C::E; ++e)
#pragma unroll
            for (int m = 0; m < C::P; ++m) v[e][m] = {T(threadIdx.x) * T(1e-3) + T(m), T(e + unit) * T(1e-4)};
    } else {
        load<C>(lpb, unit, pos, v);
    }
    if constexpr (!(ABL & 4)) {
        if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
        else fft_run<C>(v, pos, pm_smem, tw);
    }
    if constexpr (ABL & 2) {
        T acc = T(0);
#pragma unroll
        for (int e = 0; e < C::E; ++e)
#pragma unroll
            for (int m = 0; m < C::P; ++m) acc += v[e][m].x * T(0.5) + v[e][m].y;
        if (acc == T(poison)) store<C>(spb, unit, pos, v);      // never true: the results stay live, nothing is written
    } else {
        store<C>(spb, unit, pos, v);
    }
}
// the Hermitian column kernel (fft_r2c.h fft_col_herm_kernel, fast-store path only) with the same switches; bit 3: no DC reduction
template <typename C, int EPI, int ABL>
__global__ void __launch_bounds__(C::NT) abl_herm_kernel(const ColLoadTiled<typename C::T> lp0, const HermStore<typename C::T> sp0,
                                                         const cx<typename C::T>* __restrict__ tw, const int log_g_packed, const float poison) {
    using T = typename C::T;
    const int log_g = engine_stagger(log_g_packed);
    constexpr int TC = C::CI * C::E;
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    const int unit = group_remap(blockIdx.x, gridDim.x, log_g) * C::BO + pos.bo;
    cx<T> v[C::E][C::P];
    const auto lp = at_batch(lp0, blockIdx.y);
    HermStore<T> sp = sp0;
    sp.plane = blockIdx.y;
    sp.dst = reinterpret_cast<T*>(sp.dst) + int64_t(blockIdx.y) * sp.plane_dst;
    if constexpr (ABL & 1) {
#pragma unroll
        for (int e = 0; e < C::E; ++e)
#pragma unroll
            for (int m = 0; m < C::P; ++m) v[e][m] = {T(threadIdx.x) * T(1e-3) + T(m), T(e + unit) * T(1e-4)};
    } else {
        load<C>(lp, unit, pos, v);
    }
    T s = sp.scale;
    if constexpr (!(ABL & 8)) {
        double* red = reinterpret_cast<double*>(pm_smem);
        double acc = 0.0;
        for (int q = threadIdx.x; q < sp.nrows_w; q += C::NT) acc += double(sp.w0[int64_t(q) * sp.w0_stride].x);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_down(acc, off, 64);
        constexpr int NW = (C::NT + 63) / 64;
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        double dc = 1.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) dc += red[w];
        __syncthreads();
        s = T(double(sp.scale) / dc);
    }
    if constexpr (!(ABL & 4)) {
        if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
        else fft_run<C>(v, pos, pm_smem, tw);
    }
    const int col0 = unit * TC + pos.cl * C::E;
    if constexpr (ABL & 2) {
        T acc = s;
#pragma unroll
        for (int e = 0; e < C::E; ++e)
#pragma unroll
            for (int m = 0; m < C::P; ++m) acc += v[e][m].x * T(0.5) + v[e][m].y;
        if (acc == T(poison)) herm_store_fast<C, C::P / 2, EPI>(sp, col0 ? col0 : TC, pos, v, s);
    } else {
        herm_store_fast<C, C::P / 2, EPI>(sp, col0 ? col0 : TC, pos, v, s);      // (tile 0's first thread writes tile 1's place: timing only)
    }
}
}  // namespace pm

using namespace pm;

template <typename T> static cx<T>* make_twiddles(int n);

template <typename C, int ABL>
static float time_herm(const ColLoadTiled<typename C::T>& lp, const HermStore<typename C::T>& sp, const cx<typename C::T>* tw, int grid, int log_g, int reps) {
    auto kern = abl_herm_kernel<C, EPI_ABS, ABL>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(C::LDS_BYTES));
    const int lg = engine_log_g(log_g, grid * 2, C::LDS_BYTES, C::NT, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid, 2), dim3(C::NT), C::LDS_BYTES, 0, lp, sp, tw, lg, -1.25e30f);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid, 2), dim3(C::NT), C::LDS_BYTES, 0, lp, sp, tw, lg, -1.25e30f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms / reps < best ? ms / reps : best;
    }
    if (hipGetLastError() != hipSuccess) return -1.f;
    return best * 1e3f;
}

// column pass of the Hermitian path for a real M x N fp32 field (folded): N/2 columns, two planes of M/2-point tiles of 16 columns,
// |F| / F[0][0] written as fp32, every bin twice (direct and mirrored)
template <int LOGH, int VAR>
static void run_herm(const char* name, int N, int log_k, int rounds) {
    using T = float;
    using C = typename ColCfgSel<T, LOGH, VAR>::type;
    constexpr int TC = C::CI * C::E, H = 1 << LOGH;
    const int M = 2 * H, n2 = N / 2, ntiles = n2 / TC;
    const int64_t tl = int64_t(TC) << log_k, ntl = (n2 + tl - 1) / tl, plane = ntl * H * tl;
    cx<T>* W;
    T* out;
    (void)hipMalloc(&W, size_t(2) * plane * sizeof(cx<T>));
    (void)hipMalloc(&out, size_t(M) * N * sizeof(T));
    (void)hipMemset(W, 0, size_t(2) * plane * sizeof(cx<T>));
    const cx<T>* tw = make_twiddles<T>(H);
    ColLoadTiled<T> cl{W, H, AxisMap{H, H, 0, 0}, ntiles, log_k, plane};
    HermStore<T> hs{out, 2 * int64_t(N), AxisMap{H, H, 0, H / 2}, AxisMap{N, N, 0, N / 2}, H, N, EPI_ABS, T(1), T(1), 1, W, tl, H, 0, N, 1, VAR == 2 ? 1 : 0};
    const int log_g = log_k < 1 ? 1 : (log_k > 3 ? 3 : log_k);
    printf("== %s: real %d x %d -> %d columns, tiles of %d columns x %d rows, %d threads, LDS %zu KiB; %d workgroups; algorithmic %.1f MB per launch "
           "(%.1f read + %.1f written)\n", name, M, N, n2, TC, H, C::NT, C::LDS_BYTES / 1024, 2 * ntiles, (size_t(M) * n2 * 8 + size_t(M) * N * 4) / 1e6,
           size_t(M) * n2 * 8 / 1e6, size_t(M) * N * 4 / 1e6);
    const int reps = 30;
    for (int r = 0; r < rounds; ++r) {
        const float all = time_herm<C, 0>(cl, hs, tw, ntiles / C::BO, log_g, reps), nodc = time_herm<C, 8>(cl, hs, tw, ntiles / C::BO, log_g, reps),
                    mem = time_herm<C, 12>(cl, hs, tw, ntiles / C::BO, log_g, reps), xf = time_herm<C, 11>(cl, hs, tw, ntiles / C::BO, log_g, reps),
                    nold = time_herm<C, 9>(cl, hs, tw, ntiles / C::BO, log_g, reps), nost = time_herm<C, 10>(cl, hs, tw, ntiles / C::BO, log_g, reps);
        printf("   round %d: all %.1f us | without the DC reduction %.1f | then: memory only %.1f | transform only %.1f | no loads %.1f | no stores %.1f\n",
               r, all, nodc, mem, xf, nold, nost);
    }
    (void)hipFree(W);
    (void)hipFree(out);
    (void)hipFree(const_cast<cx<T>*>(tw));
}

template <typename T>
static cx<T>* make_twiddles(int n) {
    std::vector<cx<T>> h(n);
    const long double pi = acosl(-1.0L);
    for (int i = 0; i < n; ++i) h[i] = {T(cosl(-2.0L * pi * i / n)), T(sinl(-2.0L * pi * i / n))};
    cx<T>* d;
    (void)hipMalloc(&d, n * sizeof(cx<T>));
    (void)hipMemcpy(d, h.data(), n * sizeof(cx<T>), hipMemcpyHostToDevice);
    return d;
}

template <typename C, int ABL, typename L, typename S>
static float time_one(const L& lp, const S& sp, const cx<typename C::T>* tw, int grid, int planes, int log_g, int reps) {
    auto kern = abl_kernel<C, ABL, L, S>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(C::LDS_BYTES));
    const int lg = engine_log_g(log_g, grid * planes, C::LDS_BYTES, C::NT, 1);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid, planes), dim3(C::NT), C::LDS_BYTES, 0, lp, sp, tw, lg, -1.25e30f);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid, planes), dim3(C::NT), C::LDS_BYTES, 0, lp, sp, tw, lg, -1.25e30f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms / reps < best ? ms / reps : best;
    }
    if (hipGetLastError() != hipSuccess) return -1.f;
    return best * 1e3f;
}

// column pass of a FOLDED M x N transform: two planes of M/2-point tiles, tiled intermediate in, natural output out
template <typename T, int LOGH, int VAR>
static void run_config(const char* name, int N, int log_k, int rounds) {
    using C = typename ColCfgSel<T, LOGH, VAR>::type;
    constexpr int TC = C::CI * C::E, H = 1 << LOGH;
    const int M = 2 * H, ntiles = N / TC;
    const int64_t tl = int64_t(TC) << log_k, ntl = (N + tl - 1) / tl;
    const size_t bytes = size_t(M) * N * sizeof(cx<T>);
    cx<T>*W, *out;
    (void)hipMalloc(&W, bytes);
    (void)hipMalloc(&out, bytes);
    (void)hipMemset(W, 0, bytes);
    const cx<T>* tw = make_twiddles<T>(H);
    ColLoadTiled<T> cl{W, H, AxisMap{H, H, 0, 0}, ntiles, log_k, ntl * H * tl};
    ColStoreNat<T> cs{};
    cs.dst = out;
    cs.ld = 2 * N;
    cs.ay = AxisMap{H, H, 0, H / 2};
    cs.ax = AxisMap{N, N, 0, N / 2};
    cs.scale = T(1);
    cs.vec_ok = 3;
    cs.bstride = N;
    const size_t out_bytes = bytes;
    cs.nt = (out_bytes >= (size_t(192) << 20) && out_bytes < (size_t(384) << 20)) ? 1 : 0;     // capi.hip make_colstore
    const int log_g = log_k < 1 ? 1 : (log_k > 3 ? 3 : log_k);
    const int grid = ntiles / C::BO;
    const int per_cu = C::LDS_BYTES ? int(160 * 1024 / C::LDS_BYTES) : 8;
    printf("== %s: %d x %d, tiles of %d columns x %d rows (%zu KiB in registers), %d threads, LDS %zu KiB -> %d workgroup(s) per CU; %d tiles; "
           "algorithmic %.1f MB per launch\n", name, M, N, TC, H, size_t(TC) * H * sizeof(cx<T>) / 1024, C::NT, C::LDS_BYTES / 1024,
           per_cu < 2048 / C::NT ? per_cu : 2048 / C::NT, 2 * ntiles, 2.0 * bytes / 1e6);
    printf("   LDS budget of a mover / transformer split: exchange fabric %zu KiB + hand-over buffer of one tile %zu KiB = %zu KiB of 160\n",
           C::LDS_BYTES / 1024, size_t(TC) * H * sizeof(cx<T>) / 1024, (C::LDS_BYTES + size_t(TC) * H * sizeof(cx<T>)) / 1024);
    const int reps = 30;
    for (int sg = 0; sg < 2; ++sg) {
        g_stagger_group = sg;
        for (int r = 0; r < rounds; ++r) {
            const float all = time_one<C, 0>(cl, cs, tw, grid, 2, log_g, reps), mem = time_one<C, 4>(cl, cs, tw, grid, 2, log_g, reps),
                        xf = time_one<C, 3>(cl, cs, tw, grid, 2, log_g, reps), nold = time_one<C, 1>(cl, cs, tw, grid, 2, log_g, reps),
                        nost = time_one<C, 2>(cl, cs, tw, grid, 2, log_g, reps);
            const float bound = mem > xf ? mem : xf;
            printf("   stagger_group %d round %d: all %.1f us | memory only %.1f | transform only %.1f | no loads %.1f | no stores %.1f | "
                   "all / max(memory, transform) = %.3f | sum of the two = %.1f\n", sg, r, all, mem, xf, nold, nost, all / bound, mem + xf);
        }
    }
    (void)hipFree(W);
    (void)hipFree(out);
    (void)hipFree(const_cast<cx<T>*>(tw));
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 2;
    run_config<float, 11, 0>("complex64 4096^2 (two workgroups per CU: the reference)", 4096, 7, rounds);
    run_config<double, 11, 2>("complex128 4096^2 (128 B tiles)", 4096, 2, rounds);
    run_config<double, 11, 0>("complex128 4096^2 (64 B tiles, two workgroups per CU)", 4096, 2, rounds);
    run_config<float, 12, 0>("complex64 8192^2", 8192, 3, rounds);
    run_herm<11, 2>("Hermitian column pass, MTF of a 4096^2 fp32 PSF (16-column tiles)", 4096, 2, rounds);
    run_herm<11, 0>("Hermitian column pass, 4096^2, 8-column tiles (two workgroups per CU)", 4096, 2, rounds);
    run_herm<12, 0>("Hermitian column pass, MTF of an 8192^2 fp32 PSF", 8192, 2, rounds);
    return 0;
}
