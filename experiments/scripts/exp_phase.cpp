// Where does a single-round FFT pass spend its time?  The shipped row / column kernels of a 2048^2 complex64 transform, rebuilt here
// with s_memtime stamps at the phase boundaries of every wave (entry, loads issued, data arrived, transform done, stores issued,
// stores retired).  tools only: the shipped kernels carry no instrumentation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I prysm_amd/csrc -I include tools/exp_phase.cpp -o /tmp/exp_phase
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <cmath>
#include "fft_kernels.h"
using namespace pm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", int(e_), __LINE__); exit(1); } } while (0)

template <typename C, bool COL, typename L, typename S>
__global__ void __launch_bounds__(C::NT) probe(const L lp, const S sp, const cx<typename C::T>* __restrict__ tw, int log_g, unsigned long long* stamps) {
    extern __shared__ __attribute__((aligned(16))) char pm_smem[];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const ThreadPos pos = thread_pos<C>(threadIdx.x);
    int unit = group_remap(blockIdx.x, gridDim.x, log_g);
    if (COL) unit = unit * C::BO + pos.bo;
    cx<typename C::T> v[C::E][C::P];
    load<C>(lp, unit, pos, v);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if constexpr (C::E == 2 && C::COMP == 1 && C::NSTAGE > 1) fft_run_pipe2<C>(v, pos, pm_smem, tw);
    else fft_run<C>(v, pos, pm_smem, tw);
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    store<C>(sp, unit, pos, v);
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = stamps + (size_t(blockIdx.x) * (C::NT / 64) + threadIdx.x / 64) * 6;
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = t5;
    }
}

// s_memtime ticks at the shader clock and every XCD has its own counter: durations are per wave, entry times are taken relative to
// the first wave of the same XCD (workgroup b runs on XCD b % 8)
static void report(const char* name, std::vector<unsigned long long>& h, size_t nw, int waves_per_wg, double mhz) {
    const char* ph[5] = {"issue loads", "wait for data", "transform", "issue stores", "stores retire"};
    unsigned long long first[8], last[8];
    for (int x = 0; x < 8; ++x) { first[x] = ~0ull; last[x] = 0; }
    for (size_t w = 0; w < nw; ++w) {
        const int x = int((w / waves_per_wg) % 8);
        first[x] = std::min(first[x], h[w * 6]);
        last[x] = std::max(last[x], h[w * 6 + 5]);
    }
    double span = 0;
    for (int x = 0; x < 8; ++x) span = std::max(span, double(last[x] - first[x]) / mhz);
    printf("%s: %zu waves; per XCD, first entry -> last retire: %.2f us (shader clock taken as %.0f MHz)\n", name, nw, span, mhz);
    std::vector<double> v(nw);
    for (size_t w = 0; w < nw; ++w) v[w] = (h[w * 6] - first[(w / waves_per_wg) % 8]) / mhz;
    std::sort(v.begin(), v.end());
    printf("   entry after the XCD's first wave: 10%% %.2f us, median %.2f, 90%% %.2f, max %.2f\n", v[nw / 10], v[nw / 2], v[nw * 9 / 10], v[nw - 1]);
    for (int p = 0; p < 5; ++p) {
        for (size_t w = 0; w < nw; ++w) v[w] = (h[w * 6 + p + 1] - h[w * 6 + p]) / mhz;
        std::sort(v.begin(), v.end());
        printf("   %-14s median %.2f us, 10%% %.2f, 90%% %.2f\n", ph[p], v[nw / 2], v[nw / 10], v[nw * 9 / 10]);
    }
    for (size_t w = 0; w < nw; ++w) v[w] = (h[w * 6 + 5] - h[w * 6]) / mhz;
    std::sort(v.begin(), v.end());
    printf("   wave lifetime  median %.2f us, 10%% %.2f, 90%% %.2f\n", v[nw / 2], v[nw / 10], v[nw * 9 / 10]);
}

int main() {
    using T = float;
    const int n = 2048;
    const size_t elems = size_t(n) * n;
    cx<T>*in, *ws, *out, *tw;
    CK(hipMalloc(&in, elems * 8)); CK(hipMalloc(&ws, elems * 8)); CK(hipMalloc(&out, elems * 8)); CK(hipMalloc(&tw, n * 8));
    std::vector<cx<T>> h(elems), htw(n);
    for (size_t i = 0; i < elems; ++i) h[i] = {float((i * 2654435761u) % 1000) * 1e-3f - 0.5f, float((i * 40503u) % 1000) * 1e-3f - 0.5f};
    for (int i = 0; i < n; ++i) htw[i] = {float(cos(-2 * M_PI * i / n)), float(sin(-2 * M_PI * i / n))};
    CK(hipMemcpy(in, h.data(), elems * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(tw, htw.data(), n * 8, hipMemcpyHostToDevice));
    // s_memtime runs at a constant rate: calibrate it against a HIP-event timed spin
    double mhz = 2400.0;
    {
        int v = 0;
        (void)hipDeviceGetAttribute(&v, hipDeviceAttributeClockRate, 0);     // kHz
        if (v > 0) mhz = v / 1e3;
    }
    using CR = RowCfgSel<T, 11, 5>::type;     // what row_variant() picks for 2048-point complex64 rows
    using CC = ColCfgSel<T, 11, 0>::type;
    const int log_k = 1, tc = 8, ltc = 4;     // plan_fft2: N = 2048 -> log_k = 1, tile width 16
    RowLoadNat<T> rl{in, n, AxisMap{n, n, 0, n / 2}, n, 0, 0, 0};
    RowStoreTiled<T> rs{ws, n, ltc, 0};
    ColLoadTiled<T> cl{ws, n, AxisMap{n, n, 0, n / 2}, n / tc, log_k, 0};
    ColStoreNat<T> cs{};
    cs.dst = out; cs.ld = n; cs.ay = AxisMap{n, n, 0, n / 2}; cs.ax = AxisMap{n, n, 0, n / 2}; cs.scale = 1.f / n; cs.weight = 1; cs.vec_ok = 1;
    const int rgrid = n / (CR::BO * CR::E), cgrid = (n / tc) / CC::BO;
    const size_t rw = size_t(rgrid) * (CR::NT / 64), cw = size_t(cgrid) * (CC::NT / 64);
    unsigned long long* st;
    CK(hipMalloc(&st, std::max(rw, cw) * 6 * 8));
    auto kr = probe<CR, false, RowLoadNat<T>, RowStoreTiled<T>>;
    auto kc = probe<CC, true, ColLoadTiled<T>, ColStoreNat<T>>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kc), hipFuncAttributeMaxDynamicSharedMemorySize, int(CC::LDS_BYTES)));
    std::vector<unsigned long long> hs(std::max(rw, cw) * 6);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kr, dim3(rgrid), dim3(CR::NT), CR::LDS_BYTES, 0, rl, rs, tw, 1, st);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hs.data(), st, rw * 6 * 8, hipMemcpyDeviceToHost));
        if (rep == 2) report("row pass 2048^2 c64 (2048 workgroups x 128 threads)", hs, rw, CR::NT / 64, mhz);
        hipLaunchKernelGGL(kc, dim3(cgrid), dim3(CC::NT), CC::LDS_BYTES, 0, cl, cs, tw, 1, st);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hs.data(), st, cw * 6 * 8, hipMemcpyDeviceToHost));
        if (rep == 2) report("column pass 2048^2 c64 (256 workgroups x 512 threads)", hs, cw, CC::NT / 64, mhz);
    }
    return 0;
}
