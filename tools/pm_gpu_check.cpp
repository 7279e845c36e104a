// Standalone GPU check + microbenchmark for libprysm_amd.so (development tool; the judged
// parity tests are tests/test_gpu_*.py, which go through the same C ABI from Python).
//
//   build: make -C tools          run: tools/pm_gpu_check [quick|full|bench]
//
// Verifies pm_fft2 / pm_fft1 / pm_cgemm / pointwise kernels against fp64 host references (naive DFT
// for small and awkward sizes, an iterative radix-2 fp64 FFT for the large ones) and times the two FFT
// passes with hipEvents, printing achieved algorithmic GB/s next to a device-to-device copy.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <random>
#include <string>
#include <vector>

#include "../include/prysm_amd.h"

typedef std::complex<double> cd;
static int g_fail = 0;
#define HIPCHECK(x)                                                                   \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                  \
        }                                                                             \
    } while (0)

static void report(const std::string& what, double err, double tol) {
    const bool ok = err < tol;
    printf("%-64s err=%.3e %s\n", what.c_str(), err, ok ? "ok" : "FAIL");
    if (!ok) ++g_fail;
}

// ---- host references ------------------------------------------------------------------------
static void fft_pow2(std::vector<cd>& a, int sign) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = sign * 2 * M_PI / double(len);
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const cd w(cos(ang * double(k)), sin(ang * double(k)));
                cd u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}
static void dft_any(std::vector<cd>& a, int sign) {
    const size_t n = a.size();
    if (n >= 2 && (n & (n - 1)) == 0) return fft_pow2(a, sign);
    std::vector<cd> o(n);
    for (size_t k = 0; k < n; ++k) {
        cd acc = 0;
        for (size_t m = 0; m < n; ++m) {
            const double ang = sign * 2 * M_PI * double((m * k) % n) / double(n);
            acc += a[m] * cd(cos(ang), sin(ang));
        }
        o[k] = acc;
    }
    a = o;
}
// logical (padded + rotated) 2-D transform with output rotation / crop, as pm_fft2 defines it
static std::vector<cd> ref_fft2(const pm_fft2_desc& d, const std::vector<cd>& x) {
    const int64_t M = d.in_y.n, N = d.in_x.n;
    std::vector<cd> P(size_t(M) * N, cd(0, 0));
    for (int64_t r = 0; r < M; ++r)
        for (int64_t c = 0; c < N; ++c) {
            const int64_t qr = (r + d.in_y.shift) % M - d.in_y.off, qc = (c + d.in_x.shift) % N - d.in_x.off;
            if (qr >= 0 && qr < d.in_y.len && qc >= 0 && qc < d.in_x.len) P[size_t(r * N + c)] = x[size_t(qr * d.in_ld + qc)];
        }
    std::vector<cd> row(N), col(M);
    for (int64_t r = 0; r < M; ++r) {
        for (int64_t c = 0; c < N; ++c) row[c] = P[size_t(r * N + c)];
        dft_any(row, d.direction);
        for (int64_t c = 0; c < N; ++c) P[size_t(r * N + c)] = row[c];
    }
    for (int64_t c = 0; c < N; ++c) {
        for (int64_t r = 0; r < M; ++r) col[r] = P[size_t(r * N + c)];
        dft_any(col, d.direction);
        for (int64_t r = 0; r < M; ++r) P[size_t(r * N + c)] = col[r] * d.scale;
    }
    return P;  // indexed by unshifted bin (k, c)
}

template <typename T>
static std::vector<std::complex<T>> to_dev_type(const std::vector<cd>& x) {
    std::vector<std::complex<T>> o(x.size());
    for (size_t i = 0; i < x.size(); ++i) o[i] = std::complex<T>(T(x[i].real()), T(x[i].imag()));
    return o;
}

struct Case2 {
    int64_t M, N, in_r, in_c, out_r, out_c;
    bool shift;
    int dir, epi;
};

template <typename T>
static void check_fft2(const Case2& cs, double tol) {
    pm_fft2_desc d;
    memset(&d, 0, sizeof d);
    d.dtype = sizeof(T) == 4 ? PM_C64 : PM_C128;
    d.direction = cs.dir;
    d.epilogue = cs.epi;
    d.scale = 1.0 / sqrt(double(cs.M) * cs.N);
    d.weight = 1.0;
    d.in_y = {cs.M, cs.in_r, (cs.M - cs.in_r + 1) / 2, cs.shift ? cs.M / 2 : 0};
    d.in_x = {cs.N, cs.in_c, (cs.N - cs.in_c + 1) / 2, cs.shift ? cs.N / 2 : 0};
    d.out_y = {cs.M, cs.out_r, (cs.M - cs.out_r + 1) / 2, cs.shift ? cs.M / 2 : 0};
    d.out_x = {cs.N, cs.out_c, (cs.N - cs.out_c + 1) / 2, cs.shift ? cs.N / 2 : 0};
    d.in_ld = cs.in_c;
    d.out_ld = cs.out_c;
    std::mt19937_64 rng(cs.M * 7919 + cs.N);
    std::normal_distribution<double> nd;
    std::vector<cd> x(size_t(cs.in_r) * cs.in_c);
    for (auto& e : x) e = cd(nd(rng), nd(rng));
    auto hx = to_dev_type<T>(x);
    for (size_t i = 0; i < x.size(); ++i) x[i] = cd(hx[i].real(), hx[i].imag());
    const size_t es = 2 * sizeof(T);
    void *din, *dout, *ws;
    const size_t wsb = pm_fft2_workspace(&d);
    const size_t out_elems = size_t(cs.out_r) * cs.out_c;
    HIPCHECK(hipMalloc(&din, x.size() * es + 16));
    HIPCHECK(hipMalloc(&dout, out_elems * es + 16));
    HIPCHECK(hipMalloc(&ws, wsb));
    HIPCHECK(hipMemcpy(din, hx.data(), x.size() * es, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(dout, 0xff, out_elems * es));
    int rc = pm_fft2(&d, din, dout, ws, wsb, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    char name[160];
    snprintf(name, sizeof name, "fft2 %s %lldx%lld in %lldx%lld out %lldx%lld sh%d dir%+d epi%d", sizeof(T) == 4 ? "c64" : "c128",
             (long long)cs.M, (long long)cs.N, (long long)cs.in_r, (long long)cs.in_c, (long long)cs.out_r, (long long)cs.out_c,
             (int)cs.shift, cs.dir, cs.epi);
    if (rc) {
        printf("%s rc=%d (%s)\n", name, rc, pm_last_error());
        ++g_fail;
    } else {
        auto ref = ref_fft2(d, x);
        double err = 0, nrm = 0;
        if (cs.epi == PM_EPI_NONE) {
            std::vector<std::complex<T>> o(out_elems);
            HIPCHECK(hipMemcpy(o.data(), dout, out_elems * es, hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < cs.M; ++k)
                for (int64_t c = 0; c < cs.N; ++c) {
                    const int64_t qy = (k + d.out_y.shift) % cs.M - d.out_y.off, qx = (c + d.out_x.shift) % cs.N - d.out_x.off;
                    if (qy < 0 || qy >= cs.out_r || qx < 0 || qx >= cs.out_c) continue;
                    const cd r = ref[size_t(k * cs.N + c)];
                    const auto g = o[size_t(qy * cs.out_c + qx)];
                    err = fmax(err, std::abs(r - cd(g.real(), g.imag())));
                    nrm = fmax(nrm, std::abs(r));
                }
        } else {
            std::vector<T> o(out_elems);
            HIPCHECK(hipMemcpy(o.data(), dout, out_elems * sizeof(T), hipMemcpyDeviceToHost));
            for (int64_t k = 0; k < cs.M; ++k)
                for (int64_t c = 0; c < cs.N; ++c) {
                    const int64_t qy = (k + d.out_y.shift) % cs.M - d.out_y.off, qx = (c + d.out_x.shift) % cs.N - d.out_x.off;
                    if (qy < 0 || qy >= cs.out_r || qx < 0 || qx >= cs.out_c) continue;
                    const double r = std::norm(ref[size_t(k * cs.N + c)]);
                    err = fmax(err, fabs(r - double(o[size_t(qy * cs.out_c + qx)])));
                    nrm = fmax(nrm, r);
                }
        }
        report(name, err / nrm, tol);
    }
    HIPCHECK(hipFree(din));
    HIPCHECK(hipFree(dout));
    HIPCHECK(hipFree(ws));
}

template <typename T>
static void check_fft1(int64_t n, int64_t len, int64_t batch, int axis, int dir, double tol) {
    // in: (batch x len) for axis 1, (len x batch) for axis 0; zero padded to n; out full length n
    const int64_t rows = axis == 1 ? batch : len, cols = axis == 1 ? len : batch;
    const int64_t orows = axis == 1 ? batch : n, ocols = axis == 1 ? n : batch;
    std::mt19937_64 rng(n * 31 + len);
    std::normal_distribution<double> nd;
    std::vector<cd> x(size_t(rows) * cols);
    for (auto& e : x) e = cd(nd(rng), nd(rng));
    auto hx = to_dev_type<T>(x);
    for (size_t i = 0; i < x.size(); ++i) x[i] = cd(hx[i].real(), hx[i].imag());
    const size_t es = 2 * sizeof(T);
    void *din, *dout;
    HIPCHECK(hipMalloc(&din, x.size() * es));
    HIPCHECK(hipMalloc(&dout, size_t(orows) * ocols * es));
    HIPCHECK(hipMemcpy(din, hx.data(), x.size() * es, hipMemcpyHostToDevice));
    pm_axis ti = {n, len, 0, 0}, to = {n, n, 0, 0};
    int rc = pm_fft1(sizeof(T) == 4 ? PM_C64 : PM_C128, dir, axis, batch, &ti, &to, 1.0, din, cols, dout, ocols, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    char name[128];
    snprintf(name, sizeof name, "fft1 %s n=%lld len=%lld batch=%lld axis=%d dir%+d", sizeof(T) == 4 ? "c64" : "c128", (long long)n,
             (long long)len, (long long)batch, axis, dir);
    if (rc) {
        printf("%s rc=%d (%s)\n", name, rc, pm_last_error());
        ++g_fail;
    } else {
        std::vector<std::complex<T>> o(size_t(orows) * ocols);
        HIPCHECK(hipMemcpy(o.data(), dout, o.size() * es, hipMemcpyDeviceToHost));
        double err = 0, nrm = 0;
        std::vector<cd> s(n);
        for (int64_t b = 0; b < batch; ++b) {
            for (int64_t i = 0; i < n; ++i) s[i] = i < len ? (axis == 1 ? x[size_t(b * cols + i)] : x[size_t(i * cols + b)]) : cd(0, 0);
            dft_any(s, dir);
            for (int64_t k = 0; k < n; ++k) {
                const auto g = axis == 1 ? o[size_t(b * ocols + k)] : o[size_t(k * ocols + b)];
                err = fmax(err, std::abs(s[k] - cd(g.real(), g.imag())));
                nrm = fmax(nrm, std::abs(s[k]));
            }
        }
        report(name, err / nrm, tol);
    }
    HIPCHECK(hipFree(din));
    HIPCHECK(hipFree(dout));
}

template <typename T>
static void check_cgemm(int opA, int opB, int64_t M, int64_t N, int64_t K, double tol, bool use_ws) {
    std::mt19937_64 rng(M * 3 + N * 5 + K * 7 + opA * 11 + opB);
    std::normal_distribution<double> nd;
    const int64_t ar = (opA & 2) ? K : M, ac = (opA & 2) ? M : K, br = (opB & 2) ? N : K, bc = (opB & 2) ? K : N;
    std::vector<cd> A(size_t(ar) * ac), B(size_t(br) * bc);
    for (auto& e : A) e = cd(nd(rng), nd(rng));
    for (auto& e : B) e = cd(nd(rng), nd(rng));
    auto hA = to_dev_type<T>(A), hB = to_dev_type<T>(B);
    for (size_t i = 0; i < A.size(); ++i) A[i] = cd(hA[i].real(), hA[i].imag());
    for (size_t i = 0; i < B.size(); ++i) B[i] = cd(hB[i].real(), hB[i].imag());
    const size_t es = 2 * sizeof(T);
    void *dA, *dB, *dC, *ws = nullptr;
    const int dt = sizeof(T) == 4 ? PM_C64 : PM_C128;
    size_t wsb = use_ws ? pm_cgemm_workspace(dt, M, N, K) : 0;
    HIPCHECK(hipMalloc(&dA, A.size() * es));
    HIPCHECK(hipMalloc(&dB, B.size() * es));
    HIPCHECK(hipMalloc(&dC, size_t(M) * N * es));
    if (wsb) HIPCHECK(hipMalloc(&ws, wsb));
    HIPCHECK(hipMemcpy(dA, hA.data(), A.size() * es, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dB, hB.data(), B.size() * es, hipMemcpyHostToDevice));
    const double alpha = 0.37;
    int rc = pm_cgemm(dt, opA, opB, M, N, K, alpha, dA, ac, dB, bc, dC, N, ws, wsb, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    char name[128];
    snprintf(name, sizeof name, "cgemm %s opA=%d opB=%d %lldx%lldx%lld ws=%zu", sizeof(T) == 4 ? "c64" : "c128", opA, opB, (long long)M,
             (long long)N, (long long)K, wsb);
    if (rc) {
        printf("%s rc=%d (%s)\n", name, rc, pm_last_error());
        ++g_fail;
    } else {
        std::vector<std::complex<T>> o(size_t(M) * N);
        HIPCHECK(hipMemcpy(o.data(), dC, o.size() * es, hipMemcpyDeviceToHost));
        double err = 0, nrm = 0;
        // check a subset of rows when big
        const int64_t rstep = M > 64 ? M / 37 + 1 : 1;
        for (int64_t i = 0; i < M; i += rstep)
            for (int64_t j = 0; j < N; ++j) {
                cd acc = 0;
                for (int64_t k = 0; k < K; ++k) {
                    cd a = (opA & 2) ? A[size_t(k * ac + i)] : A[size_t(i * ac + k)];
                    cd b = (opB & 2) ? B[size_t(j * bc + k)] : B[size_t(k * bc + j)];
                    if (opA & 1) a = std::conj(a);
                    if (opB & 1) b = std::conj(b);
                    acc += a * b;
                }
                acc *= alpha;
                const auto g = o[size_t(i * N + j)];
                err = fmax(err, std::abs(acc - cd(g.real(), g.imag())));
                nrm = fmax(nrm, std::abs(acc));
            }
        report(name, err / nrm, tol);
    }
    HIPCHECK(hipFree(dA));
    HIPCHECK(hipFree(dB));
    HIPCHECK(hipFree(dC));
    if (ws) HIPCHECK(hipFree(ws));
}

// ---- calibration kernels: what a perfect streaming pass achieves on this box --------------------
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NTL, int NTS>
__global__ void copy_kernel(const f4* __restrict__ a, f4* __restrict__ b, size_t n) {
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        f4 v = NTL ? __builtin_nontemporal_load(a + i) : a[i];
        if (NTS)
            __builtin_nontemporal_store(v, b + i);
        else
            b[i] = v;
    }
}
static void launch_copy(int ntl, int nts, const void* a, void* b, size_t bytes) {
    const size_t n = bytes / 16;
    dim3 g(2048), blk(256);
    if (!ntl && !nts) hipLaunchKernelGGL((copy_kernel<0, 0>), g, blk, 0, nullptr, (const f4*)a, (f4*)b, n);
    if (ntl && !nts) hipLaunchKernelGGL((copy_kernel<1, 0>), g, blk, 0, nullptr, (const f4*)a, (f4*)b, n);
    if (!ntl && nts) hipLaunchKernelGGL((copy_kernel<0, 1>), g, blk, 0, nullptr, (const f4*)a, (f4*)b, n);
    if (ntl && nts) hipLaunchKernelGGL((copy_kernel<1, 1>), g, blk, 0, nullptr, (const f4*)a, (f4*)b, n);
}
static void calibrate() {
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    for (size_t mb : {8, 32, 128, 512, 2048}) {
        void *a, *b;
        HIPCHECK(hipMalloc(&a, mb << 20));
        HIPCHECK(hipMalloc(&b, mb << 20));
        HIPCHECK(hipMemset(a, 1, mb << 20));
        for (int v = 0; v < 4; ++v) {
            for (int i = 0; i < 3; ++i) launch_copy(v & 1, v >> 1, a, b, mb << 20);
            HIPCHECK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < 20; ++i) launch_copy(v & 1, v >> 1, a, b, mb << 20);
            HIPCHECK(hipEventRecord(e1, nullptr));
            HIPCHECK(hipEventSynchronize(e1));
            float ms;
            HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("CALIB copy kernel %zu MiB ntload=%d ntstore=%d: %.1f us -> %.0f GB/s (read+write)\n", mb, v & 1, v >> 1,
                   ms / 20 * 1e3, 2.0 * double(mb << 20) / (ms / 20) / 1e6);
        }
        HIPCHECK(hipFree(a));
        HIPCHECK(hipFree(b));
    }
    // the three-buffer cycle of one propagation (in -> ws -> out, 128 MiB each) done by plain copies: the
    // time two PERFECT streaming passes would take in the same cache regime as the real transform
    for (size_t mb : {8, 32, 128, 256}) {
        void *in, *ws, *out;
        HIPCHECK(hipMalloc(&in, mb << 20));
        HIPCHECK(hipMalloc(&ws, mb << 20));
        HIPCHECK(hipMalloc(&out, mb << 20));
        HIPCHECK(hipMemset(in, 1, mb << 20));
        for (int v = 0; v < 4; ++v) {
            const int nti = v & 1, nto = v >> 1;
            for (int i = 0; i < 3; ++i) {
                launch_copy(nti, 0, in, ws, mb << 20);
                launch_copy(0, nto, ws, out, mb << 20);
            }
            HIPCHECK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < 20; ++i) {
                launch_copy(nti, 0, in, ws, mb << 20);
                launch_copy(0, nto, ws, out, mb << 20);
            }
            HIPCHECK(hipEventRecord(e1, nullptr));
            HIPCHECK(hipEventSynchronize(e1));
            float ms;
            HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("CALIB in->ws->out copies, %zu MiB arrays, nt_in=%d nt_out=%d: %.1f us per pair -> %.0f GB/s algorithmic\n", mb, nti,
                   nto, ms / 20 * 1e3, 4.0 * double(mb << 20) / (ms / 20) / 1e6);
        }
        HIPCHECK(hipFree(in));
        HIPCHECK(hipFree(ws));
        HIPCHECK(hipFree(out));
    }
}

// ---- timing ---------------------------------------------------------------------------------
static double time_copy(size_t bytes) {
    void *a, *b;
    HIPCHECK(hipMalloc(&a, bytes));
    HIPCHECK(hipMalloc(&b, bytes));
    HIPCHECK(hipMemset(a, 1, bytes));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) HIPCHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, nullptr));
    HIPCHECK(hipEventRecord(e0, nullptr));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) HIPCHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, nullptr));
    HIPCHECK(hipEventRecord(e1, nullptr));
    HIPCHECK(hipEventSynchronize(e1));
    float ms;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    HIPCHECK(hipFree(a));
    HIPCHECK(hipFree(b));
    return double(ms) / reps;
}

template <typename T>
static void bench_fft2(int64_t n, int64_t in_n, int epi) {
    pm_fft2_desc d;
    memset(&d, 0, sizeof d);
    d.dtype = sizeof(T) == 4 ? PM_C64 : PM_C128;
    d.direction = -1;
    d.epilogue = epi;
    d.scale = 1.0 / double(n);
    d.weight = 1.0;
    d.in_y = d.in_x = {n, in_n, (n - in_n + 1) / 2, n / 2};
    d.out_y = d.out_x = {n, n, 0, n / 2};
    d.in_ld = in_n;
    d.out_ld = n;
    const size_t es = 2 * sizeof(T);
    std::vector<std::complex<T>> hx(size_t(in_n) * in_n);
    std::mt19937 rng(n);
    std::normal_distribution<float> nd;
    for (auto& e : hx) e = std::complex<T>(nd(rng), nd(rng));
    void *din, *dout, *ws;
    const size_t wsb = pm_fft2_workspace(&d);
    HIPCHECK(hipMalloc(&din, hx.size() * es));
    HIPCHECK(hipMalloc(&dout, size_t(n) * n * es));
    HIPCHECK(hipMalloc(&ws, wsb));
    HIPCHECK(hipMemcpy(din, hx.data(), hx.size() * es, hipMemcpyHostToDevice));
    double ms[2] = {0, 0};
    int rc = pm_fft2_time_passes(&d, din, dout, ws, wsb, 20, ms, nullptr);
    // whole transform, back to back
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) pm_fft2(&d, din, dout, ws, wsb, nullptr);
    HIPCHECK(hipEventRecord(e0, nullptr));
    const int reps = 50;
    for (int i = 0; i < reps; ++i) pm_fft2(&d, din, dout, ws, wsb, nullptr);
    HIPCHECK(hipEventRecord(e1, nullptr));
    HIPCHECK(hipEventSynchronize(e1));
    float tot;
    HIPCHECK(hipEventElapsedTime(&tot, e0, e1));
    tot /= reps;
    const double alg = 4.0 * double(n) * n * es;   // graded bytes: 2 passes x (read + write) of the transform size
    const double b1 = (double(in_n) * in_n + double(in_n) * n) * es;                       // actual pass-1 traffic
    const double b2 = (double(in_n) * n) * es + double(n) * n * (epi ? es / 2 : es);       // actual pass-2 traffic
    printf("BENCH fft2 %s N=%lld in=%lld epi=%d rc=%d: pass1 %.1f us (%.0f GB/s actual) pass2 %.1f us (%.0f GB/s actual) total %.1f us "
           "-> %.0f GB/s algorithmic (%.1f%% of 8 TB/s), %.0f props/s\n",
           sizeof(T) == 4 ? "c64" : "c128", (long long)n, (long long)in_n, epi, rc, ms[0] * 1e3, b1 / ms[0] / 1e6, ms[1] * 1e3,
           b2 / ms[1] / 1e6, tot * 1e3, alg / tot / 1e6, alg / tot / 1e6 / 8000 * 100, 1e3 / tot);
    HIPCHECK(hipFree(din));
    HIPCHECK(hipFree(dout));
    HIPCHECK(hipFree(ws));
}

// sweep of the tuning knobs on one transform size: configurations are timed round-robin for several
// rounds in ONE process (interleaved A/B, cdna guide rule 24); min and median of the whole-transform time
struct Knobs {
    int row_var, col_var, nt_in, nt_out, log_k, row_log_g;
};
static void apply(const Knobs& k) {
    pm_set_tuning("row_var", k.row_var);
    pm_set_tuning("col_var", k.col_var);
    pm_set_tuning("nt_in", k.nt_in);
    pm_set_tuning("nt_out", k.nt_out);
    pm_set_tuning("log_k", k.log_k);
    pm_set_tuning("row_log_g", k.row_log_g);
}
template <typename T>
static void sweep_fft2(int64_t n, const std::vector<Knobs>& cfgs, int rounds) {
    pm_fft2_desc d;
    memset(&d, 0, sizeof d);
    d.dtype = sizeof(T) == 4 ? PM_C64 : PM_C128;
    d.direction = -1;
    d.scale = 1.0 / double(n);
    d.weight = 1.0;
    d.in_y = d.in_x = d.out_y = d.out_x = {n, n, 0, n / 2};
    d.in_ld = d.out_ld = n;
    const size_t es = 2 * sizeof(T);
    std::vector<std::complex<T>> hx(size_t(n) * n);
    std::mt19937 rng(n);
    std::normal_distribution<float> nd;
    for (auto& e : hx) e = std::complex<T>(nd(rng), nd(rng));
    void *din, *dout, *ws;
    const size_t wsb = size_t(n) * n * es;
    HIPCHECK(hipMalloc(&din, hx.size() * es));
    HIPCHECK(hipMalloc(&dout, size_t(n) * n * es));
    HIPCHECK(hipMalloc(&ws, wsb));
    HIPCHECK(hipMemcpy(din, hx.data(), hx.size() * es, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    std::vector<std::vector<double>> t(cfgs.size());
    std::vector<double> p1(cfgs.size()), p2(cfgs.size());
    const int reps = 40;
    for (int r = 0; r < rounds; ++r)
        for (size_t c = 0; c < cfgs.size(); ++c) {
            apply(cfgs[c]);
            for (int i = 0; i < 5; ++i) pm_fft2(&d, din, dout, ws, wsb, nullptr);
            HIPCHECK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) pm_fft2(&d, din, dout, ws, wsb, nullptr);
            HIPCHECK(hipEventRecord(e1, nullptr));
            HIPCHECK(hipEventSynchronize(e1));
            float ms;
            HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            t[c].push_back(double(ms) / reps * 1e3);
            if (r == rounds - 1) {
                double pm[2];
                pm_fft2_time_passes(&d, din, dout, ws, wsb, 20, pm, nullptr);
                p1[c] = pm[0] * 1e3;
                p2[c] = pm[1] * 1e3;
            }
        }
    const double alg = 4.0 * double(n) * n * es;
    for (size_t c = 0; c < cfgs.size(); ++c) {
        std::sort(t[c].begin(), t[c].end());
        const double mn = t[c].front(), med = t[c][t[c].size() / 2];
        const Knobs& k = cfgs[c];
        printf("SWEEP %s N=%lld row_var=%d col_var=%d nt_in=%d nt_out=%d log_k=%d row_log_g=%d : total min %.1f med %.1f us "
               "(%.1f%% of 8TB/s at min) ; passes %.1f + %.1f us\n",
               sizeof(T) == 4 ? "c64" : "c128", (long long)n, k.row_var, k.col_var, k.nt_in, k.nt_out, k.log_k, k.row_log_g, mn, med,
               alg / mn / 1e6 / 8000 * 100, p1[c], p2[c]);
    }
    apply(Knobs{-1, 0, -1, -1, -1, 1});
    HIPCHECK(hipFree(din));
    HIPCHECK(hipFree(dout));
    HIPCHECK(hipFree(ws));
}

// generic interleaved A/B: "tune N c64|c128 rounds cfg cfg ..." with cfg = "key=val,key=val" (pm_set_tuning keys; keys not
// named in a cfg are reset to their defaults first)
static void set_cfg(const std::string& cfg) {
    static const char* defaults = "row_var=-1,col_var=-1,nt_in=-1,nt_out=-1,log_k=-1,row_log_g=1,fold=-1";
    for (const std::string& src : {std::string(defaults), cfg}) {
        size_t i = 0;
        while (i < src.size()) {
            size_t c = src.find(',', i);
            if (c == std::string::npos) c = src.size();
            const std::string kv = src.substr(i, c - i);
            const size_t eq = kv.find('=');
            if (eq != std::string::npos) pm_set_tuning(kv.substr(0, eq).c_str(), atoi(kv.c_str() + eq + 1));
            i = c + 1;
        }
    }
}
template <typename T>
static void tune_fft2(int64_t n, int rounds, const std::vector<std::string>& cfgs) {
    pm_fft2_desc d;
    memset(&d, 0, sizeof d);
    d.dtype = sizeof(T) == 4 ? PM_C64 : PM_C128;
    d.direction = -1;
    d.scale = 1.0 / double(n);
    d.weight = 1.0;
    d.in_y = d.in_x = d.out_y = d.out_x = {n, n, 0, n / 2};
    // PM_LD_PAD=k: leading dimensions n + k (experiment: power-of-two row strides vs the HBM channel mapping)
    const int64_t pad = getenv("PM_LD_PAD") ? atoll(getenv("PM_LD_PAD")) : 0;
    d.in_ld = d.out_ld = n + pad;
    const size_t es = 2 * sizeof(T);
    std::vector<std::complex<T>> hx(size_t(n) * (n + pad));
    std::mt19937 rng(n);
    std::normal_distribution<float> nd;
    for (auto& e : hx) e = std::complex<T>(nd(rng), nd(rng));
    void *din, *dout, *ws;
    const size_t wsb = size_t(n) * n * es;
    HIPCHECK(hipMalloc(&din, hx.size() * es));
    HIPCHECK(hipMalloc(&dout, size_t(n) * (n + pad) * es));
    HIPCHECK(hipMalloc(&ws, wsb));
    HIPCHECK(hipMemcpy(din, hx.data(), hx.size() * es, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    std::vector<std::vector<double>> t(cfgs.size()), p1(cfgs.size()), p2(cfgs.size());
    const int reps = 40;
    for (int r = 0; r < rounds; ++r)
        for (size_t c = 0; c < cfgs.size(); ++c) {
            set_cfg(cfgs[c]);
            for (int i = 0; i < 5; ++i) pm_fft2(&d, din, dout, ws, wsb, nullptr);
            HIPCHECK(hipEventRecord(e0, nullptr));
            for (int i = 0; i < reps; ++i) pm_fft2(&d, din, dout, ws, wsb, nullptr);
            HIPCHECK(hipEventRecord(e1, nullptr));
            HIPCHECK(hipEventSynchronize(e1));
            float ms;
            HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
            t[c].push_back(double(ms) / reps * 1e3);
            double pm[2];
            pm_fft2_time_passes(&d, din, dout, ws, wsb, 20, pm, nullptr);
            p1[c].push_back(pm[0] * 1e3);
            p2[c].push_back(pm[1] * 1e3);
        }
    const double alg = 4.0 * double(n) * n * es;
    for (size_t c = 0; c < cfgs.size(); ++c) {
        std::sort(t[c].begin(), t[c].end());
        std::sort(p1[c].begin(), p1[c].end());
        std::sort(p2[c].begin(), p2[c].end());
        printf("TUNE %s N=%lld [%s]: total min %.1f med %.1f us (%.1f%% of 8TB/s at min); row pass min %.1f med %.1f; column pass min %.1f med %.1f\n",
               sizeof(T) == 4 ? "c64" : "c128", (long long)n, cfgs[c].c_str(), t[c].front(), t[c][t[c].size() / 2],
               alg / t[c].front() / 1e3 / 8000 * 100, p1[c].front(), p1[c][p1[c].size() / 2], p2[c].front(), p2[c][p2[c].size() / 2]);
    }
    set_cfg("");
    HIPCHECK(hipFree(din));
    HIPCHECK(hipFree(dout));
    HIPCHECK(hipFree(ws));
}

// fused fft2 -> x H -> ifft2 (separable H) against the two-call composition: same result, fewer passes
template <typename T>
static void check_bench_fused(int64_t n, int64_t in_n, bool timeit) {
    pm_fft2_desc d;
    memset(&d, 0, sizeof d);
    d.dtype = sizeof(T) == 4 ? PM_C64 : PM_C128;
    d.direction = -1;
    d.scale = 1.0 / (double(n) * n);
    d.weight = 1.0;
    d.in_y = d.in_x = {n, in_n, (n - in_n + 1) / 2, 0};
    d.out_y = d.out_x = {n, n, 0, 0};
    d.in_ld = in_n;
    d.out_ld = n;
    d.mul_kind = PM_MUL_SEPARABLE;
    const size_t es = 2 * sizeof(T);
    std::vector<std::complex<T>> hx(size_t(in_n) * in_n);
    std::mt19937 rng(n + 1);
    std::normal_distribution<float> nd;
    for (auto& e : hx) e = std::complex<T>(nd(rng), nd(rng));
    void *din, *dout, *dout2, *dtmp, *ws, *hy, *hxv;
    HIPCHECK(hipMalloc(&din, hx.size() * es));
    HIPCHECK(hipMalloc(&dout, size_t(n) * n * es));
    HIPCHECK(hipMalloc(&dout2, size_t(n) * n * es));
    HIPCHECK(hipMalloc(&dtmp, size_t(n) * n * es));
    HIPCHECK(hipMalloc(&hy, size_t(n) * es));
    HIPCHECK(hipMalloc(&hxv, size_t(n) * es));
    HIPCHECK(hipMemcpy(din, hx.data(), hx.size() * es, hipMemcpyHostToDevice));
    pm_as_tf_vectors(d.dtype, n, n, 0.6328, 0.01, 10.0, hy, hxv, nullptr);
    d.mul = hy;
    d.mul_x = hxv;
    const size_t wsb = std::max(pm_fft2_mul_ifft2_workspace(&d), size_t(n) * n * es);
    HIPCHECK(hipMalloc(&ws, wsb));
    int rc = pm_fft2_mul_ifft2(&d, din, dout, ws, wsb, nullptr);
    // composition: forward with multiplier on the store, then inverse
    pm_fft2_desc a = d, b = d;
    a.scale = 1.0;
    b.direction = +1;
    b.mul_kind = PM_MUL_NONE;
    b.in_y = b.in_x = {n, n, 0, 0};
    b.in_ld = n;
    int rc2 = pm_fft2(&a, din, dtmp, ws, wsb, nullptr);
    int rc3 = pm_fft2(&b, dtmp, dout2, ws, wsb, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    std::vector<std::complex<T>> o1(size_t(n) * n), o2(size_t(n) * n);
    HIPCHECK(hipMemcpy(o1.data(), dout, o1.size() * es, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(o2.data(), dout2, o2.size() * es, hipMemcpyDeviceToHost));
    double err = 0, nrm = 0;
    for (size_t i = 0; i < o1.size(); ++i) {
        err = fmax(err, std::abs(std::complex<double>(o1[i]) - std::complex<double>(o2[i])));
        nrm = fmax(nrm, std::abs(std::complex<double>(o2[i])));
    }
    char name[128];
    snprintf(name, sizeof name, "fused fft2*H ifft2 %s N=%lld in=%lld rc=%d/%d/%d (vs two-call composition)", sizeof(T) == 4 ? "c64" : "c128",
             (long long)n, (long long)in_n, rc, rc2, rc3);
    report(name, (rc || rc2 || rc3) ? 1.0 : err / nrm, sizeof(T) == 4 ? 3e-6 : 1e-13);
    if (timeit) {
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreate(&e0));
        HIPCHECK(hipEventCreate(&e1));
        float t3, t4;
        for (int i = 0; i < 3; ++i) pm_fft2_mul_ifft2(&d, din, dout, ws, wsb, nullptr);
        HIPCHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 20; ++i) pm_fft2_mul_ifft2(&d, din, dout, ws, wsb, nullptr);
        HIPCHECK(hipEventRecord(e1, nullptr));
        HIPCHECK(hipEventSynchronize(e1));
        HIPCHECK(hipEventElapsedTime(&t3, e0, e1));
        HIPCHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < 20; ++i) {
            pm_fft2(&a, din, dtmp, ws, wsb, nullptr);
            pm_fft2(&b, dtmp, dout2, ws, wsb, nullptr);
        }
        HIPCHECK(hipEventRecord(e1, nullptr));
        HIPCHECK(hipEventSynchronize(e1));
        HIPCHECK(hipEventElapsedTime(&t4, e0, e1));
        const double alg = 8.0 * double(n) * n * es;
        printf("BENCH angular-spectrum step %s N=%lld in=%lld: fused 3-pass %.1f us (%.0f GB/s on 8N^2s = %.1f%% of 8 TB/s; %.0f GB/s on 6N^2s), "
               "two-call 4-pass %.1f us (%.1f%%)\n",
               sizeof(T) == 4 ? "c64" : "c128", (long long)n, (long long)in_n, t3 / 20 * 1e3, alg / (t3 / 20) / 1e6, alg / (t3 / 20) / 1e6 / 80,
               0.75 * alg / (t3 / 20) / 1e6, t4 / 20 * 1e3, alg / (t4 / 20) / 1e6 / 80);
    }
    for (void* p : {din, dout, dout2, dtmp, ws, hy, hxv}) HIPCHECK(hipFree(p));
}

template <typename T>
static void bench_cgemm(int64_t M, int64_t N, int64_t K, int opB) {
    const size_t es = 2 * sizeof(T);
    const int dt = sizeof(T) == 4 ? PM_C64 : PM_C128;
    void *dA, *dB, *dC, *ws = nullptr;
    size_t wsb = pm_cgemm_workspace(dt, M, N, K);
    HIPCHECK(hipMalloc(&dA, size_t(M) * K * es));
    HIPCHECK(hipMalloc(&dB, size_t(K) * N * es));
    HIPCHECK(hipMalloc(&dC, size_t(M) * N * es));
    if (wsb) HIPCHECK(hipMalloc(&ws, wsb));
    std::vector<std::complex<T>> h(size_t(std::max(M, N)) * K);
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> ud(-1, 1);
    for (auto& e : h) e = std::complex<T>(ud(rng), ud(rng));
    HIPCHECK(hipMemcpy(dA, h.data(), size_t(M) * K * es, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dB, h.data(), size_t(K) * N * es, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0));
    HIPCHECK(hipEventCreate(&e1));
    const int64_t ldb = (opB & 2) ? K : N;
    for (int i = 0; i < 2; ++i) pm_cgemm(dt, 0, opB, M, N, K, 1.0, dA, K, dB, ldb, dC, N, ws, wsb, nullptr);
    HIPCHECK(hipEventRecord(e0, nullptr));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) pm_cgemm(dt, 0, opB, M, N, K, 1.0, dA, K, dB, ldb, dC, N, ws, wsb, nullptr);
    HIPCHECK(hipEventRecord(e1, nullptr));
    HIPCHECK(hipEventSynchronize(e1));
    float ms;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double fl = 8.0 * M * N * K;
    printf("BENCH cgemm %s %lldx%lldx%lld opB=%d ws=%zu: %.1f us -> %.1f TFLOP/s\n", sizeof(T) == 4 ? "c64" : "c128", (long long)M,
           (long long)N, (long long)K, opB, wsb, ms * 1e3, fl / ms / 1e9);
    HIPCHECK(hipFree(dA));
    HIPCHECK(hipFree(dB));
    HIPCHECK(hipFree(dC));
    if (ws) HIPCHECK(hipFree(ws));
}

// Batched transforms: (1) a batch must reproduce the field-by-field results bit for bit, (2) time per field
// against the batch size (small fields are launch / latency bound one at a time).
template <typename T>
static void check_bench_batch(int64_t n, int64_t in_n, int batch, int epi, bool timeit) {
    pm_fft2_desc d;
    memset(&d, 0, sizeof d);
    d.dtype = sizeof(T) == 4 ? PM_C64 : PM_C128;
    d.direction = -1;
    d.epilogue = epi;
    d.scale = 1.0 / double(n);
    d.weight = 1.0;
    d.in_y = d.in_x = {n, in_n, (n - in_n + 1) / 2, n / 2};
    d.out_y = d.out_x = {n, n, 0, n / 2};
    d.in_ld = in_n;
    d.out_ld = n;
    pm_fft2_desc db = d;
    db.batch = batch;
    db.in_bstride = in_n * in_n;
    db.out_bstride = n * n;
    const size_t es = 2 * sizeof(T), oes = epi ? sizeof(T) : es;
    std::vector<std::complex<T>> hx(size_t(batch) * in_n * in_n);
    std::mt19937 rng(n + batch);
    std::normal_distribution<float> nd;
    for (auto& e : hx) e = std::complex<T>(nd(rng), nd(rng));
    void *din, *dout, *dref, *ws;
    const size_t wsb = std::max(pm_fft2_workspace(&db), pm_fft2_workspace(&d));
    const size_t out_bytes = size_t(batch) * n * n * oes;
    HIPCHECK(hipMalloc(&din, hx.size() * es));
    HIPCHECK(hipMalloc(&dout, out_bytes));
    HIPCHECK(hipMalloc(&dref, out_bytes));
    HIPCHECK(hipMalloc(&ws, wsb));
    HIPCHECK(hipMemcpy(din, hx.data(), hx.size() * es, hipMemcpyHostToDevice));
    HIPCHECK(hipMemset(dout, 0xff, out_bytes));
    int rc = pm_fft2(&db, din, dout, ws, wsb, nullptr);
    for (int b = 0; b < batch && !rc; ++b)
        rc = pm_fft2(&d, (char*)din + size_t(b) * in_n * in_n * es, (char*)dref + size_t(b) * n * n * oes, ws, wsb, nullptr);
    HIPCHECK(hipDeviceSynchronize());
    std::vector<char> a(out_bytes), r(out_bytes);
    HIPCHECK(hipMemcpy(a.data(), dout, out_bytes, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(r.data(), dref, out_bytes, hipMemcpyDeviceToHost));
    char name[160];
    snprintf(name, sizeof name, "batch %s N=%lld in=%lld B=%d epi=%d (bitwise vs field-by-field) rc=%d", sizeof(T) == 4 ? "c64" : "c128",
             (long long)n, (long long)in_n, batch, epi, rc);
    report(name, (rc || memcmp(a.data(), r.data(), out_bytes)) ? 1.0 : 0.0, 0.5);
    if (timeit) {
        hipEvent_t e0, e1;
        HIPCHECK(hipEventCreate(&e0));
        HIPCHECK(hipEventCreate(&e1));
        float tb = 0, t1 = 0;
        const int reps = 20;
        for (int i = 0; i < 3; ++i) pm_fft2(&db, din, dout, ws, wsb, nullptr);
        HIPCHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i) pm_fft2(&db, din, dout, ws, wsb, nullptr);
        HIPCHECK(hipEventRecord(e1, nullptr));
        HIPCHECK(hipEventSynchronize(e1));
        HIPCHECK(hipEventElapsedTime(&tb, e0, e1));
        HIPCHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i)
            for (int b = 0; b < batch; ++b)
                pm_fft2(&d, (char*)din + size_t(b) * in_n * in_n * es, (char*)dref + size_t(b) * n * n * oes, ws, wsb, nullptr);
        HIPCHECK(hipEventRecord(e1, nullptr));
        HIPCHECK(hipEventSynchronize(e1));
        HIPCHECK(hipEventElapsedTime(&t1, e0, e1));
        const double alg = 4.0 * double(n) * n * es;
        const double usb = tb / reps / batch * 1e3, us1 = t1 / reps / batch * 1e3;
        printf("BENCH batch %s N=%lld in=%lld B=%d epi=%d: %.2f us/field batched (%.0f GB/s algorithmic, %.1f%% of 8 TB/s) vs %.2f us/field "
               "one at a time (%.1f%%)\n", sizeof(T) == 4 ? "c64" : "c128", (long long)n, (long long)in_n, batch, epi, usb, alg / usb / 1e3,
               alg / usb / 1e3 / 8000 * 100, us1, alg / us1 / 1e3 / 8000 * 100);
    }
    HIPCHECK(hipFree(din));
    HIPCHECK(hipFree(dout));
    HIPCHECK(hipFree(dref));
    HIPCHECK(hipFree(ws));
}

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "quick";
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s (%s) CUs=%d LDS/block=%zu version=%d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount,
           prop.sharedMemPerBlock, pm_version());
    if (mode == "big") {
        const std::vector<Knobs> cfg = {{-1, 0, -1, -1, -1, 1}, {-1, 0, -1, 0, -1, 1}, {-1, 0, 0, 0, -1, 1}, {-1, 0, -1, 0, 3, 1}, {-1, 0, -1, 0, 1, 1}};
        sweep_fft2<float>(8192, cfg, 2);
        return 0;
    }
    if (mode == "tune" && argc >= 6) {
        const int64_t n = atoll(argv[2]);
        const int rounds = atoi(argv[4]);
        std::vector<std::string> cfgs;
        for (int i = 5; i < argc; ++i) cfgs.push_back(argv[i]);
        if (!strcmp(argv[3], "c128")) tune_fft2<double>(n, rounds, cfgs);
        else tune_fft2<float>(n, rounds, cfgs);
        return 0;
    }
    if (mode == "batch") {
        check_bench_batch<float>(64, 40, 5, 0, false);
        check_bench_batch<double>(128, 128, 3, 1, false);
        check_bench_batch<float>(100, 60, 3, 0, false);    // direct-DFT sizes: field by field inside the library
        pm_set_tuning("batch_ws_mib", 1);                  // force several chunks
        check_bench_batch<float>(512, 256, 7, 0, false);
        pm_set_tuning("batch_ws_mib", 128);
        for (int64_t n : {256, 512, 1024, 2048})
            for (int b : {1, 4, 16, 64}) {
                if (n * n * 8 * b > (int64_t(1) << 30)) continue;
                check_bench_batch<float>(n, n, b, 0, true);
            }
        for (int mib : {16, 32, 128, 1024}) {
            pm_set_tuning("batch_ws_mib", mib);
            printf("batch_ws_mib = %d\n", mib);
            check_bench_batch<float>(1024, 1024, 64, 0, true);
            check_bench_batch<float>(512, 512, 64, 0, true);
        }
        pm_set_tuning("batch_ws_mib", 128);
        check_bench_batch<float>(1024, 512, 16, 1, true);   // Q = 2 PSFs, fused |.|^2
        check_bench_batch<double>(1024, 1024, 16, 0, true);
        printf(g_fail ? "GPU CHECK FAILED (%d)\n" : "GPU CHECK OK\n", g_fail);
        return g_fail ? 1 : 0;
    }
    if (mode == "fused") {
        check_bench_fused<float>(64, 64, false);
        check_bench_fused<float>(256, 128, false);
        check_bench_fused<double>(128, 100, false);
        check_bench_fused<float>(2048, 2048, true);
        check_bench_fused<float>(4096, 4096, true);
        check_bench_fused<double>(2048, 2048, true);
        check_bench_fused<double>(4096, 4096, true);
        check_bench_fused<double>(4096, 2048, true);
        printf(g_fail ? "GPU CHECK FAILED (%d)\n" : "GPU CHECK OK\n", g_fail);
        return g_fail ? 1 : 0;
    }
    if (mode == "gemm") {
        for (int opA = 0; opA < 4; ++opA)
            for (int opB = 0; opB < 4; ++opB) {
                check_cgemm<float>(opA, opB, 70, 45, 100, 2e-5, false);
                check_cgemm<double>(opA, opB, 33, 50, 37, 1e-13, false);
            }
        // the 128 x 128 kernel: every op pair, slab depths of 1 .. 5 K-tiles, split and unsplit
        for (int opA = 0; opA < 4; ++opA)
            for (int opB = 0; opB < 4; ++opB) {
                check_cgemm<float>(opA, opB, 256, 128, 48, 2e-5, true);
                check_cgemm<float>(opA, opB, 192, 64, 80, 2e-5, true);      // 64 x 64 tiles
            }
        pm_set_tuning("gemm_tile", 128);
        for (int opA = 0; opA < 4; ++opA)
            for (int opB = 0; opB < 4; ++opB) check_cgemm<float>(opA, opB, 256, 128, 48, 2e-5, true);
        pm_set_tuning("gemm_tile", 0);
        for (int64_t K : {16, 32, 64, 80, 256}) {
            check_cgemm<float>(0, 0, 128, 256, K, 2e-5, true);
            check_cgemm<float>(3, 2, 128, 128, K, 2e-5, false);
        }
        check_cgemm<float>(0, 0, 512, 2048, 2048, 5e-5, true);
        check_cgemm<float>(0, 2, 512, 512, 2048, 5e-5, true);
        check_cgemm<float>(3, 0, 2048, 512, 512, 5e-5, true);
        check_cgemm<float>(0, 1, 2048, 2048, 512, 5e-5, true);
        check_cgemm<double>(3, 1, 200, 300, 1000, 1e-12, true);
        for (int tile : {0, 64, 128}) {
            pm_set_tuning("gemm_tile", tile);
            printf("gemm_tile=%d\n", tile);
            bench_cgemm<float>(512, 2048, 2048, 0);
            bench_cgemm<float>(512, 512, 2048, 2);
            bench_cgemm<float>(2048, 512, 512, 0);
            bench_cgemm<float>(512, 4096, 4096, 0);
            bench_cgemm<float>(512, 512, 4096, 2);
        }
        pm_set_tuning("gemm_tile", 0);
        bench_cgemm<double>(512, 2048, 2048, 0);
        pm_set_tuning("gemm_min_wgs", 1024);
        pm_set_tuning("gemm_bk", 32);
        bench_cgemm<float>(4096, 4096, 4096, 0);
        bench_cgemm<double>(2048, 2048, 2048, 0);
        pm_set_tuning("gemm_bk", 0);
        bench_cgemm<float>(4096, 4096, 4096, 0);
        bench_cgemm<double>(2048, 2048, 2048, 0);
        printf(g_fail ? "GPU CHECK FAILED (%d)\n" : "GPU CHECK OK\n", g_fail);
        return g_fail ? 1 : 0;
    }
    if (mode == "calib") {
        calibrate();
        return 0;
    }
    if (mode == "gemmk") {   // time against K: per-iteration cost and fixed overhead of the config-4 GEMM
        for (int64_t K : {256, 512, 1024, 2048, 4096, 8192}) bench_cgemm<float>(512, 2048, K, 0);
        pm_set_tuning("gemm_min_wgs", 1);   // no split-K
        for (int64_t K : {512, 2048, 8192}) bench_cgemm<float>(512, 2048, K, 0);
        return 0;
    }
    if (mode == "gemmprof") {   // the large config-4 GEMM only (PMC passes)
        bench_cgemm<float>(512, 2048, 2048, 0);
        return 0;
    }
    if (mode == "prof") {   // short, fixed workload for rocprofv3 (kernel trace / PMC passes)
        bench_fft2<float>(4096, 4096, 0);
        bench_fft2<double>(4096, 4096, 0);
        bench_cgemm<float>(512, 2048, 2048, 0);
        bench_cgemm<float>(512, 512, 2048, 2);
        return 0;
    }
    if (mode == "sweep") {
        const std::vector<Knobs> c64 = {
            {0, 0, -1, -1, -1, 1}, {1, 0, -1, -1, -1, 1}, {2, 0, -1, -1, -1, 1},
        };
        sweep_fft2<float>(4096, c64, 3);
        const std::vector<Knobs> c128 = {
            {-1, 0, -1, -1, -1, 1},
        };
        sweep_fft2<double>(4096, c128, 2);
        const std::vector<Knobs> small = {{0, 0, -1, -1, -1, 1}, {1, 0, -1, -1, -1, 1}};
        sweep_fft2<float>(2048, small, 3);
        sweep_fft2<double>(2048, small, 3);
        return 0;
    }
    if (mode != "bench") {
        const Case2 cases[] = {
            {16, 16, 16, 16, 16, 16, true, -1, 0},   {8, 8, 8, 8, 8, 8, false, -1, 0},       {64, 32, 64, 32, 64, 32, true, -1, 0},
            {32, 64, 16, 32, 32, 64, true, -1, 0},   {32, 64, 32, 64, 16, 32, true, 1, 0},   {128, 128, 128, 128, 128, 128, true, -1, 1},
            {9, 12, 9, 12, 9, 12, true, -1, 0},      {14, 18, 9, 12, 14, 18, true, -1, 0},   {14, 18, 14, 18, 9, 12, true, 1, 0},
            {7, 9, 7, 9, 7, 9, true, 1, 0},          {16, 12, 16, 12, 16, 12, true, -1, 0},  {12, 16, 12, 16, 12, 16, false, 1, 0},
            {256, 256, 128, 128, 256, 256, true, -1, 0}, {512, 512, 512, 512, 512, 512, true, -1, 0},
            {1024, 1024, 1024, 1024, 1024, 1024, false, -1, 0}, {1024, 2048, 512, 1024, 1024, 2048, true, -1, 1},
            {2, 4, 2, 4, 2, 4, true, -1, 0},         {4, 2, 4, 2, 4, 2, false, -1, 0},      {1, 8, 1, 8, 1, 8, false, -1, 0},
            {100, 60, 100, 60, 100, 60, true, -1, 0},
        };
        for (const auto& c : cases) {
            check_fft2<float>(c, 5e-6);
            check_fft2<double>(c, 1e-13);
        }
        if (mode == "full") {
            const Case2 big[] = {{2048, 2048, 2048, 2048, 2048, 2048, true, -1, 0}, {4096, 4096, 4096, 4096, 4096, 4096, true, -1, 0},
                                 {4096, 4096, 2048, 2048, 4096, 4096, true, -1, 1}, {8192, 8192, 8192, 8192, 8192, 8192, false, 1, 0}};
            for (const auto& c : big) check_fft2<float>(c, 5e-6);
            const Case2 bigd[] = {{2048, 2048, 2048, 2048, 2048, 2048, true, -1, 0}, {4096, 4096, 4096, 4096, 4096, 4096, true, 1, 0},
                                  {8192, 4096, 8192, 4096, 8192, 4096, false, -1, 0}};
            for (const auto& c : bigd) check_fft2<double>(c, 1e-13);
        }
        check_fft1<float>(64, 40, 33, 1, -1, 5e-6);
        check_fft1<float>(64, 40, 33, 0, -1, 5e-6);
        check_fft1<float>(50, 50, 7, 1, 1, 5e-6);
        check_fft1<float>(50, 30, 7, 0, -1, 5e-6);
        check_fft1<double>(1024, 700, 19, 0, 1, 1e-13);
        check_fft1<double>(4096, 4096, 5, 1, -1, 1e-13);
        check_fft1<float>(4096, 2500, 24, 0, -1, 5e-6);
        for (int opA = 0; opA < 4; ++opA)
            for (int opB = 0; opB < 4; ++opB) {
                check_cgemm<float>(opA, opB, 70, 45, 100, 2e-5, false);
                check_cgemm<double>(opA, opB, 33, 50, 37, 1e-13, false);
            }
        // the 128 x 128 kernel: every op pair, slab depths of 1 .. 5 K-tiles, split and unsplit
        for (int opA = 0; opA < 4; ++opA)
            for (int opB = 0; opB < 4; ++opB) {
                check_cgemm<float>(opA, opB, 256, 128, 48, 2e-5, true);
                check_cgemm<float>(opA, opB, 192, 64, 80, 2e-5, true);      // 64 x 64 tiles
            }
        pm_set_tuning("gemm_tile", 128);
        for (int opA = 0; opA < 4; ++opA)
            for (int opB = 0; opB < 4; ++opB) check_cgemm<float>(opA, opB, 256, 128, 48, 2e-5, true);
        pm_set_tuning("gemm_tile", 0);
        for (int64_t K : {16, 32, 64, 80, 256}) {
            check_cgemm<float>(0, 0, 128, 256, K, 2e-5, true);
            check_cgemm<float>(3, 2, 128, 128, K, 2e-5, false);
        }
        check_cgemm<float>(0, 0, 512, 2048, 2048, 5e-5, true);
        check_cgemm<float>(0, 2, 512, 512, 2048, 5e-5, true);
        check_cgemm<double>(3, 1, 200, 300, 1000, 1e-12, true);
    }
    if (mode != "quick") {
        for (size_t mb : {128, 256, 1024}) {
            const double ms = time_copy(mb << 20);
            printf("BENCH d2d copy %zu MiB: %.1f us -> %.0f GB/s (read+write)\n", mb, ms * 1e3, 2.0 * double(mb << 20) / ms / 1e6);
        }
        bench_fft2<float>(2048, 2048, 0);
        bench_fft2<float>(4096, 4096, 0);
        bench_fft2<float>(4096, 4096, 1);
        bench_fft2<float>(4096, 2048, 0);
        bench_fft2<float>(8192, 8192, 0);
        bench_fft2<double>(2048, 2048, 0);
        bench_fft2<double>(4096, 4096, 0);
        bench_fft2<float>(1024, 1024, 0);
        bench_fft2<float>(512, 512, 0);
        bench_cgemm<float>(512, 2048, 2048, 0);
        bench_cgemm<float>(512, 512, 2048, 2);
        bench_cgemm<float>(4096, 4096, 4096, 0);
        bench_cgemm<double>(512, 2048, 2048, 0);
        bench_cgemm<double>(2048, 2048, 2048, 0);
    }
    printf(g_fail ? "GPU CHECK FAILED (%d)\n" : "GPU CHECK OK\n", g_fail);
    return g_fail ? 1 : 0;
}
