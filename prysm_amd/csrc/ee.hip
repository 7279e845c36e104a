// Encircled energy of a PSF from its MTF (Baliga & Cohn 1988), prysm/otf.py:319-472:
//     EE(r) = r * dnx * dny * sum_ij MTF[i][j] * J1(2 pi r nu_ij) / nu_ij,     nu = hypot of the FFT-centred frequency grid
// and the adjoint's MTF-plane gradient  mtf_bar[i][j] = sum_r ee_bar_r * r * J1(2 pi r nu_ij) / nu_ij * dnx * dny.
// HBM bound on the MTF read (one pass serves up to 8 radii); the Bessel function is evaluated in fp64 for both precisions and the
// sums are accumulated in fp64 in a fixed order (fixed grid, tree reduction, second kernel over the per-workgroup partials), so
// results are reproducible run to run.
#include "pm_internal.h"

#define PM_STREAM(s) reinterpret_cast<hipStream_t>(s)

namespace pm {

constexpr int kEeMaxRadii = 8;
constexpr int kEeBlocks = 1024;
constexpr int kEeThreads = 256;

struct EeRadii {
    double r[kEeMaxRadii];   // radii in mm (the reference divides its micron radii by 1e3, otf.py:381-384)
    double w[kEeMaxRadii];   // adjoint: ee_bar per radius
};

// J1(2 pi r nu) / nu with the zero-frequency bin nudged off zero exactly like the reference (otf.py:339-341)
__device__ __forceinline__ double ee_hankel(double r, double nu) {
    constexpr double two_pi = 6.283185307179586476925286766559;
    return j1(two_pi * r * nu) / nu;
}

__device__ __forceinline__ double ee_nu(int64_t i, int64_t j, int64_t rows, int64_t cols, double df) {
    const double x = double(j - cols / 2) * df, y = double(i - rows / 2) * df;   // make_xy_grid(shape, dx=df): fftrange * dx
    const double nu = hypot(x, y);
    return nu == 0.0 ? 1e-16 : nu;
}

template <typename T>
__global__ void __launch_bounds__(kEeThreads) ee_reduce_kernel(int64_t rows, int64_t cols, const T* mtf, int64_t ld, double df, EeRadii rr,
                                                               int nrad, double* partial) {
    double acc[kEeMaxRadii];
#pragma unroll
    for (int r = 0; r < kEeMaxRadii; ++r) acc[r] = 0.0;
    const int64_t total = rows * cols, step = int64_t(gridDim.x) * blockDim.x;
    for (int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; g < total; g += step) {
        const int64_t i = g / cols, j = g - i * cols;
        const double nu = ee_nu(i, j, rows, cols, df);
        const double m = double(mtf[i * ld + j]);
#pragma unroll
        for (int r = 0; r < kEeMaxRadii; ++r)
            if (r < nrad) acc[r] += m * ee_hankel(rr.r[r], nu);
    }
    __shared__ double red[kEeMaxRadii][kEeThreads / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < kEeMaxRadii; ++r) {
        double v = acc[r];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) red[r][wave] = v;
    }
    __syncthreads();
    if (threadIdx.x < kEeMaxRadii) {
        double v = 0.0;
        for (int w = 0; w < kEeThreads / 64; ++w) v += red[threadIdx.x][w];
        partial[int64_t(blockIdx.x) * kEeMaxRadii + threadIdx.x] = v;
    }
}

// out[r] = radius_r * dnx * dny * sum over the workgroup partials (one workgroup, fixed order)
__global__ void __launch_bounds__(kEeThreads) ee_final_kernel(const double* partial, int nblocks, EeRadii rr, int nrad, double cell, double* out) {
    __shared__ double red[kEeMaxRadii][kEeThreads];
    for (int r = 0; r < kEeMaxRadii; ++r) {
        double v = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += kEeThreads) v += partial[int64_t(b) * kEeMaxRadii + r];
        red[r][threadIdx.x] = v;
    }
    __syncthreads();
    for (int s = kEeThreads / 2; s > 0; s >>= 1) {
        if (int(threadIdx.x) < s)
            for (int r = 0; r < kEeMaxRadii; ++r) red[r][threadIdx.x] += red[r][threadIdx.x + s];
        __syncthreads();
    }
    if (int(threadIdx.x) < nrad) out[threadIdx.x] = rr.r[threadIdx.x] * red[threadIdx.x][0] * cell;
}

template <typename T>
__global__ void ee_adjoint_kernel(int64_t rows, int64_t cols, double df, EeRadii rr, int nrad, double cell, int accumulate, T* out, int64_t ld) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (g >= rows * cols) return;
    const int64_t i = g / cols, j = g - i * cols;
    const double nu = ee_nu(i, j, rows, cols, df);
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < kEeMaxRadii; ++r)
        if (r < nrad) v += rr.w[r] * rr.r[r] * ee_hankel(rr.r[r], nu) * cell;
    T* o = out + i * ld + j;
    *o = accumulate ? T(double(*o) + v) : T(v);
}

}  // namespace pm

using namespace pm;

extern "C" {

size_t pm_encircled_energy_workspace(void) { return size_t(kEeBlocks) * kEeMaxRadii * sizeof(double); }

int pm_encircled_energy(int32_t dtype, int64_t rows, int64_t cols, const void* mtf, int64_t mtf_ld, double df, int64_t nradii,
                        const double* radii_mm, double* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!mtf || !out || (!radii_mm && nradii > 0) || rows < 1 || cols < 1 || nradii < 0 || mtf_ld < cols)
        return fail(PM_ERR_ARG, "pm_encircled_energy: bad argument");
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_encircled_energy: dtype must be PM_C64 (float MTF) or PM_C128 (double)");
    if (!workspace || workspace_bytes < pm_encircled_energy_workspace())
        return fail(PM_ERR_WORKSPACE, "pm_encircled_energy: workspace of %zu bytes required", pm_encircled_energy_workspace());
    hipStream_t st = PM_STREAM(stream);
    double* partial = static_cast<double*>(workspace);
    int64_t nb = (rows * cols + kEeThreads - 1) / kEeThreads;
    if (nb > kEeBlocks) nb = kEeBlocks;
    const double cell = df * df;   // dnx * dny of the square frequency grid (otf.py:342)
    for (int64_t r0 = 0; r0 < nradii; r0 += kEeMaxRadii) {   // 8 radii per pass over the MTF
        const int nr = int(nradii - r0 < kEeMaxRadii ? nradii - r0 : kEeMaxRadii);
        EeRadii rr;
        for (int i = 0; i < kEeMaxRadii; ++i) {
            rr.r[i] = i < nr ? radii_mm[r0 + i] : 0.0;
            rr.w[i] = 0.0;
        }
        if (dtype == PM_C64)
            hipLaunchKernelGGL(ee_reduce_kernel<float>, dim3(unsigned(nb)), dim3(kEeThreads), 0, st, rows, cols, (const float*)mtf, mtf_ld, df, rr, nr, partial);
        else
            hipLaunchKernelGGL(ee_reduce_kernel<double>, dim3(unsigned(nb)), dim3(kEeThreads), 0, st, rows, cols, (const double*)mtf, mtf_ld, df, rr, nr, partial);
        hipLaunchKernelGGL(ee_final_kernel, dim3(1), dim3(kEeThreads), 0, st, partial, int(nb), rr, nr, cell, out + r0);
    }
    return int(hipGetLastError());
}

int pm_encircled_energy_adjoint(int32_t dtype, int64_t rows, int64_t cols, double df, int64_t nradii, const double* radii_mm,
                                const double* ee_bar, void* mtf_bar, int64_t mtf_bar_ld, void* stream) {
    if (!mtf_bar || ((!radii_mm || !ee_bar) && nradii > 0) || rows < 1 || cols < 1 || nradii < 0 || mtf_bar_ld < cols)
        return fail(PM_ERR_ARG, "pm_encircled_energy_adjoint: bad argument");
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_encircled_energy_adjoint: dtype must be PM_C64 (float) or PM_C128 (double)");
    hipStream_t st = PM_STREAM(stream);
    const int64_t total = rows * cols;
    const dim3 grid(unsigned((total + 255) / 256)), block(256);
    const double cell = df * df;
    int acc = 0;
    int64_t r0 = 0;
    do {   // an empty radius list still writes zeros
        const int nr = int(nradii - r0 < kEeMaxRadii ? nradii - r0 : kEeMaxRadii);
        EeRadii rr;
        for (int i = 0; i < kEeMaxRadii; ++i) {
            rr.r[i] = i < nr ? radii_mm[r0 + i] : 0.0;
            rr.w[i] = i < nr ? ee_bar[r0 + i] : 0.0;
        }
        if (dtype == PM_C64)
            hipLaunchKernelGGL(ee_adjoint_kernel<float>, grid, block, 0, st, rows, cols, df, rr, nr, cell, acc, (float*)mtf_bar, mtf_bar_ld);
        else
            hipLaunchKernelGGL(ee_adjoint_kernel<double>, grid, block, 0, st, rows, cols, df, rr, nr, cell, acc, (double*)mtf_bar, mtf_bar_ld);
        acc = 1;
        r0 += kEeMaxRadii;
    } while (r0 < nradii);
    return int(hipGetLastError());
}

}  // extern "C"
