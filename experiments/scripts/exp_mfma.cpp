// f32 MFMA issue-rate probe: v_mfma_f32_32x32x2_f32 with NACC accumulators per wave, WPS waves per SIMD, optional barrier every
// 24 MFMAs (the structure of the cgemm K-tile loop).  Prints achieved TFLOP/s (2*32*32*2 flops per MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool BAR>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = seed + threadIdx.x * 1e-3f, b = seed - threadIdx.x * 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 24 / NACC; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + i, b + k, acc[i], 0, 0, 0);
        if (BAR) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int NACC, bool BAR>
static void run(int wgs_per_cu, const char* name) {
    float* out;
    hipMalloc(&out, 4096);
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, BAR>), dim3(grid), dim3(256), 0, 0, out, 10, 1.0f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NACC, BAR>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = double(grid) * 4 * iters * 24;
    printf("%-44s %d WG/CU: %.2f ms -> %.1f TFLOP/s (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, wgs_per_cu, ms,
           mfmas * 4096 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (mfmas / 1024));
    hipFree(out);
}
// short kernels launched back to back, like the cgemm bench: 32 loop iterations of 24 MFMAs per wave, 4 WG/CU
template <int NACC, bool BAR>
static void run_short(int iters, int launches, const char* name) {
    float* out;
    hipMalloc(&out, 4096);
    const int grid = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<NACC, BAR>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e0, 0);
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((probe<NACC, BAR>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / launches, ideal = double(iters) * 24 * 4 * 64 / 2.4e3;
    printf("%-44s %d iterations x %d launches: %.1f us per launch (MFMA pipe time at 2.4 GHz: %.1f us)\n", name, iters, launches, us, ideal);
    hipFree(out);
}
int main() {
    run_short<3, true>(32, 50, "short kernel, 3 acc, barrier");
    run_short<3, true>(64, 50, "short kernel, 3 acc, barrier");
    run_short<3, true>(256, 50, "short kernel, 3 acc, barrier");
    run_short<3, true>(2000, 5, "long kernel, 3 acc, barrier");
    run<4, false>(1, "4 accumulators, no barrier");
    run<3, false>(1, "3 accumulators, no barrier");
    run<3, false>(4, "3 accumulators, no barrier");
    run<3, true>(4, "3 accumulators, barrier every 24 MFMAs");
    run<2, false>(4, "2 accumulators, no barrier");
    run<1, false>(4, "1 accumulator, no barrier");
    return 0;
}
