// What keeps the f32 MFMA pipe from its rate inside the complex GEMM loop?  The loop body of cgemm128_kernel rebuilt piece by piece:
//   MODE 0: 12 accumulators (192 registers), operands constant registers
//   MODE 1: + the 3M sums (one v_fma per 3 MFMAs), operands still constant
//   MODE 2: + operands read from LDS (four ds_read_b128 per 24 MFMAs, one quarter ahead)
//   MODE 3: + LDS stores and a barrier per 96 MFMAs
// Prints achieved TFLOP/s of real MFMA work.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", int(e_), __LINE__); return; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, float seed, const float* gsrc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x16 p1[2][2], p2[2][2], p3[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) p1[i][j][r] = p2[i][j][r] = p3[i][j][r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * 36; i += 256) reinterpret_cast<float*>(smem)[i] = seed + (i % 97) * 0.01f;
    __syncthreads();
    const int fa0 = ((wave >> 1) * 64 + (lane & 31)) * 144 + (lane >> 5) * 64, fb0 = 128 * 144 + ((wave & 1) * 64 + (lane & 31)) * 144 + (lane >> 5) * 64;
    f32x4 fa[2][2], fb[2][2];
    const float sa = seed, sb = -seed;
    for (int i = 0; i < 2; ++i) {
        fa[0][i] = fa[1][i] = f32x4{seed + lane, seed - lane, seed * 2, seed * 3 + i};
        fb[0][i] = fb[1][i] = f32x4{seed - lane, seed + lane, seed * 4, seed * 5 + i};
    }
    f32x4 st = f32x4{seed, seed, seed, seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cur = q & 1;
            if (MODE >= 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    fa[cur ^ 1][i] = *reinterpret_cast<const f32x4*>(smem + fa0 + i * 32 * 144 + ((q + 1) & 3) * 16);
                    fb[cur ^ 1][i] = *reinterpret_cast<const f32x4*>(smem + fb0 + i * 32 * 144 + ((q + 1) & 3) * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float ax[2], ay[2], as[2], bx[2], by[2], bs[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ax[i] = fa[cur][i][2 * e];
                    ay[i] = fa[cur][i][2 * e + 1];
                    bx[i] = fb[cur][i][2 * e];
                    by[i] = fb[cur][i][2 * e + 1];
                    if (MODE >= 1) {
                        as[i] = ax[i] + sa * ay[i];
                        bs[i] = bx[i] + sb * by[i];
                    } else {
                        as[i] = ay[i];
                        bs[i] = by[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        p1[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[i], bx[j], p1[i][j], 0, 0, 0);
                        p2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay[i], by[j], p2[i][j], 0, 0, 0);
                        p3[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(as[i], bs[j], p3[i][j], 0, 0, 0);
                    }
                    if (MODE == 10 || (MODE >= 8 && MODE <= 9 && i == 1)) {
                        __builtin_amdgcn_sched_barrier(0);
                        const int slot = (q * 2 + e) * 2 + i;
                        if (MODE == 9) {
                            *reinterpret_cast<float2*>(smem + 2 * 128 * 144 + (tid * 16 + slot) * 16) = float2{st[0], st[1]};
                            *reinterpret_cast<float2*>(smem + 2 * 128 * 144 + (tid * 16 + slot) * 16 + 8) = float2{st[2], st[3]};
                        } else {
                            *reinterpret_cast<f32x4*>(smem + 2 * 128 * 144 + (tid * 16 + slot) * 16) = st;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE >= 3 && q == 1) {
                if (MODE != 4 && MODE < 11) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) *reinterpret_cast<f32x4*>(smem + 2 * 128 * 144 + (tid * 8 + s) * 16) = st;
                }
                if (MODE == 3 || MODE == 4) __syncthreads();
                if (MODE == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (MODE == 7) __builtin_amdgcn_s_barrier();     // bare s_barrier: no waitcnt, no fence
                if (MODE == 11 || MODE == 12) {   // LDS-DMA instead: 8 x global_load_lds_dwordx4 (1 KiB per wave-instruction)
#pragma unroll
                    for (int s = 0; s < 8; ++s)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (size_t(blockIdx.x) * 2048 + s * 256 + tid) * 4),
                                                         (__attribute__((address_space(3))) void*)(smem + 2 * 128 * 144 + (s * 256 + (tid & ~63)) * 16), 16, 0, 0);
                    if (MODE == 12) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += p1[i][j][r] + p2[i][j][r] + p3[i][j][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name) {
    float* out;
    float* gsrc;
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&gsrc, size_t(256) * 2048 * 16));
    CK(hipMemset(gsrc, 0, size_t(256) * 2048 * 16));
    const int iters = 400, grid = 256;
    const size_t lds = 2 * 128 * 144 + 256 * 16 * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(256), lds, 0, out, 10, 1.0f, gsrc);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(256), lds, 0, out, iters, 1.0f, gsrc);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double mfmas = double(grid) * 4 * iters * 96;
    printf("%-70s %.2f ms -> %.1f TFLOP/s real (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, ms, mfmas * 4096 / (ms * 1e-3) / 1e12,
           ms * 1e-3 * 2.4e9 / (mfmas / 1024));
    CK(hipFree(out));
}
int main() {
    run<0>("12 accumulators, constant operands");
    run<1>("+ 3M sums (v_fma)");
    run<2>("+ operands from LDS (ds_read_b128 a quarter ahead)");
    run<3>("+ 8 ds_write_b128 and a barrier per 96 MFMAs");
    run<4>("barrier only per 96 MFMAs (no LDS stores)");
    run<5>("8 ds_write_b128 + s_waitcnt lgkmcnt(0), no barrier");
    run<6>("8 ds_write_b128, no wait, no barrier");
    run<7>("8 ds_write_b128 + bare s_barrier (no waitcnt)");
    run<8>("8 ds_write_b128 SPREAD: one after every 12 MFMAs");
    run<9>("16 ds_write_b64 SPREAD: two after every 12 MFMAs");
    run<10>("16 ds_write_b128 SPREAD: one after every 6 MFMAs");
    run<11>("8 global_load_lds_dwordx4 per 96 MFMAs, never waited");
    run<12>("8 global_load_lds_dwordx4 + vmcnt(0) + s_barrier per 96 MFMAs");
    return 0;
}
