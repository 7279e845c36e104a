#include <hip/hip_runtime.h>
#include "fft_mixed.h"
namespace pm {
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f p_mi(v2f a) { return v2f{a.y, -a.x}; }            // * -i
__device__ __forceinline__ v2f p_swap(v2f a) { return __builtin_shufflevector(a, a, 1, 0); }
// a * (c - i s)... general constant w = (wr, wi): a*w = a*wr + swap(a)*(-wi, wi)
__device__ __forceinline__ v2f p_cmulc(v2f a, float wr, float wi) { return a * wr + p_swap(a) * v2f{-wi, wi}; }

template <int R> __device__ __forceinline__ void p_dft_odd(v2f* a) {
    constexpr int H = (R - 1) / 2;
    v2f p[H + 1], q[H + 1];
    v2f sum = a[0];
#pragma unroll
    for (int m = 1; m <= H; ++m) { p[m] = a[m] + a[R - m]; q[m] = a[m] - a[R - m]; sum += p[m]; }
    const v2f x0 = a[0];
    a[0] = sum;
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        v2f A = x0, B = v2f{0.f, 0.f};
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            const float c = float(MixRoots<R>::tab.c[(m * k) % R]), s = float(MixRoots<R>::tab.s[(m * k) % R]);
            A += p[m] * c;
            B += q[m] * s;
        }
        const v2f Bi = p_mi(B);      // -i B = (B.y, -B.x)
        a[k] = A + Bi;
        a[R - k] = A - Bi;
    }
}
__device__ __forceinline__ void p_dft2(v2f& a, v2f& b) { v2f t = a; a = t + b; b = t - b; }
__device__ __forceinline__ void p_dft10(v2f* a) {   // CT<2,5>: n = 5 n1 + n2
    v2f o[10];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        v2f t0 = a[n2], t1 = a[5 + n2];
        p_dft2(t0, t1);
        a[n2] = t0;            // k1 = 0
        if (n2) t1 = p_cmulc(t1, float(MixRoots<10>::tab.c[n2]), float(-MixRoots<10>::tab.s[n2]));
        a[5 + n2] = t1;        // k1 = 1
    }
    p_dft_odd<5>(a);
    p_dft_odd<5>(a + 5);
#pragma unroll
    for (int k2 = 0; k2 < 5; ++k2) { o[2 * k2] = a[k2]; o[1 + 2 * k2] = a[5 + k2]; }
#pragma unroll
    for (int k = 0; k < 10; ++k) a[k] = o[k];
}
__global__ void pk10(const v2f* in, v2f* out) {
    v2f a[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) a[k] = in[threadIdx.x + k * 64];
    p_dft10(a);
#pragma unroll
    for (int k = 0; k < 10; ++k) out[threadIdx.x + k * 64] = a[k];
}
__global__ void pk5(const v2f* in, v2f* out) {
    v2f a[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) a[k] = in[threadIdx.x + k * 64];
    p_dft_odd<5>(a);
#pragma unroll
    for (int k = 0; k < 5; ++k) out[threadIdx.x + k * 64] = a[k];
}
template <typename T, int R>
__global__ void sc(const cx<T>* in, cx<T>* out) {
    cx<T> a[R];
#pragma unroll
    for (int k = 0; k < R; ++k) a[k] = in[threadIdx.x + k * 64];
    MixDft<T, R>::run(a);
#pragma unroll
    for (int k = 0; k < R; ++k) out[threadIdx.x + k * 64] = a[k];
}
template __global__ void sc<float, 10>(const cx<float>*, cx<float>*);
template __global__ void sc<float, 5>(const cx<float>*, cx<float>*);
}
