// Shared small types for the prysm_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PM_HD __host__ __device__ __forceinline__
#else
#define PM_HD inline
#endif

// a fence for the instruction scheduler (device code; nothing on the host emulators): keeps unrolled batches apart so that the
// products of sixteen points are not all formed -- and held in registers -- before the first store
#if defined(__HIP_DEVICE_COMPILE__)
#define PM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PM_SCHED_FENCE() ((void)0)
#endif

namespace pm {

template <typename T>
struct cx {
    T x, y;
};

template <typename T> PM_HD cx<T> operator+(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> PM_HD cx<T> operator-(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <typename T> PM_HD cx<T> cmul(cx<T> a, cx<T> b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
template <typename T> PM_HD cx<T> cmulc(cx<T> a, cx<T> b) { return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }  // a*conj(b)
template <typename T> PM_HD cx<T> cconj(cx<T> a) { return {a.x, -a.y}; }
template <typename T> PM_HD cx<T> cscale(cx<T> a, T s) { return {a.x * s, a.y * s}; }
// multiply by -i and +i
template <typename T> PM_HD cx<T> mul_mi(cx<T> a) { return {a.y, -a.x}; }
template <typename T> PM_HD cx<T> mul_pi(cx<T> a) { return {-a.y, a.x}; }

// ---------------------------------------------------------------------------
// PACKED complex64 arithmetic (translation units that define PM_PACKED_F32 before including this header; device code only).  A complex64
// value is a 64-bit register pair from its load to its store, and CDNA3 / CDNA4 have two-lane fp32 instructions on such pairs
// (v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32, each lane choosing either half of either source and its sign): a complex add is ONE
// instruction, a complex multiply TWO, a multiplication by -i rides on the add that consumes it.  The compiler's SLP vectoriser finds
// some of this by itself but pays for it in copies and registers (csrc/Makefile: -fno-slp-vectorize); written out, nothing is copied.
// ---------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && defined(PM_PACKED_F32)
typedef float pm_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pm_v2f pm_pk(cx<float> a) { pm_v2f v; v[0] = a.x; v[1] = a.y; return v; }
__device__ __forceinline__ cx<float> pm_un(pm_v2f v) { return {v[0], v[1]}; }
__device__ __forceinline__ cx<float> operator+(cx<float> a, cx<float> b) { return pm_un(pm_pk(a) + pm_pk(b)); }
__device__ __forceinline__ cx<float> operator-(cx<float> a, cx<float> b) { return pm_un(pm_pk(a) - pm_pk(b)); }
__device__ __forceinline__ cx<float> cscale(cx<float> a, float s) { return pm_un(pm_pk(a) * s); }
// {ax bx - ay by, ax by + ay bx}
__device__ __forceinline__ cx<float> cmul(cx<float> a, cx<float> b) {
    pm_v2f t, r;
    const pm_v2f va = pm_pk(a), vb = pm_pk(b);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(va), "v"(vb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
    return pm_un(r);
}
// a conj(b) = {ax bx + ay by, ay bx - ax by}
__device__ __forceinline__ cx<float> cmulc(cx<float> a, cx<float> b) {
    pm_v2f t, r;
    const pm_v2f va = pm_pk(a), vb = pm_pk(b);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(va), "v"(vb));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(va), "v"(vb), "v"(t));
    return pm_un(r);
}
// ... by a compile-time constant (the rotations inside the small DFTs): the constant pair lives in scalar registers
__device__ __forceinline__ cx<float> cmul_k(cx<float> a, float kr, float ki) {
    pm_v2f t, r, k;
    k[0] = kr;
    k[1] = ki;
    const pm_v2f va = pm_pk(a);
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(va), "s"(k));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(va), "s"(k), "v"(t));
    return pm_un(r);
}
// a + (-i) b = {ax + by, ay - bx} and a - (-i) b = {ax - by, ay + bx}: the -i of a radix-4 butterfly inside its last add
__device__ __forceinline__ cx<float> add_mi(cx<float> a, cx<float> b) {
    pm_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(pm_pk(a)), "v"(pm_pk(b)));
    return pm_un(r);
}
__device__ __forceinline__ cx<float> sub_mi(cx<float> a, cx<float> b) {
    pm_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(pm_pk(a)), "v"(pm_pk(b)));
    return pm_un(r);
}
#endif

// One axis of a windowed, rotated view of an array (see include/prysm_amd.h pm_axis).
//   logical index i in [0,n)  ->  position p = (i + shift) mod n  ->  memory index q = p - off,
//   valid iff 0 <= q < len.
struct AxisMap {
    int n, len, off, shift;
    PM_HD int map(int i) const {  // returns q, or -1 when outside the window
        int p = i + shift;
        if (p >= n) p -= n;
        int q = p - off;
        return (q >= 0 && q < len) ? q : -1;
    }
    // inverse: memory index q -> logical index i
    PM_HD int unmap(int q) const {
        int i = q + off - shift;
        if (i < 0) i += n;
        return i;
    }
};

}  // namespace pm
