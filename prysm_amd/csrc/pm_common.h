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
