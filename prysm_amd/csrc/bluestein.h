// Bluestein (chirp-z) re-expression of a length-n DFT, n NOT a power of two, as a cyclic convolution of power-of-two
// length MB >= 2n - 1 that the Stockham engine runs:
//
//     X[k] = sum_j x[j] exp(-2 pi i jk/n),     jk = (j^2 + k^2 - (k-j)^2) / 2
//          = w[k] * sum_j (x[j] w[j]) conj(w)[k-j],                    w[j] = exp(-i pi j^2 / n)
//
//     a = x .* w, zero padded to MB;  A = FFT_MB(a);  c = IFFT_MB(A .* B);  X[k] = w[k] c[k], k < n
//     B = FFT_MB(b) / MB,  b[m] = conj(w[|m|]) for m in (-n, n) wrapped mod MB    (the 1/MB of the inverse lives in B)
//
// The tables (w, B) are built once per (n, precision) on the host in long double -- the chirp phase j^2 mod 2n is exact
// integer arithmetic -- and rounded once.  The reference reaches arbitrary lengths through scipy.fft / pocketfft
// (prysm/propagation/fft.py:24, angular_spectrum.py:35-42, fttools.py:301-321); before this path such lengths ran on
// the O(n^2) direct kernel (dft_direct.hip), which stays for short lengths and for lengths above 4096.
//
// Everything here is __host__ __device__ / plain C++ so that tools/emu_fft.cpp can run it on the CPU.
#pragma once
#include <cmath>
#include <complex>
#include <vector>

#include "pm_common.h"

namespace pm {

constexpr int kBlueMaxN = 4096;   // 2n - 1 <= 8192 = the engine's longest transform

inline int blue_conv_len(int64_t n) {  // power of two >= 2n - 1 (>= 2: the engine's shortest transform)
    int64_t m = 2;
    while (m < 2 * n - 1) m <<= 1;
    return int(m);
}

// tab[0 .. n) = w,  tab[n .. n + mb) = B
template <typename T>
void blue_make_tables(int n, int mb, std::vector<cx<T>>& tab) {
    typedef long double ld;
    typedef std::complex<ld> cld;
    const ld pi = acosl(-1.0L);
    const size_t un = size_t(n), umb = size_t(mb);
    std::vector<cld> w(un), b(umb, cld(0, 0));
    for (int j = 0; j < n; ++j) {
        const int64_t r = (int64_t(j) * j) % (2 * int64_t(n));
        const ld a = -pi * ld(r) / ld(n);
        w[size_t(j)] = cld(cosl(a), sinl(a));
    }
    b[0] = std::conj(w[0]);
    for (int j = 1; j < n; ++j) b[size_t(j)] = b[size_t(mb - j)] = std::conj(w[size_t(j)]);
    // forward FFT of b, length mb (radix-2 decimation in time, long double)
    for (int i = 1, j = 0; i < mb; ++i) {
        int bit = mb >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(b[size_t(i)], b[size_t(j)]);
    }
    for (int len = 2; len <= mb; len <<= 1) {
        const size_t half = size_t(len / 2);
        std::vector<cld> tw(half);
        for (int k = 0; k < len / 2; ++k) {
            const ld a = -2.0L * pi * ld(k) / ld(len);
            tw[size_t(k)] = cld(cosl(a), sinl(a));
        }
        for (int i = 0; i < mb; i += len)
            for (int k = 0; k < len / 2; ++k) {
                const cld u = b[size_t(i + k)], v = b[size_t(i + k + len / 2)] * tw[size_t(k)];
                b[size_t(i + k)] = u + v;
                b[size_t(i + k + len / 2)] = u - v;
            }
    }
    tab.resize(size_t(n) + size_t(mb));
    for (int j = 0; j < n; ++j) tab[size_t(j)] = {T(w[size_t(j)].real()), T(w[size_t(j)].imag())};
    for (int k = 0; k < mb; ++k) {
        const cld v = b[size_t(k)] / ld(mb);
        tab[size_t(n) + size_t(k)] = {T(v.real()), T(v.imag())};
    }
}

// One sequence element of a windowed / rotated, possibly real, possibly conjugated input (the DirectIn view of
// pm_internal.h, restated on plain fields so that the emulator can call it): logical index i of sequence `seq`.
template <typename T>
struct BlueIn {
    const void* src;   // cx<T>* or T* (real)
    int64_t s_seq, s_i;
    AxisMap ax;
    int conj, real;
};

template <typename T>
PM_HD cx<T> blue_fetch(const BlueIn<T>& in, int seq, int i) {
    const int q = in.ax.map(i);
    if (q < 0) return cx<T>{T(0), T(0)};
    const int64_t at = int64_t(seq) * in.s_seq + int64_t(q) * in.s_i;
    cx<T> x;
    if (in.real)
        x = {reinterpret_cast<const T*>(in.src)[at], T(0)};
    else
        x = reinterpret_cast<const cx<T>*>(in.src)[at];
    if (in.conj) x.y = -x.y;
    return x;
}

}  // namespace pm
