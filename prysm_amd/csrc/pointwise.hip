// Bandwidth-bound pointwise / synthesis kernels of the propagation path (gfx950).
//
// All of them are 2-D row-major with a leading dimension; one thread handles VEC adjacent
// elements of a row (16 B per lane where the shape allows), grid-stride over rows so the launch
// has >> 256 workgroups without exceeding ~2048.  Phases are reduced in fp64 before the sincos so
// the fp32 path keeps full fp32 accuracy for arguments of many thousands of radians
// (quadratic phases, OPD / lambda).
#include "pm_internal.h"

namespace pm {

static inline dim3 grid2d(int64_t rows, int64_t cols_per_thread_units, dim3& block) {
    block = dim3(64, 4);
    int64_t gx = (cols_per_thread_units + block.x - 1) / block.x;
    int64_t gy = (rows + block.y - 1) / block.y;
    if (gy > 65535) gy = 65535;
    return dim3((unsigned)gx, (unsigned)gy);
}

__device__ __forceinline__ void sincos_turns(double turns, double* s, double* c) {
    // exp(2 pi i turns): reduce to [-0.5, 0.5) turns exactly, then sincospi
    const double r = turns - rint(turns);
    sincospi(2.0 * r, s, c);
}

// ---------------------------------------------------------------- cmul
template <typename T, int OP>
__global__ void cmul_kernel(int64_t rows, int64_t cols, const cx<T>* a, int64_t lda, const cx<T>* b, int64_t ldb,
                            cx<T>* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        const cx<T> x = a[r * lda + c], y = b[r * ldb + c];
        o[r * ldo + c] = OP == 0 ? cmul(x, y) : cmulc(x, y);
    }
}

template <typename T>
int cmul_launch(int op, int64_t rows, int64_t cols, const void* a, int64_t lda, const void* b, int64_t ldb, void* o,
                int64_t ldo, hipStream_t st) {
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    if (op == 0)
        hipLaunchKernelGGL((cmul_kernel<T, 0>), grid, block, 0, st, rows, cols, (const cx<T>*)a, lda, (const cx<T>*)b, ldb, (cx<T>*)o, ldo);
    else
        hipLaunchKernelGGL((cmul_kernel<T, 1>), grid, block, 0, st, rows, cols, (const cx<T>*)a, lda, (const cx<T>*)b, ldb, (cx<T>*)o, ldo);
    return int(hipGetLastError());
}

// ---------------------------------------------------------------- real x complex
template <typename T>
__global__ void rmul_kernel(int64_t rows, int64_t cols, const T* r_, int64_t ldr, const cx<T>* a, int64_t lda, T scale,
                            cx<T>* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y)
        o[r * ldo + c] = cscale(a[r * lda + c], scale * r_[r * ldr + c]);
}

// ---------------------------------------------------------------- modulus and phase of one complex array in one sweep
template <typename T>
__global__ void abs_arg_kernel(int64_t rows, int64_t cols, const cx<T>* in, int64_t ldi, T* oabs, int64_t lda, T* oarg, int64_t ldg) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        const cx<T> z = in[r * ldi + c];
        if (oabs) oabs[r * lda + c] = sqrt(z.x * z.x + z.y * z.y);
        if (oarg) oarg[r * ldg + c] = atan2(z.y, z.x);
    }
}

// ---------------------------------------------------------------- separable scale
template <typename T>
__global__ void scale_sep_kernel(int64_t rows, int64_t cols, const cx<T>* in, int64_t ldi, const cx<T>* ry, int ryc,
                                 const cx<T>* cxv, int cxc, T scale, cx<T>* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    cx<T> fc = {scale, T(0)};
    if (cxv) {
        cx<T> w = cxv[c];
        if (cxc) w.y = -w.y;
        fc = cscale(w, scale);
    }
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        cx<T> f = fc;
        if (ry) {
            cx<T> w = ry[r];
            if (ryc) w.y = -w.y;
            f = cmul(f, w);
        }
        o[r * ldo + c] = cmul(in[r * ldi + c], f);
    }
}

// ---------------------------------------------------------------- |.|^2
template <typename T, int ACC>
__global__ void abs2_kernel(int64_t rows, int64_t cols, const cx<T>* in, int64_t ldi, T* o, int64_t ldo, T weight) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        const cx<T> x = in[r * ldi + c];
        const T i2 = x.x * x.x + x.y * x.y;
        if (ACC)
            o[r * ldo + c] += weight * i2;
        else
            o[r * ldo + c] = i2;
    }
}

// ---------------------------------------------------------------- weighted sum of modes
// out (+)= sum_b w[b] * modes[b]; the weights ride in the kernel arguments (SGPRs), 32 modes per launch
template <typename T>
struct ModeWeights {
    T w[32];
};
template <typename T>
__global__ void sum_modes_kernel(int64_t rows, int64_t cols, const T* __restrict__ modes, int64_t mstride, int64_t ldm,
                                 const ModeWeights<T> mw, int nb, int acc, T* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        const T* m = modes + r * ldm + c;
        T a = acc ? o[r * ldo + c] : T(0);
        for (int b = 0; b < nb; ++b) a += mw.w[b] * m[int64_t(b) * mstride];
        o[r * ldo + c] = a;
    }
}

// ---------------------------------------------------------------- measured focal-plane mask
// map_coordinates(order 0 | 1, mode='nearest') of a complex map at (row, col) = ((yf-cy)/dx + ny/2, (xf-cx)/dx + nx/2);
// coordinates are formed in the precision of xf / yf as numpy does, the interpolation itself in fp64 as scipy does.
template <typename T>
__global__ void sample_map_kernel(int order, int64_t ny, int64_t nx, const cx<T>* __restrict__ map, int64_t ldm, T dx, T cxo, T cyo,
                                  int64_t rows, int64_t cols, const T* xf, int64_t xsy, int64_t xsx, const T* yf, int64_t ysy,
                                  int64_t ysx, const cx<T>* fill, int64_t ldf, cx<T> fillv, cx<T>* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        const T col = (xf[r * xsy + c * xsx] - cxo) / dx + T(nx / 2);
        const T row = (yf[r * ysy + c * ysx] - cyo) / dx + T(ny / 2);
        const bool inside = row >= T(0) && row <= T(ny - 1) && col >= T(0) && col <= T(nx - 1);
        cx<T> v;
        if (!inside) {
            v = fill ? fill[r * ldf + c] : fillv;
        } else {
            const double rr = double(row), cc = double(col);
            if (order == 0) {
                int64_t ir = int64_t(floor(rr + 0.5)), ic = int64_t(floor(cc + 0.5));
                ir = ir > ny - 1 ? ny - 1 : ir;
                ic = ic > nx - 1 ? nx - 1 : ic;
                v = map[ir * ldm + ic];
            } else {
                const int64_t r0 = int64_t(floor(rr)), c0 = int64_t(floor(cc));
                const int64_t r1 = r0 + 1 > ny - 1 ? ny - 1 : r0 + 1, c1 = c0 + 1 > nx - 1 ? nx - 1 : c0 + 1;
                const double tr = rr - double(r0), tc = cc - double(c0);
                const cx<T> a = map[r0 * ldm + c0], b = map[r0 * ldm + c1], d = map[r1 * ldm + c0], e = map[r1 * ldm + c1];
                const double w00 = (1 - tr) * (1 - tc), w01 = (1 - tr) * tc, w10 = tr * (1 - tc), w11 = tr * tc;
                v = {T(w00 * a.x + w01 * b.x + w10 * d.x + w11 * e.x), T(w00 * a.y + w01 * b.y + w10 * d.y + w11 * e.y)};
            }
        }
        o[r * ldo + c] = v;
    }
}

// ---------------------------------------------------------------- spline orders 2..5 of map_coordinates(mode='nearest')
// scipy pads the map by 12 samples of edge values, runs the recursive B-spline prefilter (poles z_k, reflect initialisation, gain
// prod (1 - z)(1 - 1/z)) along each axis in fp64, and evaluates the tensor-product B-spline at coordinate + 12
// (scipy/ndimage/_interpolation.py map_coordinates / _prepad_for_spline_filter, src/ni_splines.c).  Restated here: one thread
// filters one line of the padded coefficient array in place.
struct SplinePoles {
    int n;
    double z[2];
};
static SplinePoles spline_poles(int order) {
    switch (order) {
        case 2: return {1, {sqrt(8.0) - 3.0, 0.0}};
        case 3: return {1, {sqrt(3.0) - 2.0, 0.0}};
        case 4: return {2, {sqrt(664.0 - sqrt(438976.0)) + sqrt(304.0) - 19.0, sqrt(664.0 + sqrt(438976.0)) - sqrt(304.0) - 19.0}};
        default: return {2, {sqrt(67.5 - sqrt(4436.25)) + sqrt(26.25) - 6.5, sqrt(67.5 + sqrt(4436.25)) - sqrt(26.25) - 6.5}};
    }
}
constexpr int kSplinePad = 12;

__device__ __forceinline__ void spline_filter_line(cx<double>* c, int64_t stride, int64_t n, const SplinePoles& pl) {
    for (int k = 0; k < pl.n; ++k) {
        const double z = pl.z[k];
        const double z_n = pow(z, double(n));
        // causal initialisation, reflect boundary
        const cx<double> c0 = c[0], cl = c[(n - 1) * stride];
        double ar = c0.x + z_n * cl.x, ai = c0.y + z_n * cl.y, z_i = z;
        for (int64_t i = 1; i < n; ++i) {
            const cx<double> a = c[i * stride], b = c[(n - 1 - i) * stride];
            ar += z_i * (a.x + z_n * b.x);
            ai += z_i * (a.y + z_n * b.y);
            z_i *= z;
        }
        const double f = z / (1.0 - z_n * z_n);
        cx<double> prev = {ar * f + c0.x, ai * f + c0.y};
        c[0] = prev;
        for (int64_t i = 1; i < n; ++i) {
            cx<double> v = c[i * stride];
            v.x += z * prev.x;
            v.y += z * prev.y;
            c[i * stride] = v;
            prev = v;
        }
        const double g = z / (z - 1.0);
        prev = {prev.x * g, prev.y * g};
        c[(n - 1) * stride] = prev;
        for (int64_t i = n - 2; i >= 0; --i) {
            const cx<double> v = c[i * stride];
            prev = {z * (prev.x - v.x), z * (prev.y - v.y)};
            c[i * stride] = prev;
        }
    }
}

// coeff[r][c] = gain^2 * map[clamp(r - 12)][clamp(c - 12)]   (edge padding; the gain of both axes applied up front)
template <typename T>
__global__ void spline_pad_kernel(int64_t ny, int64_t nx, const cx<T>* map, int64_t ldm, double gain2, cx<double>* coeff, int64_t ldc) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t py = ny + 2 * kSplinePad, px = nx + 2 * kSplinePad;
    if (g >= py * px) return;
    const int64_t r = g / px, c = g - r * px;
    int64_t sr = r - kSplinePad, sc = c - kSplinePad;
    sr = sr < 0 ? 0 : (sr > ny - 1 ? ny - 1 : sr);
    sc = sc < 0 ? 0 : (sc > nx - 1 ? nx - 1 : sc);
    const cx<T> v = map[sr * ldm + sc];
    coeff[r * ldc + c] = {double(v.x) * gain2, double(v.y) * gain2};
}
// axis = 0: one thread per column (line stride ldc); axis = 1: one thread per row (line stride 1)
__global__ void spline_filter_kernel(int axis, int64_t py, int64_t px, cx<double>* coeff, int64_t ldc, SplinePoles pl) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (axis == 0) {
        if (g < px) spline_filter_line(coeff + g, ldc, py, pl);
    } else {
        if (g < py) spline_filter_line(coeff + g * ldc, 1, px, pl);
    }
}

// centred cardinal B-spline of degree n at t: (1/n!) sum_j (-1)^j C(n+1, j) (t + (n+1)/2 - j)_+^n
__device__ __forceinline__ double bspline_value(double t, int n) {
    const double binom[6][7] = {{1, 1, 0, 0, 0, 0, 0}, {1, 2, 1, 0, 0, 0, 0}, {1, 3, 3, 1, 0, 0, 0}, {1, 4, 6, 4, 1, 0, 0},
                                {1, 5, 10, 10, 5, 1, 0}, {1, 6, 15, 20, 15, 6, 1}};
    const double fact[6] = {1, 1, 2, 6, 24, 120};
    double s = 0.0, sign = 1.0;
    for (int j = 0; j <= n + 1; ++j) {
        const double u = t + 0.5 * double(n + 1) - double(j);
        if (u > 0.0) {
            double p = 1.0;
            for (int e = 0; e < n; ++e) p *= u;
            s += sign * binom[n][j] * p;
        }
        sign = -sign;
    }
    return s / fact[n];
}

template <typename T>
__global__ void sample_spline_kernel(int order, int64_t ny, int64_t nx, const cx<double>* __restrict__ coeff, int64_t ldc, T dx, T cxo, T cyo,
                                     int64_t rows, int64_t cols, const T* xf, int64_t xsy, int64_t xsx, const T* yf, int64_t ysy,
                                     int64_t ysx, const cx<T>* fill, int64_t ldf, cx<T> fillv, cx<T>* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        const T col = (xf[r * xsy + c * xsx] - cxo) / dx + T(nx / 2);
        const T row = (yf[r * ysy + c * ysx] - cyo) / dx + T(ny / 2);
        const bool inside = row >= T(0) && row <= T(ny - 1) && col >= T(0) && col <= T(nx - 1);
        cx<T> v;
        if (!inside) {
            v = fill ? fill[r * ldf + c] : fillv;
        } else {
            const double rr = double(row) + kSplinePad, cc = double(col) + kSplinePad;
            const int64_t r0 = int64_t(floor((order & 1) ? rr : rr + 0.5)) - order / 2;
            const int64_t c0 = int64_t(floor((order & 1) ? cc : cc + 0.5)) - order / 2;
            double wc[6];
            for (int b = 0; b <= order; ++b) wc[b] = bspline_value(cc - double(c0 + b), order);
            double ar = 0.0, ai = 0.0;
            for (int a = 0; a <= order; ++a) {
                const double wa = bspline_value(rr - double(r0 + a), order);
                const cx<double>* line = coeff + (r0 + a) * ldc + c0;
                double lr = 0.0, li = 0.0;
                for (int b = 0; b <= order; ++b) {
                    lr += wc[b] * line[b].x;
                    li += wc[b] * line[b].y;
                }
                ar += wa * lr;
                ai += wa * li;
            }
            v = {T(ar), T(ai)};
        }
        o[r * ldo + c] = v;
    }
}

// ---------------------------------------------------------------- pupil synthesis
template <typename T, typename A>
__global__ void pupil_kernel(int64_t rows, int64_t cols, const A* amp, int64_t lda, const T* opd, int64_t ldp,
                             double k_over_2pi, cx<T>* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        double s, co;
        sincos_turns(double(opd[r * ldp + c]) * k_over_2pi, &s, &co);
        const double a = amp ? double(amp[r * lda + c]) : 1.0;
        o[r * ldo + c] = {T(a * co), T(a * s)};
    }
}

template <typename T>
__global__ void quad_phase_kernel(int64_t rows, int64_t cols, const T* x, int64_t ldx, const T* y, int64_t ldy,
                                  double c_over_2pi, cx<T>* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y) {
        // rsq = x*x + y*y (wavefront.py:139-140), formed in fp64 from the stored coordinates
        const double xv = double(x[r * ldx + c]), yv = double(y[r * ldy + c]);
        double s, co;
        sincos_turns((xv * xv + yv * yv) * c_over_2pi, &s, &co);
        o[r * ldo + c] = {T(co), T(s)};
    }
}

// ---------------------------------------------------------------- transfer function factors
// both factors in ONE launch (round 4; two launches cost 9.8 us of config 3's 320): blocks [0, ceil(rows / 256)) fill hy, the rest hx
template <typename T>
__global__ void as_tf_vec_kernel(int64_t rows, int64_t cols, double d, double coef_over_2pi, cx<T>* hy, cx<T>* hx) {
    const int64_t by = (rows + blockDim.x - 1) / blockDim.x;
    const bool isx = int64_t(blockIdx.x) >= by;
    const int64_t n = isx ? cols : rows;
    cx<T>* h = isx ? hx : hy;
    const int64_t i = (int64_t(blockIdx.x) - (isx ? by : 0)) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // fftfreq(n, d): [0, 1, ..., (n-1)/2, -(n/2), ..., -1] / (d n), then cast to the real dtype
    const int64_t half = (n - 1) / 2;
    const int64_t ii = i <= half ? i : i - n;
    const T k = T(double(ii) / (double(n) * d));   // .astype(config.precision), angular_spectrum.py:107
    const double kd = double(k);
    double s, co;
    sincos_turns(kd * kd * coef_over_2pi, &s, &co);
    h[i] = {T(co), T(s)};
}

template <typename T>
__global__ void outer_kernel(int64_t rows, int64_t cols, const cx<T>* hy, const cx<T>* hx, cx<T>* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    const cx<T> wx = hx[c];
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < rows; r += int64_t(gridDim.y) * blockDim.y)
        o[r * ldo + c] = cmul(hy[r], wx);
}

// ---------------------------------------------------------------- embed (pad / crop)
template <typename V>
__global__ void embed_kernel(int64_t irows, int64_t icols, const V* in, int64_t ldi, int64_t orows, int64_t ocols,
                             int64_t offy, int64_t offx, V fill, V* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= ocols) return;
    const int64_t ic = c - offx;
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < orows; r += int64_t(gridDim.y) * blockDim.y) {
        const int64_t ir = r - offy;
        V val = fill;
        if (ir >= 0 && ir < irows && ic >= 0 && ic < icols) val = in[ir * ldi + ic];
        o[r * ldo + c] = val;
    }
}

// np.pad's index-mapping modes (fttools.pad2d(mode=...), prysm/fttools.py:96-98): output index r (relative to the first input
// sample) reads input index map(r); 1 edge, 2 reflect (period 2n - 2, the edge sample not repeated), 3 symmetric (period 2n), 4 wrap
__device__ __forceinline__ int64_t pad_map(int64_t r, int64_t n, int mode) {
    if (r >= 0 && r < n) return r;
    if (n == 1) return 0;
    if (mode == 1) return r < 0 ? 0 : n - 1;
    if (mode == 4) {
        r %= n;
        return r < 0 ? r + n : r;
    }
    const int64_t period = mode == 2 ? 2 * n - 2 : 2 * n;
    r %= period;
    if (r < 0) r += period;
    if (r < n) return r;
    return mode == 2 ? period - r : period - 1 - r;
}
template <typename V>
__global__ void pad_index_kernel(int mode, int64_t irows, int64_t icols, const V* in, int64_t ldi, int64_t orows, int64_t ocols,
                                 int64_t offy, int64_t offx, V* o, int64_t ldo) {
    const int64_t c = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (c >= ocols) return;
    const int64_t ic = pad_map(c - offx, icols, mode);
    for (int64_t r = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; r < orows; r += int64_t(gridDim.y) * blockDim.y)
        o[r * ldo + c] = in[pad_map(r - offy, irows, mode) * ldi + ic];
}

// ---------------------------------------------------------------- MDFT basis
template <typename T>
__global__ void mdft_basis_kernel(int64_t M, int64_t N, const T* f, const T* x, double sign, cx<T>* E, int64_t ldE) {
    const int64_t n = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double xv = double(x[n]);
    for (int64_t m = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; m < M; m += int64_t(gridDim.y) * blockDim.y) {
        // exp(sign 2 pi i outer(f, x)) (fttools.py:189-191); the product and its reduction to one turn
        // are done in fp64 so the fp32 basis carries a single rounding
        double s, co;
        sincos_turns(sign * double(f[m]) * xv, &s, &co);
        E[m * ldE + n] = {T(co), T(s)};
    }
}

// ---------------------------------------------------------------- chirp-Z vectors
// The three chirps of one CZT axis (prysm/fttools.py:364-389 _prepare_czt_basis) from its scalars, one launch: input chirp
// b[j] = e(half n^2), n = j - N/2; output chirp a[i] = e(half q^2), q = i - M/2 + shift; convolution kernel h[t] = e(-half (d + shift)^2),
// d = t - M/2 - (N - 1 - N/2) for t < N + M - 1 and 0 up to K; e(t) = exp(2 pi i t) with the turns reduced in fp64.
template <typename T>
__global__ void czt_vectors_kernel(int64_t N, int64_t M, int64_t K, double shift, double half, cx<T>* b, cx<T>* a, cx<T>* h) {
    const int64_t g = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    double s, c;
    if (g < N) {
        const double n = double(g - N / 2);
        sincos_turns(half * n * n, &s, &c);
        b[g] = {T(c), T(s)};
    } else if (g < N + M) {
        const int64_t i = g - N;
        const double q = double(i - M / 2) + shift;
        sincos_turns(half * q * q, &s, &c);
        a[i] = {T(c), T(s)};
    } else if (g < N + M + K) {
        const int64_t t = g - N - M;
        cx<T> v = {T(0), T(0)};
        if (t < N + M - 1) {
            const double d = double(t - M / 2 - (N - 1 - N / 2)) + shift;
            sincos_turns(-half * d * d, &s, &c);
            v = {T(c), T(s)};
        }
        h[t] = v;
    }
}

// rounded-once products / sums in T: hipcc contracts a * b + c into an fma by default (and HIP's __fmul_rn is a plain product that
// contracts just the same after inlining), but the grids must equal the vectors torch / numpy build operation by operation
template <typename T> __device__ __forceinline__ T mul_rn(T a, T b) {
#pragma clang fp contract(off)
    return a * b;
}
template <typename T> __device__ __forceinline__ T add_rn(T a, T b) {
#pragma clang fp contract(off)
    return a + b;
}

// the same basis with both vectors given as FFT-centred grids, built in the kernel exactly as coordinates_for_focus builds them:
//   x[n] = (n - N/2) * x_step,   f[m] = ((m - M/2) * f_step + f_shift) * f_scale,   every operation rounded in T
template <typename T>
__global__ void mdft_basis_grid_kernel(int64_t M, int64_t N, T f_step, T f_shift, T f_scale, T x_step, double sign, cx<T>* E, int64_t ldE) {
    const int64_t n = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double xv = double(mul_rn(T(n - N / 2), x_step));
    for (int64_t m = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; m < M; m += int64_t(gridDim.y) * blockDim.y) {
        const T fv = mul_rn(add_rn(mul_rn(T(m - M / 2), f_step), f_shift), f_scale);
        double s, co;
        sincos_turns(sign * double(fv) * xv, &s, &co);
        E[m * ldE + n] = {T(co), T(s)};
    }
}

}  // namespace pm

using namespace pm;

#define PM_STREAM(s) reinterpret_cast<hipStream_t>(s)

// ---------------------------------------------------------------- real-input 2-D spectrum from the transform of the packed array
// A real M x N array (N even) read as M x N/2 complex numbers z[r][j] = x[r][2j] + i x[r][2j+1] costs nothing to form -- it is the same
// memory -- and its 2-D transform Zf (any route of pm_fft2: engine, mixed-radix, Bluestein) is half the work of transforming x as
// complex.  One sweep untangles it:
//     A = Zf[u][k],  B = conj Zf[(M - u) % M][(N/2 - k) % (N/2)]
//     F[u][k] = (A + B) / 2 + W_N^k (A - B) / (2i),        k = 0 .. N/2 - 1;      F[u][N/2] = (A0 + B0) / 2 - (A0 - B0) / (2i) at k = 0
// and F[(M - u) % M][N - k] = conj F[u][k] fills the other half.  Rotations of the INPUT by (sy, sx) samples (ifftshift) are phases
// here -- conj(W_M^(u sy) W_N^(k sx)), signs for half a length -- because the packed view cannot rotate by an odd number of samples;
// rotations of the OUTPUT are index maps.  Optional division by F[0][0] = Re Zf[0][0] + Im Zf[0][0] (the sum of all samples: real)
// and the |.|, |.|^2, angle epilogues of the Hermitian path: otf.mtf_from_psf / ptf_from_psf / otf_from_psf on ANY even width
// (prysm/otf.py:28-33, 62-135 take any size through scipy) in one half-size transform + this sweep.
template <typename T, int EPI>
__device__ __forceinline__ void untangle_put(void* out, int64_t ldo, int64_t qy, int64_t qx, cx<T> f) {
    const int64_t at = qy * ldo + qx;
    if constexpr (EPI == PM_EPI_NONE) reinterpret_cast<cx<T>*>(out)[at] = f;
    else if constexpr (EPI == PM_EPI_ABS) reinterpret_cast<T*>(out)[at] = sqrt(f.x * f.x + f.y * f.y);
    else if constexpr (EPI == PM_EPI_ABS2) reinterpret_cast<T*>(out)[at] = f.x * f.x + f.y * f.y;
    else reinterpret_cast<T*>(out)[at] = atan2(f.y, f.x);
}

// One thread takes the PAIR of packed bins a = Zf[u][k], b = Zf[(M - u) % M][(N/2 - k) % (N/2)] for k = 0 .. N/4: every element of Zf is
// read once (the first form read each twice: 86 MB for a 36 MB array, profiles/r05), and the pair yields two bins of F --
// F[u][k] from (A, B) = (a, conj b) with W_N^k, F[(M - u) % M][N/2 - k] from (b, conj a) with W_N^(N/2 - k) = -conj W_N^k -- each stored with
// its conjugate mirror image: four coalesced stores per thread.  Column k = 0 also yields the Nyquist column N/2.
template <typename T, int EPI>
struct UntangleOut {
    int64_t M, N, isy, isx, osy, osx, ldo;
    const cx<T>* twn;
    const cx<T>* twm;
    T s;
    void* out;
    // F[u][k] (0 <= k <= N/2) through the input-rotation phase, the scale, the epilogue and the output rotation, with its mirror image
    __device__ __forceinline__ void emit(int64_t u, int64_t k, cx<T> f) const {
        cx<T> ph = {T(1), T(0)};
        if (2 * isx == N) {                          // half a length (ifftshift of an even axis): a sign
            if (k & 1) ph.x = T(-1);
        } else if (isx) {
            const cx<T> t = twn[(k * isx) % N];
            ph = {t.x, -t.y};
        }
        if (2 * isy == M) {
            if (u & 1) ph = {-ph.x, -ph.y};
        } else if (isy) {
            const cx<T> t = twm[(u * isy) % M];
            ph = cmul(ph, cx<T>{t.x, -t.y});
        }
        f = cscale(cmul(f, ph), s);
        int64_t qy = u + osy, qx = k + osx;
        if (qy >= M) qy -= M;
        if (qx >= N) qx -= N;
        untangle_put<T, EPI>(out, ldo, qy, qx, f);
        if (k != 0 && 2 * k != N) {
            int64_t qym = (u == 0 ? 0 : M - u) + osy, qxm = N - k + osx;
            if (qym >= M) qym -= M;
            if (qxm >= N) qxm -= N;
            untangle_put<T, EPI>(out, ldo, qym, qxm, cx<T>{f.x, -f.y});
        }
    }
};

template <typename T>
__device__ __forceinline__ cx<T> untangle_bin(cx<T> a, cx<T> bconj, cx<T> w) {      // (A + B) / 2 + W (A - B) / (2i)
    const cx<T> xe = {T(0.5) * (a.x + bconj.x), T(0.5) * (a.y + bconj.y)};
    const cx<T> d = {T(0.5) * (a.x - bconj.x), T(0.5) * (a.y - bconj.y)};
    return xe + cmul(w, mul_mi(d));
}

template <typename T, int EPI>
__global__ void r2c_untangle_kernel(int64_t M, int64_t N, const cx<T>* __restrict__ zf, int64_t ldz, const cx<T>* __restrict__ twn,
                                    const cx<T>* __restrict__ twm, int64_t isy, int64_t isx, int64_t osy, int64_t osx, int norm_dc, T scale,
                                    void* out, int64_t ldo) {
    const int64_t n2 = N / 2;
    const int64_t k = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;      // 0 .. N/4 inclusive
    if (k > n2 / 2) return;
    T s = scale;
    if (norm_dc) {
        const cx<T> z0 = zf[0];
        s = scale / (z0.x + z0.y);
    }
    const UntangleOut<T, EPI> o{M, N, isy, isx, osy, osx, ldo, twn, twm, s, out};
    const int64_t kb = k == 0 ? 0 : n2 - k;
    const cx<T> w = twn[k];                                                 // W_N^k; W_N^kb = -conj(W_N^k)
    for (int64_t u = int64_t(blockIdx.y) * blockDim.y + threadIdx.y; u < M; u += int64_t(gridDim.y) * blockDim.y) {
        const int64_t um = u == 0 ? 0 : M - u;
        const cx<T> a = zf[u * ldz + k];
        const cx<T> b = zf[um * ldz + kb];
        const cx<T> ac = {a.x, -a.y}, bc = {b.x, -b.y};
        o.emit(u, k, untangle_bin(a, bc, w));
        if (k == 0) o.emit(u, n2, untangle_bin(a, bc, cx<T>{T(-1), T(0)}));           // the Nyquist column: W_N^(N/2) = -1
        else if (kb != k) o.emit(um, kb, untangle_bin(b, ac, cx<T>{-w.x, w.y}));
    }
}

template <typename T>
static int r2c_untangle_launch(int64_t M, int64_t N, const void* zf, int64_t ldz, int64_t isy, int64_t isx, int64_t osy, int64_t osx, int epi,
                               int norm_dc, double scale, void* out, int64_t ldo, hipStream_t st) {
    int err = 0;
    const cx<T>* twn = twiddles<T>(N, &err);
    if (!twn) return err;
    const cx<T>* twm = isy ? twiddles<T>(M, &err) : twn;
    if (!twm) return err;
    dim3 block(64, 4);
    int64_t gy = (M + 3) / 4;
    if (gy > 16384) gy = 16384;
    dim3 grid((unsigned)((N / 4 + 1 + 63) / 64), (unsigned)gy);
#define PM_UNT(E)                                                                                                                      \
    hipLaunchKernelGGL((r2c_untangle_kernel<T, E>), grid, block, 0, st, M, N, (const cx<T>*)zf, ldz, twn, twm, isy, isx, osy, osx, norm_dc, \
                       T(scale), out, ldo)
    switch (epi) {
        case PM_EPI_NONE: PM_UNT(PM_EPI_NONE); break;
        case PM_EPI_ABS: PM_UNT(PM_EPI_ABS); break;
        case PM_EPI_ABS2: PM_UNT(PM_EPI_ABS2); break;
        case PM_EPI_ARG: PM_UNT(PM_EPI_ARG); break;
        default: return fail(PM_ERR_ARG, "pm_r2c_untangle: epilogue must be PM_EPI_NONE, PM_EPI_ABS, PM_EPI_ABS2 or PM_EPI_ARG");
    }
#undef PM_UNT
    return int(hipGetLastError());
}

extern "C" {

int pm_r2c_untangle(int32_t dtype, int64_t M, int64_t N, const void* zf, int64_t zf_ld, int64_t in_shift_y, int64_t in_shift_x,
                    int64_t out_shift_y, int64_t out_shift_x, int32_t epilogue, int32_t norm_dc, double scale, void* out, int64_t out_ld,
                    void* stream) {
    if (!zf || !out || M < 1 || N < 2 || (N % 2) || zf_ld < N / 2 || out_ld < N) return fail(PM_ERR_ARG, "pm_r2c_untangle: bad argument (N must be even)");
    if (in_shift_y < 0 || in_shift_y >= M || in_shift_x < 0 || in_shift_x >= N || out_shift_y < 0 || out_shift_y >= M || out_shift_x < 0 ||
        out_shift_x >= N)
        return fail(PM_ERR_ARG, "pm_r2c_untangle: shifts must lie in [0, length)");
    if (dtype == PM_C64)
        return r2c_untangle_launch<float>(M, N, zf, zf_ld, in_shift_y, in_shift_x, out_shift_y, out_shift_x, epilogue, norm_dc, scale, out, out_ld, PM_STREAM(stream));
    if (dtype == PM_C128)
        return r2c_untangle_launch<double>(M, N, zf, zf_ld, in_shift_y, in_shift_x, out_shift_y, out_shift_x, epilogue, norm_dc, scale, out, out_ld, PM_STREAM(stream));
    return fail(PM_ERR_ARG, "pm_r2c_untangle: dtype must be PM_C64 or PM_C128");
}

int pm_cmul(int32_t dtype, int32_t op, int64_t rows, int64_t cols, const void* a, int64_t a_ld, const void* b,
            int64_t b_ld, void* out, int64_t out_ld, void* stream) {
    if (!a || !b || !out || rows < 0 || cols < 0) return fail(PM_ERR_ARG, "pm_cmul: bad argument");
    if (rows == 0 || cols == 0) return 0;
    if (dtype == PM_C64) return cmul_launch<float>(op, rows, cols, a, a_ld, b, b_ld, out, out_ld, PM_STREAM(stream));
    if (dtype == PM_C128) return cmul_launch<double>(op, rows, cols, a, a_ld, b, b_ld, out, out_ld, PM_STREAM(stream));
    return fail(PM_ERR_ARG, "pm_cmul: dtype must be PM_C64 or PM_C128");
}

int pm_rmul(int32_t dtype, int64_t rows, int64_t cols, const void* r, int64_t r_ld, const void* a, int64_t a_ld, double scale,
            void* out, int64_t out_ld, void* stream) {
    if (!r || !a || !out || rows < 0 || cols < 0) return fail(PM_ERR_ARG, "pm_rmul: bad argument");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(rmul_kernel<float>, grid, block, 0, PM_STREAM(stream), rows, cols, (const float*)r, r_ld, (const cx<float>*)a,
                           a_ld, float(scale), (cx<float>*)out, out_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(rmul_kernel<double>, grid, block, 0, PM_STREAM(stream), rows, cols, (const double*)r, r_ld,
                           (const cx<double>*)a, a_ld, scale, (cx<double>*)out, out_ld);
    else
        return fail(PM_ERR_ARG, "pm_rmul: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_scale_sep(int32_t dtype, int64_t rows, int64_t cols, const void* in, int64_t in_ld, const void* ry,
                 int32_t ry_conj, const void* cxv, int32_t cx_conj, double scale, void* out, int64_t out_ld,
                 void* stream) {
    if (!in || !out || rows < 0 || cols < 0) return fail(PM_ERR_ARG, "pm_scale_sep: bad argument");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(scale_sep_kernel<float>, grid, block, 0, PM_STREAM(stream), rows, cols, (const cx<float>*)in, in_ld,
                           (const cx<float>*)ry, ry_conj, (const cx<float>*)cxv, cx_conj, float(scale), (cx<float>*)out, out_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(scale_sep_kernel<double>, grid, block, 0, PM_STREAM(stream), rows, cols, (const cx<double>*)in, in_ld,
                           (const cx<double>*)ry, ry_conj, (const cx<double>*)cxv, cx_conj, scale, (cx<double>*)out, out_ld);
    else
        return fail(PM_ERR_ARG, "pm_scale_sep: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_abs2(int32_t dtype, int64_t rows, int64_t cols, const void* in, int64_t in_ld, void* out, int64_t out_ld,
            int32_t accumulate, double weight, void* stream) {
    if (!in || !out || rows < 0 || cols < 0) return fail(PM_ERR_ARG, "pm_abs2: bad argument");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    hipStream_t st = PM_STREAM(stream);
    if (dtype == PM_C64) {
        if (accumulate)
            hipLaunchKernelGGL((abs2_kernel<float, 1>), grid, block, 0, st, rows, cols, (const cx<float>*)in, in_ld, (float*)out, out_ld, float(weight));
        else
            hipLaunchKernelGGL((abs2_kernel<float, 0>), grid, block, 0, st, rows, cols, (const cx<float>*)in, in_ld, (float*)out, out_ld, 1.f);
    } else if (dtype == PM_C128) {
        if (accumulate)
            hipLaunchKernelGGL((abs2_kernel<double, 1>), grid, block, 0, st, rows, cols, (const cx<double>*)in, in_ld, (double*)out, out_ld, weight);
        else
            hipLaunchKernelGGL((abs2_kernel<double, 0>), grid, block, 0, st, rows, cols, (const cx<double>*)in, in_ld, (double*)out, out_ld, 1.0);
    } else
        return fail(PM_ERR_ARG, "pm_abs2: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_abs_arg(int32_t dtype, int64_t rows, int64_t cols, const void* in, int64_t in_ld, void* out_abs, int64_t abs_ld, void* out_arg,
               int64_t arg_ld, void* stream) {
    if (!in || (!out_abs && !out_arg) || rows < 0 || cols < 0) return fail(PM_ERR_ARG, "pm_abs_arg: bad argument");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(abs_arg_kernel<float>, grid, block, 0, PM_STREAM(stream), rows, cols, (const cx<float>*)in, in_ld, (float*)out_abs,
                           abs_ld, (float*)out_arg, arg_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(abs_arg_kernel<double>, grid, block, 0, PM_STREAM(stream), rows, cols, (const cx<double>*)in, in_ld,
                           (double*)out_abs, abs_ld, (double*)out_arg, arg_ld);
    else
        return fail(PM_ERR_ARG, "pm_abs_arg: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_sum_modes(int32_t dtype, int64_t nmodes, int64_t rows, int64_t cols, const void* modes, int64_t mode_stride,
                 int64_t modes_ld, const double* weights, int32_t accumulate, void* out, int64_t out_ld, void* stream) {
    if (!modes || !out || (!weights && nmodes > 0) || rows < 0 || cols < 0 || nmodes < 0) return fail(PM_ERR_ARG, "pm_sum_modes: bad argument");
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_sum_modes: dtype must be PM_C64 (float images) or PM_C128 (double)");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    hipStream_t st = PM_STREAM(stream);
    int acc = accumulate ? 1 : 0;
    if (nmodes == 0 && acc) return 0;
    int64_t b0 = 0;
    do {   // 32 modes per launch; an empty sum still writes zeros
        const int nb = int(nmodes - b0 < 32 ? nmodes - b0 : 32);
        if (dtype == PM_C64) {
            ModeWeights<float> mw;
            for (int i = 0; i < 32; ++i) mw.w[i] = i < nb ? float(weights[b0 + i]) : 0.f;
            hipLaunchKernelGGL(sum_modes_kernel<float>, grid, block, 0, st, rows, cols, (const float*)modes + b0 * mode_stride, mode_stride,
                               modes_ld, mw, nb, acc, (float*)out, out_ld);
        } else {
            ModeWeights<double> mw;
            for (int i = 0; i < 32; ++i) mw.w[i] = i < nb ? weights[b0 + i] : 0.0;
            hipLaunchKernelGGL(sum_modes_kernel<double>, grid, block, 0, st, rows, cols, (const double*)modes + b0 * mode_stride,
                               mode_stride, modes_ld, mw, nb, acc, (double*)out, out_ld);
        }
        acc = 1;
        b0 += 32;
    } while (b0 < nmodes);
    return int(hipGetLastError());
}

int pm_sample_map(int32_t dtype, int32_t order, int64_t map_rows, int64_t map_cols, const void* map, int64_t map_ld, double dx,
                  double center_x, double center_y, int64_t rows, int64_t cols, const void* xf, int64_t xf_sy, int64_t xf_sx,
                  const void* yf, int64_t yf_sy, int64_t yf_sx, const void* fill, int64_t fill_ld, double fill_re, double fill_im,
                  void* out, int64_t out_ld, void* stream) {
    if (!map || !xf || !yf || !out || rows < 0 || cols < 0 || map_rows < 1 || map_cols < 1 || !(dx != 0.0))
        return fail(PM_ERR_ARG, "pm_sample_map: bad argument");
    if (order != 0 && order != 1) return fail(PM_ERR_UNSUPPORTED, "pm_sample_map: spline order must be 0 or 1");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    hipStream_t st = PM_STREAM(stream);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(sample_map_kernel<float>, grid, block, 0, st, order, map_rows, map_cols, (const cx<float>*)map, map_ld, float(dx),
                           float(center_x), float(center_y), rows, cols, (const float*)xf, xf_sy, xf_sx, (const float*)yf, yf_sy, yf_sx,
                           (const cx<float>*)fill, fill_ld, cx<float>{float(fill_re), float(fill_im)}, (cx<float>*)out, out_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(sample_map_kernel<double>, grid, block, 0, st, order, map_rows, map_cols, (const cx<double>*)map, map_ld, dx,
                           center_x, center_y, rows, cols, (const double*)xf, xf_sy, xf_sx, (const double*)yf, yf_sy, yf_sx,
                           (const cx<double>*)fill, fill_ld, cx<double>{fill_re, fill_im}, (cx<double>*)out, out_ld);
    else
        return fail(PM_ERR_ARG, "pm_sample_map: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_spline_prefilter(int32_t dtype, int32_t order, int64_t map_rows, int64_t map_cols, const void* map, int64_t map_ld, void* coeff,
                        int64_t coeff_ld, void* stream) {
    if (!map || !coeff || map_rows < 1 || map_cols < 1 || map_ld < map_cols || coeff_ld < map_cols + 2 * kSplinePad)
        return fail(PM_ERR_ARG, "pm_spline_prefilter: bad argument");
    if (order < 2 || order > 5) return fail(PM_ERR_UNSUPPORTED, "pm_spline_prefilter: spline order must be 2 .. 5 (orders 0 and 1 need no prefilter)");
    if (dtype != PM_C64 && dtype != PM_C128) return fail(PM_ERR_ARG, "pm_spline_prefilter: dtype must be PM_C64 or PM_C128");
    hipStream_t st = PM_STREAM(stream);
    const SplinePoles pl = spline_poles(order);
    double gain = 1.0;
    for (int k = 0; k < pl.n; ++k) gain *= (1.0 - pl.z[k]) * (1.0 - 1.0 / pl.z[k]);
    const int64_t py = map_rows + 2 * kSplinePad, px = map_cols + 2 * kSplinePad;
    const dim3 block(256), grid(unsigned((py * px + 255) / 256));
    if (dtype == PM_C64)
        hipLaunchKernelGGL(spline_pad_kernel<float>, grid, block, 0, st, map_rows, map_cols, (const cx<float>*)map, map_ld, gain * gain,
                           (cx<double>*)coeff, coeff_ld);
    else
        hipLaunchKernelGGL(spline_pad_kernel<double>, grid, block, 0, st, map_rows, map_cols, (const cx<double>*)map, map_ld, gain * gain,
                           (cx<double>*)coeff, coeff_ld);
    hipLaunchKernelGGL(spline_filter_kernel, dim3(unsigned((px + 63) / 64)), dim3(64), 0, st, 0, py, px, (cx<double>*)coeff, coeff_ld, pl);
    hipLaunchKernelGGL(spline_filter_kernel, dim3(unsigned((py + 63) / 64)), dim3(64), 0, st, 1, py, px, (cx<double>*)coeff, coeff_ld, pl);
    return int(hipGetLastError());
}

int pm_sample_spline(int32_t dtype, int32_t order, int64_t map_rows, int64_t map_cols, const void* coeff, int64_t coeff_ld, double dx,
                     double center_x, double center_y, int64_t rows, int64_t cols, const void* xf, int64_t xf_sy, int64_t xf_sx,
                     const void* yf, int64_t yf_sy, int64_t yf_sx, const void* fill, int64_t fill_ld, double fill_re, double fill_im,
                     void* out, int64_t out_ld, void* stream) {
    if (!coeff || !xf || !yf || !out || rows < 0 || cols < 0 || map_rows < 1 || map_cols < 1 || !(dx != 0.0) ||
        coeff_ld < map_cols + 2 * kSplinePad)
        return fail(PM_ERR_ARG, "pm_sample_spline: bad argument");
    if (order < 2 || order > 5) return fail(PM_ERR_UNSUPPORTED, "pm_sample_spline: spline order must be 2 .. 5 (pm_sample_map serves 0 and 1)");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    hipStream_t st = PM_STREAM(stream);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(sample_spline_kernel<float>, grid, block, 0, st, order, map_rows, map_cols, (const cx<double>*)coeff, coeff_ld,
                           float(dx), float(center_x), float(center_y), rows, cols, (const float*)xf, xf_sy, xf_sx, (const float*)yf, yf_sy,
                           yf_sx, (const cx<float>*)fill, fill_ld, cx<float>{float(fill_re), float(fill_im)}, (cx<float>*)out, out_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(sample_spline_kernel<double>, grid, block, 0, st, order, map_rows, map_cols, (const cx<double>*)coeff, coeff_ld, dx,
                           center_x, center_y, rows, cols, (const double*)xf, xf_sy, xf_sx, (const double*)yf, yf_sy, yf_sx,
                           (const cx<double>*)fill, fill_ld, cx<double>{fill_re, fill_im}, (cx<double>*)out, out_ld);
    else
        return fail(PM_ERR_ARG, "pm_sample_spline: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_pupil_synth(int32_t dtype, int64_t rows, int64_t cols, const void* amp, int32_t amp_dtype, int64_t amp_ld,
                   const void* opd, int64_t opd_ld, double k, void* out, int64_t out_ld, void* stream) {
    if (!opd || !out || rows < 0 || cols < 0) return fail(PM_ERR_ARG, "pm_pupil_synth: bad argument");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    hipStream_t st = PM_STREAM(stream);
    const double k2 = k / (2.0 * 3.14159265358979323846264338327950288);
#define PM_PUPIL(T, A) \
    hipLaunchKernelGGL((pupil_kernel<T, A>), grid, block, 0, st, rows, cols, (const A*)amp, amp_ld, (const T*)opd, opd_ld, k2, (cx<T>*)out, out_ld)
    if (dtype == PM_C64) {
        if (!amp || amp_dtype == PM_F32) PM_PUPIL(float, float);
        else if (amp_dtype == PM_F64) PM_PUPIL(float, double);
        else if (amp_dtype == PM_BOOL) PM_PUPIL(float, unsigned char);
        else return fail(PM_ERR_ARG, "pm_pupil_synth: amp_dtype");
    } else if (dtype == PM_C128) {
        if (!amp || amp_dtype == PM_F64) PM_PUPIL(double, double);
        else if (amp_dtype == PM_F32) PM_PUPIL(double, float);
        else if (amp_dtype == PM_BOOL) PM_PUPIL(double, unsigned char);
        else return fail(PM_ERR_ARG, "pm_pupil_synth: amp_dtype");
    } else
        return fail(PM_ERR_ARG, "pm_pupil_synth: dtype must be PM_C64 or PM_C128");
#undef PM_PUPIL
    return int(hipGetLastError());
}

int pm_quadratic_phase(int32_t dtype, int64_t rows, int64_t cols, const void* x, int64_t x_ld, const void* y,
                       int64_t y_ld, double c, void* out, int64_t out_ld, void* stream) {
    if (!x || !y || !out || rows < 0 || cols < 0) return fail(PM_ERR_ARG, "pm_quadratic_phase: bad argument");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    const double c2 = c / (2.0 * 3.14159265358979323846264338327950288);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(quad_phase_kernel<float>, grid, block, 0, PM_STREAM(stream), rows, cols, (const float*)x, x_ld, (const float*)y, y_ld, c2, (cx<float>*)out, out_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(quad_phase_kernel<double>, grid, block, 0, PM_STREAM(stream), rows, cols, (const double*)x, x_ld, (const double*)y, y_ld, c2, (cx<double>*)out, out_ld);
    else
        return fail(PM_ERR_ARG, "pm_quadratic_phase: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_as_tf_vectors(int32_t dtype, int64_t rows, int64_t cols, double wvl_um, double dx, double z, void* hy, void* hx,
                     void* stream) {
    if (!hy || !hx || rows <= 0 || cols <= 0) return fail(PM_ERR_ARG, "pm_as_tf_vectors: bad argument");
    // exp(-i pi (wvl/1e3) z k^2) = exp(2 pi i * (-(wvl/1e3) z / 2) k^2)
    const double coef = -(wvl_um / 1e3) * z * 0.5;
    hipStream_t st = PM_STREAM(stream);
    const unsigned blocks = unsigned((rows + 255) / 256 + (cols + 255) / 256);
    if (dtype == PM_C64) {
        hipLaunchKernelGGL(as_tf_vec_kernel<float>, dim3(blocks), dim3(256), 0, st, rows, cols, dx, coef, (cx<float>*)hy, (cx<float>*)hx);
    } else if (dtype == PM_C128) {
        hipLaunchKernelGGL(as_tf_vec_kernel<double>, dim3(blocks), dim3(256), 0, st, rows, cols, dx, coef, (cx<double>*)hy, (cx<double>*)hx);
    } else
        return fail(PM_ERR_ARG, "pm_as_tf_vectors: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_outer(int32_t dtype, int64_t rows, int64_t cols, const void* hy, const void* hx, void* out, int64_t out_ld,
             void* stream) {
    if (!hy || !hx || !out || rows < 0 || cols < 0) return fail(PM_ERR_ARG, "pm_outer: bad argument");
    if (rows == 0 || cols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(rows, cols, block);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(outer_kernel<float>, grid, block, 0, PM_STREAM(stream), rows, cols, (const cx<float>*)hy, (const cx<float>*)hx, (cx<float>*)out, out_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(outer_kernel<double>, grid, block, 0, PM_STREAM(stream), rows, cols, (const cx<double>*)hy, (const cx<double>*)hx, (cx<double>*)out, out_ld);
    else
        return fail(PM_ERR_ARG, "pm_outer: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_embed(int32_t elem_bytes, int64_t irows, int64_t icols, const void* in, int64_t in_ld, int64_t orows,
             int64_t ocols, int64_t off_y, int64_t off_x, const void* fill, void* out, int64_t out_ld, void* stream) {
    if (!in || !out || irows < 0 || icols < 0 || orows < 0 || ocols < 0) return fail(PM_ERR_ARG, "pm_embed: bad argument");
    if (orows == 0 || ocols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(orows, ocols, block);
    hipStream_t st = PM_STREAM(stream);
#define PM_EMBED(V)                                                                                              \
    {                                                                                                            \
        V f{};                                                                                                   \
        if (fill) f = *reinterpret_cast<const V*>(fill);                                                         \
        hipLaunchKernelGGL(embed_kernel<V>, grid, block, 0, st, irows, icols, (const V*)in, in_ld, orows, ocols, \
                           off_y, off_x, f, (V*)out, out_ld);                                                    \
    }
    switch (elem_bytes) {
        case 1: PM_EMBED(unsigned char) break;
        case 4: PM_EMBED(float) break;
        case 8: PM_EMBED(double) break;
        case 16: PM_EMBED(double2) break;
        default: return fail(PM_ERR_ARG, "pm_embed: elem_bytes must be 1, 4, 8 or 16");
    }
#undef PM_EMBED
    return int(hipGetLastError());
}

int pm_pad_index(int32_t elem_bytes, int32_t mode, int64_t irows, int64_t icols, const void* in, int64_t in_ld, int64_t orows,
                 int64_t ocols, int64_t off_y, int64_t off_x, void* out, int64_t out_ld, void* stream) {
    if (!in || !out || irows < 1 || icols < 1 || orows < 0 || ocols < 0 || mode < 1 || mode > 4)
        return fail(PM_ERR_ARG, "pm_pad_index: bad argument");
    if (orows == 0 || ocols == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(orows, ocols, block);
    hipStream_t st = PM_STREAM(stream);
#define PM_PADI(V) \
    hipLaunchKernelGGL(pad_index_kernel<V>, grid, block, 0, st, mode, irows, icols, (const V*)in, in_ld, orows, ocols, off_y, off_x, (V*)out, out_ld)
    switch (elem_bytes) {
        case 1: PM_PADI(unsigned char); break;
        case 4: PM_PADI(float); break;
        case 8: PM_PADI(double); break;
        case 16: PM_PADI(double2); break;
        default: return fail(PM_ERR_ARG, "pm_pad_index: elem_bytes must be 1, 4, 8 or 16");
    }
#undef PM_PADI
    return int(hipGetLastError());
}

int pm_mdft_basis(int32_t dtype, int64_t M, int64_t N, const void* f, const void* x, int32_t sign, void* E,
                  int64_t E_ld, void* stream) {
    if (!f || !x || !E || M < 0 || N < 0 || (sign != 1 && sign != -1)) return fail(PM_ERR_ARG, "pm_mdft_basis: bad argument");
    if (M == 0 || N == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(M, N, block);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(mdft_basis_kernel<float>, grid, block, 0, PM_STREAM(stream), M, N, (const float*)f, (const float*)x, double(sign), (cx<float>*)E, E_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(mdft_basis_kernel<double>, grid, block, 0, PM_STREAM(stream), M, N, (const double*)f, (const double*)x, double(sign), (cx<double>*)E, E_ld);
    else
        return fail(PM_ERR_ARG, "pm_mdft_basis: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_mdft_basis_grid(int32_t dtype, int64_t M, int64_t N, double f_step, double f_shift, double f_scale, double x_step,
                       int32_t sign, void* E, int64_t E_ld, void* stream) {
    if (!E || M < 0 || N < 0 || (sign != 1 && sign != -1)) return fail(PM_ERR_ARG, "pm_mdft_basis_grid: bad argument");
    if (M == 0 || N == 0) return 0;
    dim3 block;
    dim3 grid = grid2d(M, N, block);
    if (dtype == PM_C64)
        hipLaunchKernelGGL(mdft_basis_grid_kernel<float>, grid, block, 0, PM_STREAM(stream), M, N, float(f_step), float(f_shift),
                           float(f_scale), float(x_step), double(sign), (cx<float>*)E, E_ld);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(mdft_basis_grid_kernel<double>, grid, block, 0, PM_STREAM(stream), M, N, f_step, f_shift, f_scale, x_step,
                           double(sign), (cx<double>*)E, E_ld);
    else
        return fail(PM_ERR_ARG, "pm_mdft_basis_grid: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

int pm_czt_vectors(int32_t dtype, int64_t N, int64_t M, int64_t K, double shift, double half, void* b, void* a, void* h, void* stream) {
    if (!b || !a || !h || N < 1 || M < 1 || K < N + M - 1) return fail(PM_ERR_ARG, "pm_czt_vectors: bad argument");
    const int64_t total = N + M + K;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == PM_C64)
        hipLaunchKernelGGL(czt_vectors_kernel<float>, grid, dim3(256), 0, PM_STREAM(stream), N, M, K, shift, half, (cx<float>*)b, (cx<float>*)a,
                           (cx<float>*)h);
    else if (dtype == PM_C128)
        hipLaunchKernelGGL(czt_vectors_kernel<double>, grid, dim3(256), 0, PM_STREAM(stream), N, M, K, shift, half, (cx<double>*)b,
                           (cx<double>*)a, (cx<double>*)h);
    else
        return fail(PM_ERR_ARG, "pm_czt_vectors: dtype must be PM_C64 or PM_C128");
    return int(hipGetLastError());
}

}  // extern "C"
