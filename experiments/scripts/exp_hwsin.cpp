// accuracy of the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) against double, on [-0.5, 0.5]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(int n, float* s, float* c, float* s2, float* c2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = -0.5f + float(i) / float(n);
    s[i] = __builtin_amdgcn_sinf(x);
    c[i] = __builtin_amdgcn_cosf(x);
    float ss, cc;
    sincospif(2.0f * x, &ss, &cc);
    s2[i] = ss;
    c2[i] = cc;
}
int main() {
    const int n = 1 << 22;
    float *s, *c, *s2, *c2;
    hipMalloc(&s, n * 4); hipMalloc(&c, n * 4); hipMalloc(&s2, n * 4); hipMalloc(&c2, n * 4);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, s, c, s2, c2);
    std::vector<float> hs(n), hc(n), hs2(n), hc2(n);
    hipMemcpy(hs.data(), s, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hs2.data(), s2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc2.data(), c2, n * 4, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        float x = -0.5f + float(i) / float(n);
        double rs = sin(2 * M_PI * double(x)), rc = cos(2 * M_PI * double(x));
        e1 = fmax(e1, fmax(fabs(hs[i] - rs), fabs(hc[i] - rc)));
        e2 = fmax(e2, fmax(fabs(hs2[i] - rs), fabs(hc2[i] - rc)));
    }
    printf("max abs error: hardware v_sin/v_cos %.3e, sincospif %.3e\n", e1, e2);
    return 0;
}
