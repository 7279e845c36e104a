// Fixed cost of a kernel in a stream of dependent kernels, for the launch shapes of the 2048^2 passes: an (almost) empty kernel
// launched back to back.  What a fused single-launch 2-D transform could save is one of these per propagation.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", int(e_), __LINE__); return 1; } } while (0)
__global__ void empty(float* p) {
    extern __shared__ char smem[];
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) p[0] = smem[0];
}
__global__ void touch(float* p, size_t n) {     // every thread writes 8 bytes x 16: the store footprint of an FFT pass (dirty lines for the end-of-kernel write-back)
    extern __shared__ char smem[];
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    for (int m = 0; m < 16; ++m) reinterpret_cast<float2*>(p)[(i + size_t(m) * gridDim.x * blockDim.x) % (n / 8)] = float2{1.f, 2.f};
}
int main() {
    float* buf;
    const size_t bytes = size_t(32) << 20;
    CK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { int grid, block, lds; const char* name; } shapes[] = {{2048, 128, 17408, "row pass 2048^2 c64"}, {256, 512, 69632, "column pass 2048^2 c64"},
                                                                   {1024, 256, 34816, "row pass 4096 rows c64"}, {256, 1024, 139264, "column pass 4096 c64 unfolded"}};
    for (auto& s : shapes) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(empty), hipFuncAttributeMaxDynamicSharedMemorySize, s.lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(touch), hipFuncAttributeMaxDynamicSharedMemorySize, s.lds));
        for (int mode = 0; mode < 2; ++mode) {
            for (int i = 0; i < 20; ++i) {
                if (mode) hipLaunchKernelGGL(touch, dim3(s.grid), dim3(s.block), s.lds, 0, buf, bytes);
                else hipLaunchKernelGGL(empty, dim3(s.grid), dim3(s.block), s.lds, 0, buf);
            }
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 200; ++i) {
                if (mode) hipLaunchKernelGGL(touch, dim3(s.grid), dim3(s.block), s.lds, 0, buf, bytes);
                else hipLaunchKernelGGL(empty, dim3(s.grid), dim3(s.block), s.lds, 0, buf);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-32s grid %4d x %4d threads, %6d B LDS: %s kernel %.2f us per launch\n", s.name, s.grid, s.block, s.lds,
                   mode ? "32 MiB store" : "empty", ms / 200 * 1e3);
        }
    }
    return 0;
}
