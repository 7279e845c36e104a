// mixed-radix MIDDLE pass of the fused fft2 -> x H -> ifft2 chain on composite column lengths, complex128: forward stages, multiplier,
// transposed stages with the columns resident in LDS (fft_mixed.h, fft_mixed_kernels.h mix_cols_mul_kernel), all three kernel classes
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_cols_mul<double>(const DirectIn<double>& in, const MidMul<double>& mm, cx<double>* dst, int64_t dst_pitch, hipStream_t st) {
    return mix_cols_mul_impl<double>(in, mm, dst, dst_pitch, st);
}

}  // namespace pm
