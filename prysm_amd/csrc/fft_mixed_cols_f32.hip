// mixed-radix cols pass, complex64 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_cols<float>(const DirectIn<float>& in, const ColStoreNat<float>& out, hipStream_t st) {
    return mix_cols_impl<float>(in, out, st);
}

}  // namespace pm
