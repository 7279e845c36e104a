// mixed-radix cols pass, complex128 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_cols<double>(const DirectIn<double>& in, const ColStoreNat<double>& out, hipStream_t st) {
    return mix_cols_impl<double>(in, out, st);
}

}  // namespace pm
