// mixed-radix rows pass, complex64 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_rows<float>(const DirectIn<float>& in, cx<float>* out, int64_t out_ld, hipStream_t st, const RowStoreNat<float>* o) {
    return mix_rows_impl<float>(in, out, out_ld, st, o);
}

}  // namespace pm
