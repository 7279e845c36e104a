// mixed-radix rows pass, complex128 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_rows<double>(const DirectIn<double>& in, cx<double>* out, int64_t out_ld, hipStream_t st, const RowStoreNat<double>* o) {
    return mix_rows_impl<double>(in, out, out_ld, st, o);
}

}  // namespace pm
