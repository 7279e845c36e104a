// mixed-radix rows pass, complex64: entry point and the kernel classes of factors up to 10 and up to 16 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_rows_launch<float, 10>(const MixPlan* p, MixShape sh, const DirectIn<float>& in, const MixRowOut<float>& ro, const cx<float>* tw, int groups, int nt, size_t lds, hipStream_t st) {
    return mix_rows_launch_impl<float, 10>(p, sh, in, ro, tw, groups, nt, lds, st);
}
template <> int mix_rows_launch<float, 16>(const MixPlan* p, MixShape sh, const DirectIn<float>& in, const MixRowOut<float>& ro, const cx<float>* tw, int groups, int nt, size_t lds, hipStream_t st) {
    return mix_rows_launch_impl<float, 16>(p, sh, in, ro, tw, groups, nt, lds, st);
}

template <> int mix_rows<float>(const DirectIn<float>& in, cx<float>* out, int64_t out_ld, hipStream_t st, const RowStoreNat<float>* o, const MixFold<float>* fold, int64_t out_bstride) {
    return mix_rows_impl<float>(in, out, out_ld, st, o, fold, out_bstride);
}

}  // namespace pm
