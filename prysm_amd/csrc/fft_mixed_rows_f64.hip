// mixed-radix rows pass, complex128: entry point and the kernel classes of factors up to 10 and up to 16 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_rows_launch<double, 10>(const MixPlan* p, MixShape sh, const DirectIn<double>& in, const MixRowOut<double>& ro, const cx<double>* tw, int groups, int nt, size_t lds, hipStream_t st) {
    return mix_rows_launch_impl<double, 10>(p, sh, in, ro, tw, groups, nt, lds, st);
}
template <> int mix_rows_launch<double, 16>(const MixPlan* p, MixShape sh, const DirectIn<double>& in, const MixRowOut<double>& ro, const cx<double>* tw, int groups, int nt, size_t lds, hipStream_t st) {
    return mix_rows_launch_impl<double, 16>(p, sh, in, ro, tw, groups, nt, lds, st);
}

template <> int mix_rows<double>(const DirectIn<double>& in, cx<double>* out, int64_t out_ld, hipStream_t st, const RowStoreNat<double>* o, const MixFold<double>* fold, int64_t out_bstride) {
    return mix_rows_impl<double>(in, out, out_ld, st, o, fold, out_bstride);
}

}  // namespace pm
