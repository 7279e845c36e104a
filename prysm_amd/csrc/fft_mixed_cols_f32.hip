// mixed-radix cols pass, complex64: entry point and the kernel classes of factors up to 10 and up to 16 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_cols_launch<float, 10>(const MixPlan* p, MixShape sh, const DirectIn<float>& in, const ColStoreNat<float>& out, const cx<float>* tw, int log_g, int groups, int nt, size_t lds,
                                  hipStream_t st) {
    return mix_cols_launch_impl<float, 10>(p, sh, in, out, tw, log_g, groups, nt, lds, st);
}
template <> int mix_cols_launch<float, 16>(const MixPlan* p, MixShape sh, const DirectIn<float>& in, const ColStoreNat<float>& out, const cx<float>* tw, int log_g, int groups, int nt, size_t lds,
                                  hipStream_t st) {
    return mix_cols_launch_impl<float, 16>(p, sh, in, out, tw, log_g, groups, nt, lds, st);
}

template <> int mix_cols<float>(const DirectIn<float>& in, const ColStoreNat<float>& out, hipStream_t st) {
    return mix_cols_impl<float>(in, out, st);
}

}  // namespace pm
