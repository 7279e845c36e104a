// mixed-radix cols pass, complex128: entry point and the kernel classes of factors up to 10 and up to 16 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_cols_launch<double, 10>(const MixPlan* p, MixShape sh, const DirectIn<double>& in, const ColStoreNat<double>& out, const cx<double>* tw, int log_g, int groups, int nt, size_t lds,
                                  hipStream_t st) {
    return mix_cols_launch_impl<double, 10>(p, sh, in, out, tw, log_g, groups, nt, lds, st);
}
template <> int mix_cols_launch<double, 16>(const MixPlan* p, MixShape sh, const DirectIn<double>& in, const ColStoreNat<double>& out, const cx<double>* tw, int log_g, int groups, int nt, size_t lds,
                                  hipStream_t st) {
    return mix_cols_launch_impl<double, 16>(p, sh, in, out, tw, log_g, groups, nt, lds, st);
}

template <> int mix_cols<double>(const DirectIn<double>& in, const ColStoreNat<double>& out, hipStream_t st) {
    return mix_cols_impl<double>(in, out, st);
}

}  // namespace pm
