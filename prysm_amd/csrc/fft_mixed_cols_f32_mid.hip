// mixed-radix cols pass, complex64: the kernel class of factors up to 20 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_cols_launch<float, 20>(const MixPlan* p, MixShape sh, const DirectIn<float>& in, const ColStoreNat<float>& out, const cx<float>* tw, int log_g, int groups, int nt, size_t lds,
                                  hipStream_t st) {
    return mix_cols_launch_impl<float, 20>(p, sh, in, out, tw, log_g, groups, nt, lds, st);
}

}  // namespace pm
