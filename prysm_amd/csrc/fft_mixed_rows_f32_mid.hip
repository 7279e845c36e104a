// mixed-radix rows pass, complex64: the kernel class of factors up to 20 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_rows_launch<float, 20>(const MixPlan* p, MixShape sh, const DirectIn<float>& in, const MixRowOut<float>& ro, const cx<float>* tw, int groups, int nt, size_t lds, hipStream_t st) {
    return mix_rows_launch_impl<float, 20>(p, sh, in, ro, tw, groups, nt, lds, st);
}

}  // namespace pm
