// mixed-radix rows pass, complex128: the kernel class of factors up to 20 (fft_mixed_kernels.h)
#include "fft_mixed_kernels.h"

namespace pm {

template <> int mix_rows_launch<double, 20>(const MixPlan* p, MixShape sh, const DirectIn<double>& in, const MixRowOut<double>& ro, const cx<double>* tw, int groups, int nt, size_t lds, hipStream_t st) {
    return mix_rows_launch_impl<double, 20>(p, sh, in, ro, tw, groups, nt, lds, st);
}

}  // namespace pm
