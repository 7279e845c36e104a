// Fused one-axis chirp-Z convolution kernels, float precision (explicit instantiation; see fft_conv1.h).
#include "fft_conv1.h"
namespace pm {
template <> int launch_conv1_rows<float>(int logk, const Conv1<float>& p, const cx<float>* tw, hipStream_t st) { return launch_conv1_impl<float, false>(logk, p, tw, st); }
template <> int launch_conv1_cols<float>(int logk, const Conv1<float>& p, const cx<float>* tw, hipStream_t st) { return launch_conv1_impl<float, true>(logk, p, tw, st); }
}  // namespace pm
