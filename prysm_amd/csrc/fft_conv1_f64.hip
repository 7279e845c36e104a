// Fused one-axis chirp-Z convolution kernels, double precision (explicit instantiation; see fft_conv1.h).
#include "fft_conv1.h"
namespace pm {
template <> int launch_conv1_rows<double>(int logk, const Conv1<double>& p, const cx<double>* tw, hipStream_t st) { return launch_conv1_impl<double, false>(logk, p, tw, st); }
template <> int launch_conv1_cols<double>(int logk, const Conv1<double>& p, const cx<double>* tw, hipStream_t st) { return launch_conv1_impl<double, true>(logk, p, tw, st); }
}  // namespace pm
