// Parameter block and launcher declarations of the fused one-axis chirp-Z convolution (kernels: fft_conv1.h).
#pragma once
#include <hip/hip_runtime.h>

#include "fft_engine.h"

namespace pm {

template <typename T>
struct Conv1 {
    const cx<T>* in;
    int64_t in_ld;
    cx<T>* out;
    int64_t out_ld;
    int nseq;               // sequences: rows (axis 1) or columns (axis 0)
    int in_len, in_off;     // the input samples sit at [in_off, in_off + in_len) of the K-point sequence, zero elsewhere
    int out_len, out_off;   // results [out_off, out_off + out_len) are stored
    const cx<T>* pre;       // in_len factors on the input (or null)
    const cx<T>* H;         // K factors between the transforms
    const cx<T>* post;      // out_len factors on the output (or null)
    int pre_conj, h_conj, post_conj;
    T scale;
    int single;             // 0: the convolution above; 1 / 2: ONE transform of the K-point sequence, forward exp(-2 pi i ..) / inverse
                            // exp(+..) unnormalised (H unused): out[m] = scale post[m] FFT_K(pad_K(pre . in))[out_off + m] -- one axis of
                            // fttools.FFTDFT with its phase ramps in the load and the store (pm_fft1_ramp)
};

template <typename T> int launch_conv1_rows(int logk, const Conv1<T>&, const cx<T>* tw, hipStream_t);
template <typename T> int launch_conv1_cols(int logk, const Conv1<T>&, const cx<T>* tw, hipStream_t);

}  // namespace pm
