// csc_kernels_dev.h -- device-side helpers shared by the generic-kernel translation units
// (ck_admm.hip, ck_dict.hip, ck_ism.hip, ck_misc.hip: what used to be one csc_kernels.hip): launch
// geometry, vector types, the half-spectrum Parseval weight, the complex wave reduction.
#pragma once

#include "csc_kernels.h"

#include <gfx950_intrin.h>
#include "csc_ctl_dev.h"
#include "csc_post_elem.h"

#include "../../include/sporco_amd.h"

namespace sporco_amd {

constexpr int kThreads = 256;

template <typename T, int V> struct alignas(sizeof(T) * V) Vec {
    T v[V];
};

template <typename T> struct alignas(2 * sizeof(cx<T>)) cxpair {
    cx<T> a, b;
};

static inline int grid_for(int64_t work_items, int threads = kThreads) {
    int64_t g = ceil_div(work_items, threads);
    if (g < 1) g = 1;
    if (g > kMaxPartialBlocks) g = kMaxPartialBlocks;
    return (int)g;
}



__device__ __forceinline__ double parseval_weight(int wf, int Wf, int W) {
    // weights 1, 2, ..., 2, (1 if W even else 2) over the half spectrum (fft.py:476-484)
    return (wf == 0 || ((W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
}

template <typename T> __device__ __forceinline__ cx<T> wave_sum_cx(cx<T> v) {
#pragma unroll
    for (int m = kWave / 2; m > 0; m >>= 1) {
        v.re += __shfl_xor(v.re, m, kWave);
        v.im += __shfl_xor(v.im, m, kWave);
    }
    return v;
}

// Weighted gradient spectrum w_k * sum_i |G_i|^2 at (pixel, filter): GHGf of cbpdn.py:1141-1143
template <typename T>
__device__ __forceinline__ T grad_gh(const GradTerm<T> &g, int64_t pix, int Wf) {
    return g.ghh[pix / Wf] + g.ghw[pix % Wf];
}
template <typename T> __device__ __forceinline__ T grad_w(const GradTerm<T> &g, int k) {
    return g.wg ? g.wg[k] : T(1);
}

}  // namespace sporco_amd
