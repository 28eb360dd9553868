// csc_post_elem.h -- the ADMM epilogue of one coefficient (relax_AX, ystep, ustep and the terms
// of the residual / objective sums: admm.py:877-885, cbpdn.py:614-620, :297-311, admm.py:434-437,
// :462-486), shared by the streaming epilogue kernel (csc_kernels.hip admm_post_kernel) and the
// epilogue fused into the store of the generic half-spectrum -> real row pass (fft.hip).
#pragma once

#include "csc_kernels.h"

namespace sporco_amd {

template <typename T>
__device__ __forceinline__ T weight_at(const Weight<T> &w, int h, int x, int c, int n, int k) {
    return w.ptr[h * w.stride[0] + x * w.stride[1] + c * w.stride[2] + n * w.stride[3] +
                 k * w.stride[4]];
}

template <typename T> __device__ __forceinline__ T soft(T v, T thr) {
    // sign(v) * max(|v| - thr, 0)            (prox/_lp.py:181)
    T m = (v < T(0) ? -v : v) - thr;
    m = m > T(0) ? m : T(0);
    return v < T(0) ? -m : m;
}

// True when (h, x) lies in the band zeroed by NoBndryCross: Y[1-dH:, ...] = 0,
// Y[:, 1-dW:, ...] = 0  (cbpdn.py:308-311; a size-1 filter gives slice(0, None),
// i.e. the whole axis, which is mirrored here).
__device__ __forceinline__ bool in_bndry(int h, int x, int H, int W, int dH, int dW) {
    const int h0 = (dH > 1) ? H - (dH - 1) : 0;
    const int x0 = (dW > 1) ? W - (dW - 1) : 0;
    return h >= h0 || x >= x0;
}

// filter k is one of the ams_n impulse filters AddMaskSim appended at ams_k (one, or one per
// channel of a multi-channel dictionary: cbpdn.py:2339-2346); ams_k < 0: there are none
__device__ __forceinline__ bool is_ams(int k, int ams_k, int ams_n) {
    return ams_k >= 0 && k >= ams_k && k < ams_k + ams_n;
}

// idx: flat index of the element in the (H, W, C, N, K) array, P = C N K.  x: X at idx; y, u: the
// iterate at idx on entry (u unscaled), the new iterate on return (vnew: AX + U of it, from which
// both follow).  acc: r2, s2, ax2, y2, u2, l1.
// GENERAL: weight arrays, NoBndryCross or AddMaskSim need the 5-D index of the element.
template <typename T, bool GENERAL>
__device__ __forceinline__ void admm_post_elem(const PostParams<T> &p, int64_t idx, int64_t P, T x, T &y, T &u,
                                               double (&acc)[8], T *vnew = nullptr) {
    const T a = p.rlx, oma = T(1) - p.rlx;
    const bool nonneg = p.flags & F_NONNEG, nob = p.flags & F_NOBNDRY, gy = p.flags & F_GEVAL_Y;
    const T yo = y, uo = p.u_scale * u;
    const T ax = a * x + oma * yo;
    T w = T(1);
    bool kill = false, ams = false;
    if (GENERAL) {
        const int64_t pix = idx / P;
        const int r = (int)(idx - pix * P);
        const int k = r % p.d.K, n = (r / p.d.K) % p.d.N, c = r / (p.d.K * p.d.N);
        const int xw = (int)(pix % p.d.W), h = (int)(pix / p.d.W);
        if (p.wl1.ptr) w = weight_at(p.wl1, h, xw, c, n, k);
        kill = nob && in_bndry(h, xw, p.d.H, p.d.W, p.dH, p.dW);
        if (p.ams.ptr && is_ams(k, p.ams_k, p.ams_n)) {
            // AddMaskSim impulse slice (cbpdn.py:2378-2394): no shrinkage, no
            // NonNeg / NoBndryCross, zero where the mask is set; invisible to
            // the regulariser (:2398-2412)
            ams = true;
            w = T(0);
            kill = weight_at(p.ams, h, xw, c, n, k - p.ams_k) != T(0);
        }
    }
    const T vsum = ax + uo;   // (the iterate in its single-array form: y and u below follow from it)
    if (vnew) *vnew = vsum;
    T yn = soft(vsum, p.thr * w);
    if (nonneg && !ams && yn < T(0)) yn = T(0);
    if (kill) yn = T(0);
    const T un = uo + ax - yn;
    y = yn;
    u = un;
    const double dr = (double)(x - yn), ds = (double)(yn - yo);
    acc[0] += dr * dr;
    acc[1] += ds * ds;
    acc[2] += (double)x * (double)x;
    acc[3] += (double)yn * (double)yn;
    acc[4] += (double)un * (double)un;
    const T gv = w * (gy ? yn : x);
    acc[5] += (double)(gv < T(0) ? -gv : gv);
}

}  // namespace sporco_amd
