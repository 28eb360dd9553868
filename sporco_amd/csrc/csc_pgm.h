// csc_pgm.h -- the column passes of the fused PGM (FISTA) iteration of
// pgm.cbpdn.ConvBPDN (sporco/pgm/pgm.py:779-846, sporco/pgm/cbpdn.py:263-372).
//
// With all spectral iterates kept tile-major (csc_fused.h) one default-option
// iteration -- gradient step, proximal step, momentum, residual and objective --
// is three launches and nine float32-array passes:
//
//   pgm_grad_ifft      Yf            -> T    Vf = Yf - conj(Df)(sum_k Df Yf - Sf)/L, IFFT along H
//   rows_inv_prox_fwd  T             -> T'   csc_rows.h: irfft along W, prox_l1, rfft along W
//   pgm_fft_momentum   T', Xf, Yf    -> Xf', Yf'   FFT along H, Yf' = Xf' + beta (Xf' - Xf),
//                                            sums for rsdl (pgm/cbpdn.py:314-320) and the objective
//
// Both kernels are the register-resident column transform of csc_fused.hip with
// one LDS exchange instead of two.
#pragma once

#include "common.h"

namespace sporco_amd {

template <typename T> struct PgmColsArgs {
    // tile-major (Wf, CN, H, K) complex arrays
    const cx<T> *yf;      // grad_ifft: input Yf;       fft_momentum: Yf of the previous iteration
    const cx<T> *xf_old;  // fft_momentum: Xf of the previous iteration
    cx<T> *t;             // grad_ifft: output T;       fft_momentum: T' in, new Xf out (in place)
    cx<T> *yf_new;        // fft_momentum: output Yf'
    const cx<T> *dft, *sft;
    const cx<T> *twA, *twB;  // fused_twiddles of csc_fused.h
    T inv_L, beta;
    int H, W, CN, K;
    int want_stats;       // fft_momentum: also evaluate f(Xf') (needs the Df inner products)
    double *partials;     // grad_ifft: [tile] sum |sum_k Df Yf - Sf|^2;
                          // fft_momentum: [tile][4] pw*|Xf' - Yf|^2, pw*|e|^2, |e|^2, 0
};

template <typename T> int64_t launch_pgm_grad_ifft(hipStream_t st, const PgmColsArgs<T> &a);
template <typename T> int64_t launch_pgm_fft_momentum(hipStream_t st, const PgmColsArgs<T> &a);

}  // namespace sporco_amd
