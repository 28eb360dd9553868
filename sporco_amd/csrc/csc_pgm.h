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
    cx<T> *ey;            // tile-major (Wf, CN, H), or null: e_y = sum_k Df Yf - Sf per frequency,
                          // written by grad_ifft and read by fft_momentum (with want_stats) for
                          // the linear term of the backtracking model Q_L (pgm.py:886-894)
    const cx<T> *ey_in = nullptr;   // grad_ifft: the residual per frequency, tile-major (Wf, CN, H), taken
                          // from memory instead of formed from Yf (the masked classes: it has
                          // been through the spatial domain for the mask); K <= 64, no held trial
    cx<T> *qpart = nullptr;   // K > 64 (fft_momentum runs per (tile, 64-filter slab), grid.y = slabs):
                          // with want_stats the slab's share of sum_k Df Xf' goes to
                          // qpart[tile][slab][f] and launch_pgm_stats_slabs forms the objective sums
    // persistent launch (H = 512, K = 64; csc_fused.h): workgroup slot s starts
    // (s % stagger_groups) * stagger_sleeps * 8128 cycles late (set by the launchers)
    int stagger_groups = 1, stagger_sleeps = 0;
    double *partials;     // grad_ifft: [tile] sum |sum_k Df Yf - Sf|^2 (only with ey: a held trial);
                          // fft_momentum: [tile][6] pw*|Xf' - Yf|^2, pw*|e|^2, |e|^2,
                          //   Re<e - e_y, e_y> (= <Xf' - Yf, grad f(Yf)>, 0 without ey), |Xf' - Yf|^2, 0
};
constexpr int kPgmPartialStride = 6;

template <typename T> int64_t launch_pgm_grad_ifft(hipStream_t st, const PgmColsArgs<T> &a);
// t <- FFT_H(t) only (row spectra -> full tile-major spectrum, in place): uses t, twA, H, W, CN, K
template <typename T> int64_t launch_cols_fft(hipStream_t st, const PgmColsArgs<T> &a);

// Dictionary-update gradient on a tile-major coefficient spectrum (pgm/ccmod.py:295-317):
//     r[n, f]  = sum_k zf[n, f, k] d[f, k] - sf[n, f]
//     g[f, k]  = sum_n conj(zf[n, f, k]) r[n, f]
// One workgroup owns a row frequency wf and a fixed group of the C*N tiles, so the sum
// over images is accumulated in registers in a fixed order; the G group partials are
// added by launch_sum_groups.  d and the gradient are in the reference layout (H, Wf, K).
template <typename T> struct CcmodTiledArgs {
    const cx<T> *zf;     // tile-major (Wf, CN, H, K)
    const cx<T> *d;      // (H, Wf, K)
    const cx<T> *sft;    // tile-major (Wf, CN, H)
    cx<T> *gpart;        // (G, H, Wf, K), or null: sums only
    int H, W, CN, K, G;
    double *partials;    // per workgroup 4 doubles: |r|^2, pw |r|^2, |r + sf|^2, 0
    // K > 64 (two passes over zf, one workgroup per (row frequency, group, 64-filter slab)): the
    // slabs' shares of sum_k zf d, (Wf*CN, slabs, H) complex, and the residual r, (Wf*CN, H)
    // complex; `partials` then has one row per tile
    cx<T> *qpart = nullptr, *rbuf = nullptr;
};
template <typename T> bool ccmod_tiled_supported(int H, int K);
// returns the number of workgroups (rows of `partials`)
template <typename T> int64_t launch_ccmod_grad_tiled(hipStream_t st, const CcmodTiledArgs<T> &a);
// out[i] = sum_g part[g * n + i]
template <typename T>
void launch_sum_groups(hipStream_t st, const cx<T> *part, cx<T> *out, int64_t n, int G);
// (K > 64: returns tiles * slabs rows of partials, of which only the residual sums [0] and [4] are
// filled; the objective sums come from launch_pgm_stats_slabs)
template <typename T> int64_t launch_pgm_fft_momentum(hipStream_t st, const PgmColsArgs<T> &a);
// K > 64, after fft_momentum with want_stats: e_x = sum_slab qpart - Sf per frequency;
// partials2[tile][0..2] = pw |e_x|^2, |e_x|^2, Re<e_x - e_y, e_y> (0 without a.ey).  Returns the
// number of rows (tiles).
template <typename T>
int64_t launch_pgm_stats_slabs(hipStream_t st, const PgmColsArgs<T> &a, double *partials2);

// Mixed-radix heights (csc_fused.h fused_mr_height: H = 16 x {10 ... 30}; csc_pgm_mr.hip): the three
// column kernels above with that many rows per thread, K <= 64.
// plain: the forward transform alone (launch_cols_fft).
int64_t launch_pgm_grad_ifft_mr(hipStream_t st, const PgmColsArgs<float> &a);
int64_t launch_pgm_fft_momentum_mr(hipStream_t st, const PgmColsArgs<float> &a, bool plain);
int64_t launch_ccmod_grad_tiled_mr(hipStream_t st, const CcmodTiledArgs<float> &a);

}  // namespace sporco_amd
