// csc_fused.h -- the fused fast path of the ADMM X-step for gfx950.
//
// The unfused path (fft.hip + csc_kernels.hip) runs the X-step of
// sporco/admm/cbpdn.py:267-281 as five kernels and moves 16 float32 passes over
// an X-sized array per iteration.  The kernels declared here bring that down to
// the algorithmic 10 passes by keeping every intermediate of
//     column FFT -> Sherman-Morrison solve (linalg.py:232-297) -> column IFFT
// in the register file of one workgroup.  They work on a *tile-major* layout of
// the half-spectrum intermediates that only these kernels see:
//
//     T[wf][cn][h][k]      (wf: row frequency 0..W/2, cn: (channel, image),
//                           h: row / column frequency, k: filter, fastest)
//
// so that one (wf, cn) tile -- all H points of all K filters, the unit the
// Sherman-Morrison inner product couples -- is one contiguous block of HBM.
#pragma once

#include "common.h"
#include "csc_kernels.h"

namespace sporco_amd {

// start-up stagger of the persistent column workgroups (csc_fused.hip launch_fused_cols)
constexpr int kColsStaggerGroups = 4, kColsStaggerSleeps = 2;

// out[(b*A + a)*C + c] = in[(a*B + b)*C + c]: (A, B, C) -> (B, A, C).  Used to
// re-lay Df (H, Wf, K), the per-pixel gram (H, Wf) and Sf (H, Wf*CN) tile-major.
// (in_stride / out_stride: elements between consecutive C-runs when they are padded; 0 = C)
template <typename E>
void launch_permute_ab(hipStream_t st, const E *in, E *out, int64_t A, int64_t B, int64_t C,
                       int64_t in_stride = 0, int64_t out_stride = 0);

template <typename T> struct FusedColsArgs {
    cx<T> *t;          // in: row spectra of Y - sU, tile-major; out (in place): column-
                       // inverse-transformed solution, unnormalised
    const cx<T> *dft;  // Df   tile-major [Wf][H][K]
    const cx<T> *sft;  // Sf   tile-major [Wf][CN][H]
    const T *gramt;    // sum_k |Df|^2 [Wf][H]
    const cx<T> *twA;  // H entries: exp(-2 pi i w brev(i) / H), the twiddles between the two
    const cx<T> *twB;  // H entries: exp(-2 pi i (w + NW j) h2 / H)    FFT stages (fused_twiddles)
    T rho;
    int H, W, CN, K;
    double *partials;  // one double per tile: Parseval-weighted sum |Df.xf - Sf|^2
    // ConvBPDNGradReg (g1t != nullptr; K <= 64 kernel only): the system diagonal is
    // mu wg[k] (ghh[f] + ghw[wf]) + rho, g1t[wf][f] = 1 + sum_k |Df|^2 / diagonal
    // (launch_grad_g1 fills it), and partials holds two doubles per tile, the second the
    // Parseval-weighted sum of wg GHGf |xf|^2.
    // A few more than 64 filters (Kv = 64 < K): the kernel owns the first 64 of the K filters
    // of a row; the caller has already folded the inner products of the others
    // into sft (sft = Sf - sum_{k >= Kv} Df yuf) and gramt covers all K, so the multiplier
    // coef = (Sf - sum_k Df yuf) / (gram + rho) the kernel forms is that of the whole system;
    // it is stored to coef_out[tile][f] for the update of the remaining filters.
    int Kv = 0;
    cx<T> *coef_out = nullptr;
    // ... and in that mode the rows of t and dft may be Ks > K filters apart (0: K), so that
    // every row starts on a cache line although K * 8 bytes is not a multiple of one
    int Ks = 0;
    // per_tile: dft is tile-major like t ([Wf][CN][H][K]) and gramt is [Wf][CN][H] -- one rank-one
    // term per (frequency, image), the X-step of the consensus dictionary update
    // (admm/ccmod.py:766-778) with the coefficient spectra in the role of the dictionary.
    int per_tile = 0;
    // persistent launch: workgroup slot s starts (s % stagger_groups) * stagger_sleeps * 8128
    // cycles late, so that the workgroups' memory and arithmetic phases interleave across CUs
    int stagger_groups = 1, stagger_sleeps = 0;   // (set by the launchers: kColsStagger*)
    // device-driven solve (csc_kernels.h AdmmCtl): rho is ctl->rho_f, and the launch returns at
    // once when ctl->stop is set
    const AdmmCtl *ctl = nullptr;
    // Striped output (K <= 64 kernel; both null: the tile is stored where it was loaded).  The
    // tile (wf, cn) goes to out_even / out_odd by the parity of wf, at tile index (wf >> 1) CN + cn
    // of that buffer -- two half-sized spectra that the caller has placed in DIFFERENT regions of
    // the device memory: one array takes streaming stores at ~4.5 TB/s on the MI355X, two in
    // different regions at ~6.3 TB/s together (profiles/r05_placement_notes.md), and this
    // kernel's store phases (half of its time) run at the single-region rate when it rewrites
    // `t` in place.  The reader (csc_rows.h RowsPostArgs::t_odd) takes the planes from the two.
    cx<T> *out_even = nullptr, *out_odd = nullptr;
    const T *g1t = nullptr;
    T *g1t_out = nullptr;
    const T *ghh = nullptr, *ghw = nullptr, *wg = nullptr;
    T mu = T(0);
};
// The two small kernels around the column kernel when it owns only the first Kv filters:
// sft_eff = sft - sum_{k >= Kv} dft t (t column-transformed on those filters), and afterwards
// t[.., k >= Kv] += conj(dft) coef_out.
template <typename T>
void launch_tail_inner(hipStream_t st, const FusedColsArgs<T> &a, const cx<T> *sft, cx<T> *sft_eff);
template <typename T> void launch_tail_update(hipStream_t st, const FusedColsArgs<T> &a);
// out[row] = sum_k |z[row, k]|^2 over rows of K contiguous filters (one wave per row): the
// per_tile gram of a tile-major spectrum.
template <typename T> void launch_gram_rows(hipStream_t st, const cx<T> *z, T *out, int64_t nrows, int K);
// g1t_out[wf][h] from dft, ghh, ghw, wg, mu, rho.
template <typename T> void launch_grad_g1(hipStream_t st, const FusedColsArgs<T> &a);

// 64 < K <= 256 filters (NH = ceil(K/64) slabs, the last one possibly partial): a tile of all K filters does not fit the register file
// of one workgroup, so the X-step column pass runs as two kernels over (tile, 64-filter
// slab) pairs and exchanges only the partial inner products sum_k Df yuf through `qpart`:
//   cols_fwd_partial    t <- FFT_H(t);            qpart[tile][slab][f] = sum_{k in slab} Df yuf
//   cols_sm_apply_inv   q = sum_slab qpart;  xf = yuf + conj(Df)(Sf - q)/(gram + rho);
//                       t <- IFFT_H(xf);  data-fidelity partials (slab 0 only)
// (12 float32 passes per ADMM iteration instead of 10.)  Uses the FusedColsArgs fields plus:
template <typename T> struct FusedSlabArgs {
    FusedColsArgs<T> c;
    cx<T> *qpart;   // (Wf*CN, NH, H) complex
    // the one-launch form (launch_cols_slab_coop): per (tile, slab) flags that the slab's
    // partial sums of launch `coop_seq` are in qpart, and a host-visible error word
    unsigned *coop_flags = nullptr;
    unsigned coop_seq = 0;
    int *coop_err = nullptr;
    // launch_pgm_grad_slabs (the gradient step of the fused FISTA iteration, csc_pgm.h, for K > 64):
    // input spectrum Yf (output: c.t), 1 / L, and optionally e_y per frequency (Wf, CN, H)
    const cx<T> *pgm_yf = nullptr;
    T pgm_inv_L = T(0);
    cx<T> *pgm_ey = nullptr;
};
// Both kernels above as ONE launch of cooperating slab workgroups (csc_fused.hip): two X-sized
// passes instead of four.  Returns the number of tiles.
template <typename T> int64_t launch_cols_slab_coop(hipStream_t st, const FusedSlabArgs<T> &a);
// pgm_grad_ifft of csc_pgm.h for 64 < K <= 256 by the same cooperating slab workgroups:
// t = IFFT_H(Yf - conj(Df)(sum_k Df Yf - Sf) / L), partials[tile] = sum_f |sum_k Df Yf - Sf|^2.
// Uses c.{t, dft, sft, twA, twB, H, W, CN, K, partials}, qpart, the coop fields and the pgm ones.
template <typename T> int64_t launch_pgm_grad_slabs(hipStream_t st, const FusedSlabArgs<T> &a);
template <typename T> bool fused_slabs_supported(int H, int K);
template <typename T> void launch_cols_fwd_partial(hipStream_t st, const FusedSlabArgs<T> &a);
template <typename T> int64_t launch_cols_sm_apply_inv(hipStream_t st, const FusedSlabArgs<T> &a);

// Multi-channel dictionary (Cd = 2..4 channels in D and S, one coefficient channel): the
// column pass with the Cd rank-one terms in registers (csc_fused_mc.hip).  Woodbury form of the
// system of sporco/admm/cbpdn.py:277-279: x = yuf + Df^H B (Sf - Df yuf), B = (rho I + Df Df^H)^-1.
template <typename T> struct FusedMcArgs {
    cx<T> *t;            // in/out as FusedColsArgs::t, tile-major [Wf][N][H][K]
    const cx<T> *dft;    // Df tile-major [Wf][H][Cd][K]
    const cx<T> *sft;    // Sf tile-major [Wf][H][Cd][N]
    const T *bt;         // B  [Wf][H][Cd][Cd] complex (row-major), from launch_mc_binv
    const cx<T> *twA, *twB;   // fused_twiddles
    T rho;
    int H, W, N, K, Cd;
    double *partials;    // one double per tile: Parseval-weighted sum_c |Df.xf - Sf|^2
};
template <typename T> bool fused_mc_supported(int H, int K, int Cd);
template <typename T> int64_t launch_fused_cols_mc(hipStream_t st, const FusedMcArgs<T> &a);
// bt[row] = (rho I + Df Df^H)^-1 for the nrows = Wf * H frequency rows of dft
template <typename T>
void launch_mc_binv(hipStream_t st, const cx<T> *dft, T *bt, int64_t nrows, int Cd, int K, T rho);

// Dual residual of the mask-decoupled X-step: partials[tile] = pw(wf) sum_{f, k} |conj(dft[wf][f][k])
// sft[tile][f] + FFT_H(t)[tile][f][k]|^2 with t the tile-major ROW spectra of u1 and sft the
// tile-major 2-D spectrum of u0 (fields t, dft, sft, twA, H, W, CN, K, Ks, partials); t is only read.
template <typename T> int64_t launch_cols_dualres(hipStream_t st, const FusedColsArgs<T> &a);

// Mixed-radix heights (round 6): H = 320, 384, 448, 480 = 16 waves x 20 / 24 / 28 / 30 rows per thread
// (regfft.h).  The plain column pass only (K <= 64, no gradient term, per-tile operands or stored
// multipliers).
bool fused_mr_height(int H);
// Host tables twA, twB for fused_cols_supported shapes: fused_twiddle_count(H) entries each
// (H, except for the mixed-radix heights).
int fused_twiddle_count(int H);
template <typename T> void fused_twiddles(int H, int K, cx<T> *twA, cx<T> *twB);
// True when the register-resident column kernel handles this shape.
template <typename T> bool fused_cols_supported(int H, int K);
// Number of tiles (= partials written).
template <typename T> int64_t launch_fused_cols(hipStream_t st, const FusedColsArgs<T> &a);

}  // namespace sporco_amd
