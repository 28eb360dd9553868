// csc_kernels.h -- launchers of the non-FFT kernels of the ConvBPDN path.
#pragma once

#include "common.h"

namespace sporco_amd {

constexpr int kMaxPartialBlocks = 4096;  // upper bound on reduction-producing grids
constexpr int kOutSlots = 16;            // == SPORCO_AMD_OUT_COUNT

// Broadcastable weight array over (H, W, C, N, K): stride 0 on broadcast axes.
template <typename T> struct Weight {
    const T *ptr = nullptr;  // nullptr => scalar 1
    int64_t stride[5] = {0, 0, 0, 0, 0};
};

struct Dims5 {
    int H, W, C, N, K;
};

// Flags shared with the C ABI (SPORCO_AMD_FLAG_*).
enum : uint32_t {
    F_NONNEG = 1u << 0,
    F_NOBNDRY = 1u << 1,
    F_JOINT = 1u << 2,
    F_RESID = 1u << 3,
    F_OBJ = 1u << 4,
    F_XRRS = 1u << 5,
    F_GEVAL_Y = 1u << 6,
    F_FEVAL_Y = 1u << 7,
    F_KEEP_X = 1u << 8,
    F_NO_X = 1u << 9,
    F_GRADREG = 1u << 10,
    F_AMS = 1u << 11,
    F_DMASK = 1u << 12,
};

// Gradient penalty of ConvBPDNGradReg (cbpdn.py:1133-1143): GHGf[h, wf] = ghh[h] + ghw[wf]
// (signal.gradient_filters, signal.py:204-240, is separable), per-filter weights wg
// (GradWeight; nullptr => 1) and the regularisation parameter mu.
template <typename T> struct GradTerm {
    const T *ghh = nullptr;  // H values   2 - 2 cos(2 pi h / H)
    const T *ghw = nullptr;  // Wf values  2 - 2 cos(2 pi wf / W)
    const T *wg = nullptr;   // K values or nullptr
    T mu = T(0);
};

// dst(H, W, K) = zero-padded src(dH, dW, K)              (cnvrep.zpad, cnvrep.py:704-726)
template <typename T>
void launch_pad_dict(hipStream_t st, const T *src, T *dst, int H, int W, int K, int dH, int dW,
                     int Ksrc = -1);   // Ksrc < K: src holds Ksrc filters, the rest are zero

// gram[pix] = sum_k |df[pix, k]|^2   (the a^H a term of linalg.solvedbi_sm_c, linalg.py:297)
template <typename T>
void launch_gram(hipStream_t st, const cx<T> *df, T *gram, int64_t npix, int K);

// Sherman-Morrison X-step solve in the DFT domain (linalg.solvedbi_sm, linalg.py:232-297)
// with b = conj(Df) Sf + rho * yuf formed on the fly (cbpdn.py:273):
//     xf = yuf + conj(Df) * (Sf - sum_k Df*yuf) / (gram + rho)
// partial sums (per block, 4 doubles): Parseval-weighted |Df.xf - Sf|^2, and the
// LinSolveCheck triple |ax-b|^2, |ax|^2, |b|^2 (cbpdn.py:283-291).  Returns the
// number of blocks that wrote partials.
// With `grad` the system diagonal is mu wg GHGf + rho (linalg.solvedbd_sm, linalg.py:300-366,
// as called by ConvBPDNGradReg.xstep, cbpdn.py:1163-1175); partials are then 5 doubles per
// block, the fifth being the Parseval-weighted sum of wg GHGf |xf|^2 (cbpdn.py:1204-1214).
template <typename T>
int launch_sm_solve(hipStream_t st, const cx<T> *yuf, cx<T> *xf, const cx<T> *df,
                    const cx<T> *sf, const T *gram, T rho, int64_t npix, int CN, int K, int W,
                    bool want_obj, bool want_xrrs, double *partials,
                    const GradTerm<T> *grad = nullptr, int per_grp = 0);
// (per_grp = d > 0: df is (npix, CN / d, K) -- one rank-one term per d consecutive systems of a
// pixel: d = 1 the consensus dictionary update's per-image systems, admm/ccmod.py:766-778;
// d = Cd the same with a multi-channel dictionary, whose channels share the image's matrix)
// partial[block] = Parseval-weighted sum of wg GHGf |vf|^2 (RegGrad at an arbitrary spectrum)
template <typename T>
int launch_grad_norm(hipStream_t st, const cx<T> *vf, const GradTerm<T> &g, int64_t npix, int CN,
                     int K, int W, double *partials);

// out[pix, cn] = sum_k df[pix, k] * v[pix, cn, k]   (linalg.inner over axisM, linalg.py:41-88)
template <typename T>
void launch_inner(hipStream_t st, const cx<T> *df, const cx<T> *v, cx<T> *out, int64_t npix,
                  int CN, int K);

// partial[block][0] = sum wgt(wf) |ef - sf|^2 with half-spectrum weights
// (fft.rfl2norm2, fft.py:449-484); sf may be null.  Returns #blocks.
template <typename T>
int launch_rfl2norm2(hipStream_t st, const cx<T> *ef, const cx<T> *sf, int64_t npix, int64_t cols,
                     int W, double *partials);

// ADMM epilogue: relax_AX + ystep + ustep + residual/objective sums in one pass
// (admm.py:877-885, cbpdn.py:614-620 / :785-794, :297-311, admm.py:434-437, :462-486).
// per-block partials (8 doubles): r2, s2, ax2, y2, u2, l1, l21, unused.
template <typename T> struct PostParams {
    const T *x;
    T *y;
    T *u;
    T rlx, thr, thr21, u_scale;
    uint32_t flags;
    Dims5 d;
    int dH, dW;
    Weight<T> wl1, wl21;
    // AddMaskSim (cbpdn.py:2287-2485): when ams.ptr is set, the last filter is the appended
    // impulse; its slice of Y is (AX + U) zeroed where the mask (H, W, C, N, 1) is nonzero,
    // and it does not enter the l1 / l2,1 sums.
    Weight<T> ams;
    int ams_k = -1;   // index of that filter (K - 1 unless the handle pads the filter axis)
    int ams_n = 1;    // a multi-channel dictionary gets one impulse per channel (cbpdn.py:2339-2346):
                      // filters ams_k .. ams_k + ams_n - 1, the mask then (H, W, 1, N, ams_n)
    // Single-array state (csc_rows.h), plain l1 term only: v_in, when set, is the iterate as
    // V = AX + U, from which Y = prox_l1(V; thr_prev) (+ NonNegCoef) and U = V - Y are derived in
    // place of the y / u loads; v_out, when set, receives V' = AX' + U in place of the y / u
    // stores (v_out may alias v_in, or u)
    const T *v_in = nullptr;
    T *v_out = nullptr;
    T thr_prev = T(0);
};
template <typename T> int launch_admm_post(hipStream_t st, const PostParams<T> &p, double *partials);

// Complex-valued signals / dictionaries as channel pairs of the real machinery (ck_admm.hip
// sm_cplx_kernel): the X-step solve in place on xf (npix, 2 Cc N, K) with the half spectra of the
// real and imaginary parts of the dictionary, and the matching inner product over the filters.
template <typename T>
int launch_sm_cplx(hipStream_t st, cx<T> *xf, const cx<T> *dfa, const cx<T> *dfb, const cx<T> *sf,
                   int64_t npix, int Cc, int N, int K, T rho, int W, bool want_obj, double *partials);
template <typename T>
void launch_inner_cplx(hipStream_t st, const cx<T> *dfa, const cx<T> *dfb, const cx<T> *vf, cx<T> *out,
                       int64_t npix, int Cc, int N, int K);

// Staged pieces (each mirrors one overridable method of the reference class).
template <typename T>
void launch_relax(hipStream_t st, const T *x, const T *y, T *ax, T rlx, int64_t n);   // relax_AX
// the (Y, U) pair of an iterate kept as V = AX + U (csc_rows.h): Y = prox_l1(V; thr) (+ NonNeg),
// U = V - Y; y or u may be null, u may alias v
template <typename T>
void launch_vform_split(hipStream_t st, const T *v, T *y, T *u, T thr, bool nonneg, int64_t n);
// ... under an L1Weight array / NoBndryCross / AddMaskSim (flags: F_NONNEG | F_NOBNDRY; ams: the
// mask of the impulse slice ams_k, or a null Weight)
template <typename T>
void launch_vform_split_general(hipStream_t st, const T *v, T *y, T *u, T thr, uint32_t flags,
                                Dims5 d, int dH, int dW, Weight<T> wl1, Weight<T> ams, int ams_k);
// ... of ConvBPDNJoint: Y = prox_sl1l2(V; thr, thr21) over the C <= 4 channels (arrays
// (npixel, C, NK), NK = N * K)
template <typename T>
void launch_vform_split_joint(hipStream_t st, const T *v, T *y, T *u, T thr, T thr21, bool nonneg,
                              int C, int64_t NK, int64_t npixel);
template <typename T>
void launch_ystep(hipStream_t st, const T *ax, const T *u, T *y, T thr, T thr21, T u_scale,
                  uint32_t flags, Dims5 d, int dH, int dW, Weight<T> wl1, Weight<T> wl21,
                  Weight<T> ams = Weight<T>(), int ams_k = -1, int ams_n = 1);
template <typename T>
void launch_ustep(hipStream_t st, const T *ax, const T *y, T *u, T u_scale, int64_t n);  // ustep
// residual/objective sums of the staged path: x, ax unused for relaxed r (r uses x = AXnr)
template <typename T>
int launch_admm_stats(hipStream_t st, const T *x, const T *y, const T *yprev, const T *u,
                      uint32_t flags, Dims5 d, Weight<T> wl1, Weight<T> wl21, int ams_k,
                      double *partials, int ams_n = 1);
template <typename T> void launch_scale(hipStream_t st, T *v, T s, int64_t n);

// out = soft(v, thr * w) (+ NonNeg / NoBndryCross), l1 partial = sum |w * out|
// (prox_g of pgm/cbpdn.py:288-300; also the prox_l1 primitive).  Returns #blocks.
template <typename T>
int launch_prox_l1(hipStream_t st, const T *v, T *out, T thr, uint32_t flags, Dims5 d, int dH,
                   int dW, Weight<T> wl1, double *partials);
// prox_sl1l2 primitive on (outer, C, inner) (prox/_l21.py:51-88)
template <typename T>
void launch_prox_sl1l2(hipStream_t st, const T *v, T *out, T alpha, T beta, int64_t outer, int C,
                       int64_t inner);

// PGM gradient: gf = conj(df) * (sum_k df*v - sf)  (pgm/cbpdn.py:263-286);
// partial[0] = sum |sum_k df*v - sf|^2 (unweighted), partial[1] = Parseval-weighted.
template <typename T>
int launch_pgm_grad(hipStream_t st, const cx<T> *v, const cx<T> *df, const cx<T> *sf, cx<T> *gf,
                    int64_t npix, int CN, int K, int W, double *partials);
// vf = yf - gf / L
template <typename T>
void launch_axpy_c(hipStream_t st, const cx<T> *y, const cx<T> *g, cx<T> *out, T a, int64_t n);
// dst = a*va + b*vb + c*vc (vb, vc may be null; dst may alias an operand)
template <typename T>
void launch_lincomb(hipStream_t st, cx<T> *dst, T a, const cx<T> *va, T b, const cx<T> *vb, T c,
                    const cx<T> *vc, int64_t n);
// complex pair statistics over (npix, cols) arrays with half-spectrum weights where noted:
//   partial[0] = rfl2norm2-weighted sum |a - b|^2, partial[1] = sum Re(conj(a-b) g),
//   partial[2] = sum |a-b|^2, partial[3] = sum |g|^2
template <typename T>
int launch_pair_stats(hipStream_t st, const cx<T> *a, const cx<T> *b, const cx<T> *g, int64_t npix,
                      int64_t cols, int W, double *partials, int64_t wf_div = 0);

// Dictionary-update gradient (pgm/ccmod.py:295-317), both contractions in one pass
// over the coefficient spectra zf(npix, CN, K):
//     r[n]  = sum_k zf[n,k] d[k] - sf[n]           (inner over filters, axisM)
//     gf[k] = sum_n conj(zf[n,k]) r[n]             (inner over images, axisK; C folded in)
// gf may be null (evaluation only).  partials per block (3 doubles): sum |r|^2,
// Parseval-weighted sum |r|^2, sum |r + sf|^2 (= <d, Hess d>).  Returns #blocks.
template <typename T>
int launch_ccmod_grad(hipStream_t st, const cx<T> *zf, const cx<T> *d, const cx<T> *sf, cx<T> *gf,
                      int64_t npix, int CN, int K, int W, double *partials, int Cd = 1, int zch = 0);
// (Cd > 1: d, gf (npix, Cd, K), sf (npix, Cd, CN): one least-squares problem per channel, sharing
// zf -- or, zch, with coefficient maps of its own: zf (npix, CN, Cd, K))

// Constraint-set projection Pcn = normalise(zeromean(zpad(bcrop(v)))) of a
// dictionary v(H, W, K) with filter support (dH, dW) (cnvrep.py:868-913).
// stats[2k] = mean over support (0 unless zm), stats[2k+1] = 1/norm (1 if norm is 0).
// Multi-scale dictionary (dsz a tuple of size blocks, cnvrep.py:634-662, :778-812): per-filter
// support sizes in device memory (K ints each), or null pointers for the one support (dH, dW).
struct FilterSizes {
    const int *h = nullptr, *w = nullptr;
    // a volume handle (csc_api.hip): rows are (depth slab, height) folded, Hs rows per slab, and
    // the support spans the first dD slabs -- row r lies inside iff r / Hs < dD and r % Hs < dH
    int Hs = 0, dD = 1;
};
template <typename T>
void launch_pcn_stats(hipStream_t st, const T *v, T *stats, int H, int W, int K, int dH, int dW,
                      bool zm, int Cd = 1,    // Cd > 1: v (H, W, Cd, K), stats sized 2 Cd K
                      FilterSizes fs = FilterSizes());
// out = projected v (out may be null: measure only); partial[block] = sum (P(v) - v)^2
template <typename T>
int launch_pcn_apply(hipStream_t st, const T *v, const T *stats, T *out, int H, int W, int K,
                     int dH, int dW, double *partials, int Kvalid = -1,   // k >= Kvalid -> 0
                     int Cd = 1, FilterSizes fs = FilterSizes());
// partial[block] = sum |v|
template <typename T> int launch_asum(hipStream_t st, const T *v, int64_t n, double *partials);

// max |conj(df) * sf| over (npix, CN, K)   (cbpdn.py:573-578); partial[block][0] = block max
template <typename T>
int launch_dhs_absmax(hipStream_t st, const cx<T> *df, const cx<T> *sf, int64_t npix, int CN,
                      int K, double *partials);

// Masked data fidelity (pgm.cbpdn.ConvBPDNMask, pgm.ccmod.ConvCnstrMODMask): r (H, W, C, N) real
// <- w r or w^2 r in place, partial[block] = sum (w r)^2 of the incoming r.  Returns #blocks.
template <typename T>
int launch_mask_apply(hipStream_t st, T *r, const Weight<T> &w, bool squared, int H, int W, int C,
                      int N, double *partials);
// ConvBPDNMaskDcpl (cbpdn.py:2066-2283), the signal-sized block of the two-block constraint:
// out = y0 - us u0 + s   (the block-0 part of the X-step right-hand side, :1615-1616)
template <typename T>
void launch_md_pre(hipStream_t st, const T *y0, const T *u0, const T *s, T *out, T us, int64_t n);
// relax_AX / ystep / ustep of block 0 (:1664-1677, :2236-2241, admm.py:434-437) given
// ax0nr = D x.  partials (5): |ax0nr - y0 - s|^2, |ax0nr|^2, |y0|^2, |u0|^2 (all new values) and
// |w g0|^2 with g0 = y0 (geval_y) or ax0nr - s (:2251-2270).
template <typename T> struct MdY0Args {
    const T *ax0nr;
    T *y0;
    T *u0;
    const T *s;
    Weight<T> w;
    T rho, rlx, us;
    int geval_y;
    int H, W, C, N;
};
template <typename T> int launch_md_y0step(hipStream_t st, const MdY0Args<T> &a, double *partials);
// gf[pix, cn, k] = conj(df[pix, k]) r[pix, cn]
template <typename T>
void launch_conj_outer(hipStream_t st, const cx<T> *df, const cx<T> *r, cx<T> *gf, int64_t npix,
                       int CN, int K);
// gf[pix, k] = sum_n conj(zf[pix, n, k]) r[pix, n]
template <typename T>   // multi-channel dictionary: gf[pix, c, k] = sum_n conj(zf[pix, n, k]) r[pix, c, n]
void launch_mc_zf_adjoint(hipStream_t st, const cx<T> *zf, const cx<T> *r, cx<T> *gf, int64_t npix,
                          int Cd, int N, int K, int zch = 0);   // (zch: zf[pix, n, c, k])
template <typename T>
void launch_zf_adjoint(hipStream_t st, const cx<T> *zf, const cx<T> *r, cx<T> *gf, int64_t npix,
                       int CN, int K);

// ADMM consensus dictionary update (admm/ccmod.py:605-908, admm/admm.py:1441-1707): per-image
// dictionary copies x, duals u (npixr, CN, K) with npixr = H * W, consensus y (npixr, K).
template <typename T>   // out = y - s u
void launch_cns_yu(hipStream_t st, const T *y, const T *u, T *out, T s, int64_t npixr, int CN,
                   int K);
template <typename T>   // m = mean_n(a x + (1 - a) y + s u)
void launch_cns_mean(hipStream_t st, const T *x, const T *u, const T *y, T *m, T a, T s,
                     int64_t npixr, int CN, int K);
template <typename T>   // u = s u + a x + (1 - a) yold - ynew; partials (4): |x - ynew|^2, |x|^2, |u|^2
int launch_cns_ustep(hipStream_t st, const T *x, T *u, const T *yold, const T *ynew, T a, T s,
                     int64_t npixr, int CN, int K, double *partials);
template <typename T>   // partials (2): |ynew - yold|^2, |ynew|^2
int launch_cns_ystats(hipStream_t st, const T *yold, const T *ynew, int64_t n, double *partials);
// LinSolveCheck of the consensus update (admm/ccmod.py:783-792): the residual of the per-image
// systems SUMMED over the images, rrs(sum_n (Z_n^H Z_n + rho) x_n, sum_n b_n).
//   rhs  (before the solve): bsum[pix, k] = sum_n (conj(zf) sf + rho yuf)[pix, n, k]
//   fin  (after it):         asum = sum_n (conj(zf) <zf, xf> + rho xf); partials (3) =
//                            |asum - bsum|^2, |asum|^2, |bsum|^2 summed over (pix, k).
// zf, yuf / xf: (npix, CN, K); sf: (npix, CN); bsum: (npix, K).  K <= 256.
// Cd > 1 (multi-channel dictionary): the CN systems of a pixel are (image, channel) pairs,
// channel fastest, sharing the image's zf row (npix, CN / Cd, K); sf (npix, CN / Cd, Cd); bsum
// (npix, Cd, K).  zch: every (image, channel) pair has a zf row of its own, (npix, CN, K).
template <typename T>
void launch_cns_xrrs_rhs(hipStream_t st, const cx<T> *zf, const cx<T> *sf, const cx<T> *yuf, T rho,
                         cx<T> *bsum, int64_t npix, int CN, int K, int Cd = 1, int zch = 0);
template <typename T>
int launch_cns_xrrs_fin(hipStream_t st, const cx<T> *zf, const cx<T> *xf, T rho, const cx<T> *bsum,
                        int64_t npix, int CN, int K, double *partials, int Cd = 1, int zch = 0);
// dst[r, b, a] = src[r, a, b]
template <typename T>
void launch_swap_inner(hipStream_t st, const cx<T> *src, cx<T> *dst, int64_t rows, int A, int B);

// out[tile][h] = sum_k dft[wf][h][k] v[tile][h][k] - sft[tile][h] on the tile-major layout of
// csc_fused.h (v: (Wf CN, H, Ks) rows of K filters; dft (Wf, H, Ks); sft, out (Wf CN, H))
template <typename T>
void launch_tiled_resid(hipStream_t st, const cx<T> *v, const cx<T> *dft, const cx<T> *sft, cx<T> *out,
                        int64_t ntiles, int H, int K, int Ks, int CN);
// sum over (tile, h, k) of pw(wf) |conj(dft[wf][h][k]) u0t[tile][h] + t[tile][h][k]|^2 on the
// tile-major layout of csc_fused.h (rows Ks filters apart; 0 = K): one double per block
template <typename T>
int launch_md_dualres_tiled(hipStream_t st, const cx<T> *t, const cx<T> *dft, const cx<T> *u0t,
                            int64_t ntiles, int H, int K, int Ks, int CN, int W, double *partials);
// sums over tile-major residual spectra for the PGM step-size policies (ck_dict.hip)
template <typename T>
int launch_resid_stats(hipStream_t st, const cx<T> *a, const cx<T> *b, const cx<T> *c, const cx<T> *d,
                       const T *gramt, int64_t ntiles, int H, int CN, double *partials);
// Complex-valued maps, signals and dictionary on a handle with Cd = 2 (the real and the imaginary
// part are the two channels; admm/ccmod.py with complex input, tests/admm/test_ccmod.py:49-140).
// With A, B the half spectra of the two parts, the full spectrum of a + ib is A + iB at a stored
// frequency f and conj(A - iB) at -f: in the variables P = A + iB, M = A - iB every per-frequency
// equation of the dictionary update holds for P and for M separately -- two "channels" again, each
// with its own coefficient maps, which is a layout the update kernels already take (zch).
// Array (npix, mid, 2, inner), in place or not.  mode 0: (A, B) -> (P, M); mode 1: the same times
// s(f) = sqrt(w(f) / 2), w the Parseval weights of the half spectrum, so that UNWEIGHTED sums over
// the stored half are sums over the full spectrum (the CG dot products and residual norms of the
// reference's fftn path); mode 2: the inverse of mode 1.
template <typename T>
void launch_pm_butterfly(hipStream_t st, const cx<T> *src, cx<T> *dst, int64_t npix, int mid, int inner,
                         int W, int mode);
// dst[(pix, c), n, k] = zch ? src[pix, n, c, k] : src[pix, n, k]  (npix Cd "frequencies" of a
// single-channel dictionary update: api_dstep.inc)
template <typename T>
void launch_zf_per_channel(hipStream_t st, const cx<T> *src, cx<T> *dst, int64_t npix, int N, int Cd,
                           int K, int zch);

// Multi-channel dictionary (Cd > 1) X-step, linalg.solvemdbi_ism (linalg.py:370-444):
// gam(npix, Cd, K), del(npix, Cd), mm(npix, Cd, Cd) hold the recursion's gamma / delta and the
// products <ah_c, gamma_l> (functions of Df and rho);
// the solve forms b = sum_c conj(Df) Sf + rho yuf on the fly.  Layouts: df (npix, Cd, K),
// sf (npix, Cd, N), yuf / xf (npix, N, K).  K <= 256, Cd <= 8.  Partials as launch_sm_solve.
template <typename T>
void launch_ism_setup(hipStream_t st, const cx<T> *df, cx<T> *gam, cx<T> *del, cx<T> *mm,
                      int64_t npix, int Cd, int K, T rho, const GradTerm<T> *grad = nullptr,
                      int W = 0);
// (grad: the identity term is the diagonal mu wg GHGf + rho -- ConvBPDNGradReg with a
// multi-channel dictionary, cbpdn.py:1181-1184; pass the same term and W to both calls.  The
// solve then writes 5 partials per block, the fifth as launch_sm_solve's.)
template <typename T>
int launch_ism_solve(hipStream_t st, const cx<T> *yuf, cx<T> *xf, const cx<T> *df,
                     const cx<T> *sf, const cx<T> *gam, const cx<T> *del, const cx<T> *mm, T rho,
                     int64_t npix, int Cd, int N, int K, int W, bool want_obj, bool want_xrrs,
                     double *partials, const GradTerm<T> *grad = nullptr);
// PGM gradient for a multi-channel dictionary (pgm/cbpdn.py:263-279); partials as launch_pgm_grad
template <typename T>
int launch_mc_pgm_grad(hipStream_t st, const cx<T> *v, const cx<T> *df, const cx<T> *sf, cx<T> *gf,
                       int64_t npix, int Cd, int N, int K, int W, double *partials);
// out[pix, c, n] = sum_k df[pix, c, k] v[pix, n, k]   (vch: v[pix, n, c, k])
template <typename T>
void launch_mc_inner(hipStream_t st, const cx<T> *df, const cx<T> *v, cx<T> *out, int64_t npix,
                     int Cd, int N, int K, int vch = 0);
// gf[pix, n, k] (+)= sum_c conj(df[pix, c, k]) r[pix, c, n] (adjoint of launch_mc_inner)
template <typename T>
void launch_mc_conj_outer(hipStream_t st, const cx<T> *df, const cx<T> *r, cx<T> *gf, int64_t npix,
                          int Cd, int N, int K, bool add);
// max |conj(df[pix, c, k]) sf[pix, c, n]|^2; partial[block] = block max
template <typename T>
int launch_mc_dhs_absmax(hipStream_t st, const cx<T> *df, const cx<T> *sf, int64_t npix, int Cd,
                         int N, int K, double *partials);

// out[slot[i]] = scale[i] * sum_b partials[b*stride + i]  (or max when is_max)
void launch_finalize(hipStream_t st, const double *partials, int nblocks, int stride, int nvals,
                     const int *slots, const double *scales, bool is_max, double *out);
// the same for two partial arrays in one launch (sums only)
void launch_finalize2(hipStream_t st, const double *pa, int nblocks_a, int stride_a, int nvals_a,
                      const int *slots_a, const double *scales_a, const double *pb, int nblocks_b,
                      int stride_b, int nvals_b, const int *slots_b, const double *scales_b,
                      double *out);

// ---------------------------------------------------------------------------
// Pre / post-processing around the solvers, on device arrays (sporco/signal.py:244-301,
// sporco/fft.py:376-417): elementwise pieces; the transforms are fft.h's.
// ---------------------------------------------------------------------------
// out(H + 2 npd, W + 2 npd, P) = numpy.pad(in(H, W, P), npd on axes 0 and 1, 'symmetric')
template <typename T>
void launch_sympad(hipStream_t st, const T *in, T *out, int H, int W, int64_t P, int npd);
// spf(Hp, Wp/2+1, P) /= 1 + lmbda (|Gr|^2 + |Gc|^2), the two-tap gradient spectra
// 2 - 2 cos(2 pi f / n) of signal.py:286-289 written down directly
template <typename T>
void launch_tikhonov_divide(hipStream_t st, cx<T> *spf, int Hp, int Wp, int64_t P, double lmbda);
// slp = sp[npd : npd + H, npd : npd + W]; shp = s - slp
template <typename T>
void launch_crop_highpass(hipStream_t st, const T *sp, const T *s, T *slp, T *shp, int H, int W,
                          int64_t P, int npd);
// out(H, W, P) = in(h, w, P) zero-padded (or cropped) to (H, W)
template <typename T>
void launch_zeropad2(hipStream_t st, const T *in, T *out, int h, int w, int H, int W, int64_t P);
// out[pix, i0, i1, i2] = a[pix, ...] * b[pix, ...] over three trailing axes of extents d[0..2],
// element strides sa / sb (0 on an axis the operand broadcasts over)
template <typename T>
void launch_cmul_bcast(hipStream_t st, const cx<T> *a, const cx<T> *b, cx<T> *out, int64_t npix,
                       const int64_t d[3], const int64_t sa[3], const int64_t sb[3], int64_t pa,
                       int64_t pb);
// out[h, w, p] = in[(h + oh) mod H, (w + ow) mod W, p]   (numpy.roll by -origin)
template <typename T>
void launch_roll2(hipStream_t st, const T *in, T *out, int H, int W, int64_t P, int oh, int ow);
// out = a x + b y (y may be null)
template <typename T>
void launch_axpby(hipStream_t st, T a, const T *x, T b, const T *y, T *out, int64_t n);
// Placement probe (api_placement.inc; profiles/r05_placement_notes.md): writes two arrays AT THE
// SAME TIME, 16 bytes per lane, half of the workgroups each -- each array either rewritten in place (every word loaded and stored back:
// contents unchanged) or, `fresh`, filled with zeros.  n16 words of each, taken as windows of win16
// words that start step_a / step_b words apart (win16 = n16, steps 0: the first n16 words).  The
// caller times it: two arrays whose physical memory lies in the same region of the device's HBM
// take the stores at ~4.7 TB/s together, two in different regions at ~6.3 TB/s.
// b is walked from word rot16 on (wrapping); a real kernel's streams are not in step, so the caller
// sweeps rot16 over the array and sums the times.
void launch_place_probe(hipStream_t st, void *a, void *b, int64_t n16, int64_t win16, int64_t step_a,
                        int64_t step_b, bool fresh_a, bool fresh_b, int64_t rot16 = 0);

// ---------------------------------------------------------------------------
// Conjugate gradients with the scalars on the device (the CG dictionary update,
// linalg.solvemdbi_cg -> scipy.sparse.linalg.cg, sporco/linalg.py:570-579): alpha, beta and the
// stopping test live in a control block that the vector updates read, so an iteration is
// nine launches and no host read-back.
// ---------------------------------------------------------------------------
struct CgCtl {
    double rr, rr_prev, pq, atol;
    double alpha, beta;        // as T values (the casts of the host-driven loop)
    int done, it, info, maxit;
    double rr2[2];             // self-serve mode (CgSelf): <r, r> of iteration i in slot i & 1
};
struct CgPinned {              // host-visible progress: tops processed, and the verdict
    volatile int seq, done, it, info;
};
// Self-serve mode of the operator / update kernels: every workgroup starts by summing the partial
// rows of the preceding kernel itself (launch_finalize's order) and derives beta and the
// stopping verdict (operator) or alpha (update) from them; workgroup 0 records the sums, the
// verdict and the progress word.  No scalar-step launches: two launches per CG iteration.
struct CgSelf {
    CgCtl *c = nullptr;
    CgPinned *pin = nullptr;
    double *cgout = nullptr;
    const double *prev = nullptr;   // partial rows to sum (stride 4)
    int prev_nb = 0;
    int iter = 0;                   // CG iteration index
};
void launch_cg_init(hipStream_t st, CgCtl *c, CgPinned *pin, double atol, int maxit);
// phase 0 (top of an iteration): rr = sum_b partials[b][2]; stop when maxit iterations ran
// (info = maxit) or sqrt(rr) < atol (info = 0); else beta = rr / rr_prev (0 in the first).
// phase 1: pq = sum_b partials[b][1]; alpha = rr / pq; rr_prev = rr; ++it.
// Both sum in the fixed order of launch_finalize.  cgout: device double[2] <- (info, it) when done.
template <typename T>
void launch_cg_ctl(hipStream_t st, int phase, const double *partials, int nb, CgCtl *c, CgPinned *pin,
                   double *cgout);
// q = (Z^H Z + rho I) p per frequency, one wave per pixel with the coefficient spectra of all
// images passing through registers once: t_n = sum_k zf[n,k] p[k] (wave reduction), q[k] = sum_n
// conj(zf[n,k]) t_n + rho p[k].  with_update: p <- r + beta p first (ctl->beta; p = r when 0).
// partials[block][1] = sum Re(conj(p) q) (the <p, Ap> of CG).  ctl may be null (plain operator);
// with ctl the launch does nothing once ctl->done.  Returns the number of blocks.
template <typename T>
int launch_cg_op(hipStream_t st, const CgCtl *ctl, bool with_update, const cx<T> *zf,
                 const cx<T> *r, cx<T> *p, cx<T> *q, T rho, int64_t npix, int CN, int K,
                 double *partials, const CgSelf &self = CgSelf());
template <typename T>   // p = r + beta p (p = r when beta = 0); nothing once done
void launch_cg_update_p(hipStream_t st, const CgCtl *c, const cx<T> *r, cx<T> *p, int64_t n);
// x += alpha p; r -= alpha q; partials[block][2] = sum |r_new|^2 (the next iteration's <r, r>, in
// the element order of launch_pair_stats); nothing once done.  alpha from ctl, or `alpha_host`
// when ctl is null.  Returns the number of blocks.
template <typename T>
int launch_cg_update_xr(hipStream_t st, const CgCtl *c, T alpha_host, cx<T> *x, cx<T> *r,
                        const cx<T> *p, const cx<T> *q, int64_t n, double *partials,
                        const CgSelf &self = CgSelf());

// ---------------------------------------------------------------------------
// Device-resident ADMM control (sporco_amd_csc_admm_run): the residuals, tolerances, the
// adaptive penalty parameter and the stopping test of sporco/admm/admm.py:462-486, 549-575,
// 375-377 evaluated by a one-thread kernel at the end of every iteration, in the same
// precisions and order as the host code of sporco_amd/admm/admm.py evaluates them, so that
// a solve driven from here reproduces the host-driven one bit for bit.  The iteration's
// kernels take rho, lambda / rho, the pending U scale and the speculate / skip / stop
// decisions from this block instead of from launch arguments; the host only enqueues.
// ---------------------------------------------------------------------------
struct AdmmCtl {
    // read by the iteration kernels
    float rho_f, thr_f, u_scale_f, thr21_f;
    float thr_prev_f, thr21_prev_f;   // thr_f / thr21_f of the previous iteration (V form: csc_rows.h)
    int skip_fwd;      // T already holds rows_fwd of the current iterate (emitted, rho unchanged)
    int emit;          // this iteration's epilogue also emits the next iteration's T
    int stop;          // stopping test met: every later launch returns at once
    int k, stable_run;
    int emitted;       // the last executed epilogue emitted T (valid while u_scale == 1)
    int is_f32;
    double rho, u_scale;   // (rho: exact image of the solver-precision value)
    // constants of the solve
    double lmbda, abstol, reltol, sqrt_nc, sqrt_nx, tau, mu, xi;
    double mu21;       // l2,1 weight of ConvBPDNJoint (thr21 = mu21 / rho)
    int autorho, period, autoscaling, stdres;
    int need_resid, no_speculation, pad0;
    unsigned long long t0;
};
// One per executed iteration, written to host-visible (pinned) memory; `seq` last.
struct AdmmRecord {
    double sums[16];
    double r, s, epri, edua, rho, u_scale;   // rho / u_scale: the values the iteration ran with
    unsigned long long ticks;                 // 100 MHz device clock at the end of the iteration
    int k, stop, emit, skip_fwd;
    volatile int seq;                         // index in the run + 1
    int pad;
};
struct AdmmCtlInit {
    double rho, u_scale, lmbda, abstol, reltol, sqrt_nc, sqrt_nx, tau, mu, xi, mu21;
    int k, stable_run, emitted, is_f32, autorho, period, autoscaling, stdres, need_resid,
        no_speculation;
    float thr_prev = 0.f, thr21_prev = 0.f;   // thresholds that produced an incoming V-form iterate
};
void launch_admm_ctl_init(hipStream_t st, AdmmCtl *ctl, const AdmmCtlInit &in);
// sums: the 16 output slots of the iteration in device memory (already all-reduced when the
// images are sharded); rec: slot of this iteration in the host-visible ring
void launch_admm_ctl_update(hipStream_t st, AdmmCtl *ctl, const double *sums, AdmmRecord *rec,
                            int index, bool f32);

}  // namespace sporco_amd
