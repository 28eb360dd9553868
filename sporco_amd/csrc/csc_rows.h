// csc_rows.h -- the row (W axis) passes of the fused ADMM iteration.
//
// Together with the column kernel of csc_fused.h they bring one iteration of
// sporco/admm/admm.py:331-367 for ConvBPDN down to three launches and the
// algorithmic ten float32 passes of SURVEY.md 8(d):
//
//   rows_fwd       Y, U                 -> T   rfft along W of Y - s U     (3 passes)
//   fused_cols     T                    -> T   csc_fused.h                 (2 passes)
//   rows_inv_post  T, Y, U [, X]        -> Y, U [, X] + reduction partials (5 passes)
//
// Single-array state ("V form").  The epilogue forms V' = AX + U, Y' = prox(V'), U' = V' - Y':
// the new iterate is a function of V' alone.  Storing V' instead of (Y', U') and re-deriving
// Y, U from V wherever they are read (here and in rows_fwd) takes the iteration from ten
// X-sized passes to seven (six when the epilogue also emits the next row spectra):
//
//   rows_fwd       V                    -> T                               (2 passes)
//   fused_cols     T                    -> T                               (2 passes)
//   rows_inv_post  T, V                 -> V' [, T']                       (3 [4] passes)
//
// with iterates identical, bit for bit, to the (Y, U) form.  The API layer (csc_api.hip)
// switches a handle into this form for runs of several iterations and materialises Y, U
// (and the previous iterate) when anything else needs them.
//
// rows_inv_post fuses irfft along W (the second half of irfftn in
// GenericConvBPDN.xstep, cbpdn.py:281) with relax_AX (admm.py:877-885), the
// l1 shrinkage ystep (cbpdn.py:614-620, :297-311), ustep (admm.py:434-437) and
// the sums of compute_residuals / obfn_reg (admm.py:462-486, cbpdn.py:624-630),
// so X exists only in registers unless the caller asks for it.
//
// Both kernels are register-resident real FFTs of two adjacent filters packed
// into one complex line (lane = filter pair); T is the tile-major half spectrum
// T[wf][cn][h][k] of csc_fused.h.
#pragma once

#include "common.h"
#include "csc_fused.h"
#include "csc_kernels.h"

namespace sporco_amd {

template <typename T> struct RowsFwdArgs {
    const T *y, *u;    // real (H, W, P)
    T s2;              // transform Y - s2 * U
    cx<T> *t;          // out: tile-major (Wf, CN, H, K)
    const cx<T> *twA;  // [NW][32]: exp(-2 pi i w brev5(i) / W)   (rows_twiddles)
    int H, W, CN, K;
    int64_t P;
    // Single-array state (csc_rows.h, "V form"): when set, y and u are not read; the iterate is
    // derived from V = AX + U of the iteration that produced it, Y = prox(V; thr_prev) (+ NonNeg),
    // U = V - Y -- the very operations that produced Y and U from V in the epilogue, so the
    // values are those a stored (Y, U) pair would hold, for one read pass instead of two.
    const T *v = nullptr;
    T thr_prev = T(0);     // lambda / rho of the iteration that produced v (ctl: ctl->thr_prev_f)
    uint32_t flags = 0;    // F_NONNEG (the derivation repeats the clamp); F_JOINT: Y =
                           // prox_sl1l2(V) over the channel axis -- the kernel then tiles as the
                           // joint epilogue does (lane = (channel, filter pair)) and needs C, N
    T thr21_prev = T(0);   // F_JOINT: mu / rho of that iteration (ctl: ctl->thr21_prev_f)
    int C = 1, N = 0;      // channels and images (CN = C * N): F_JOINT, and the options below
    // ... with an L1Weight array, NoBndryCross or AddMaskSim the derivation repeats those too
    // (same fields as RowsPostArgs; flags then carries F_NOBNDRY as well)
    Weight<T> wl1;
    int dH = 1, dW = 1;
    const uint32_t *ams_bits = nullptr;
    int ams_k = -1;
    int y_bcast = 0;   // y is (H, W, K), broadcast over the CN blocks (consensus D-step)
    int Ks = 0;        // row stride of t in filters when it is not K (0: K), see csc_fused.h
    // device-driven solve (csc_kernels.h AdmmCtl): s2 is ctl->u_scale_f, and the launch returns
    // at once when ctl->stop or ctl->skip_fwd is set
    const AdmmCtl *ctl = nullptr;
    int persist = 0;   // set by the launcher: a device-filling grid of workgroups that loop over tiles
    int stagger_groups = 1, stagger_sleeps = 0;
};

template <typename T> struct RowsPostArgs {
    const cx<T> *t;    // in: tile-major column-inverse-transformed solution, unnormalised
    const cx<T> *t_odd = nullptr;   // the column pass stored its output striped (csc_fused.h
                       // FusedColsArgs::out_even / out_odd): t holds the even row-frequency planes,
                       // t_odd the odd ones, plane wf at index wf >> 1 of its buffer
    const cx<T> *twW;  // exp(-2 pi i t / W), t in [0, W)
    const cx<T> *twA;  // rows_twiddles table (only read when t_next is set)
    cx<T> *t_next;     // optional: also emit rfft_W(Y' - U') tile-major, i.e. the next
                       // iteration's rows_fwd for an unchanged rho; may alias t
    int emit_u = 0;    // (Y, U) form only: emit rfft_W(U') instead -- the row spectra the dual
                       // residual of the mask-decoupled iteration needs (api_maskdcpl.inc)
    const T *y, *u;    // in: real (H, W, P)
    T *y_out, *u_out;  // out (may alias y, u: every element is read and written by one thread)
    T *x;              // out (optional, may be null): X = irfftn(Xf)
    // Single-array state ("V form"; every epilogue variant without an X output):
    //   v_out  the epilogue stores V' = AX + U (scaled) instead of Y' and U' -- both are
    //          functions of V' alone (Y' = prox(V'), U' = V' - Y'), so one array carries the
    //          iterate: one write pass instead of two;
    //   v_in   the previous iterate arrives the same way (y, u are not read): Y = prox(V;
    //          thr_prev) (+ NonNeg), U = V - Y, recomputed per element -- one read pass instead
    //          of two.  Same arithmetic as the (Y, U) form, operation by operation.
    const T *v_in = nullptr;
    T *v_out = nullptr;
    T thr_prev = T(0);     // lambda / rho of the iteration that produced v_in (ctl: thr_prev_f)
    T thr21_prev = T(0);   // F_JOINT: mu / rho of that iteration (ctl: thr21_prev_f)
    T scale;           // 1 / (H W)
    T rlx, thr, u_scale;
    T thr21 = T(0);    // F_JOINT: mu / rho, the l2,1 threshold of prox_sl1l2 (cbpdn.py:785-794)
    uint32_t flags;    // F_NONNEG | F_NOBNDRY | F_GEVAL_Y | F_JOINT
    int H, W, C, N, K, dH, dW;
    int64_t P;
    Weight<T> wl1;
    int Ks = 0;        // row stride of t / t_next in filters when it is not K (0: K)
    const uint32_t *ams_bits = nullptr;   // AddMaskSim mask packed by launch_ams_pack, or null
    int ams_k = -1;    // filter index of the impulse slice
    double *partials;  // per tile 8 doubles: r2, s2, ax2, y2, u2, l1, 0, 0
    // device-driven solve: thr, u_scale and whether t_next is emitted come from this block,
    // and the launch returns at once when ctl->stop is set
    const AdmmCtl *ctl = nullptr;
    int persist = 0;   // set by the launcher (see RowsFwdArgs)
    int stagger_groups = 1, stagger_sleeps = 0;   // persistent launch: start-up stagger (csc_fused.h)
};

// Pack an AddMaskSim mask (as PostParams::ams: broadcastable (H, W, C, N, 1), nonzero = masked)
// into one bit per pixel in the order the row kernel reads it: bits[(h * C N + cn) * NW + w],
// bit n1 <-> pixel x = NW n1 + w, NW = W / 32.  H * C * N * NW words.
template <typename T>
void launch_ams_pack(hipStream_t st, const Weight<T> &mask, uint32_t *bits, int H, int W, int C,
                     int N);

// The row pass of the fused PGM (FISTA) iteration, sporco/pgm/pgm.py:800-803 with
// prox_g of sporco/pgm/cbpdn.py:288-300:  X = prox_l1(irfft_W(t_in) / (H W), thr * wl1)
// (+ NonNeg / NoBndryCross), then t_out = rfft_W(X), tile-major.  t_out == null stops
// after writing X (used to materialise X on demand); x may be null.
template <typename T> struct RowsProxArgs {
    const cx<T> *t_in;
    cx<T> *t_out;
    T *x;
    const cx<T> *twA, *twW;
    T scale, thr;
    uint32_t flags;    // F_NONNEG | F_NOBNDRY
    int H, W, C, N, K, dH, dW;
    int64_t P;
    int Ks = 0;        // row stride of t_in / t_out in filters when it is not K (0: K)
    Weight<T> wl1;
    double *partials;  // per tile 1 double: sum |wl1 * X|
    int persist = 0;   // set by the launcher (see RowsFwdArgs)
    int stagger_groups = 1, stagger_sleeps = 0;
};
template <typename T> int64_t launch_rows_inv_prox_fwd(hipStream_t st, const RowsProxArgs<T> &a);

// ConvBPDNJoint in the row epilogue (F_JOINT): a workgroup then owns one row, one image and 32
// filters of ALL C <= 4 channels -- lane = (channel, filter pair), 16 lanes per channel -- so
// that the l2 norm over the channels that prox_sl1l2 needs (prox/_l21.py:51-88 through
// prox_l2, _lp.py:283-290) is a sum over the 16-lane rows of a wave (two permlane swaps).
// Scalar weights, no NoBndryCross / AddMaskSim; K a multiple of 32.  partials[6] = the l2,1 sum.
template <typename T> bool rows_joint_supported(int W, int C, int K);
// Shapes the register-resident row kernels handle (float32, W in {128, 256, 512}, K even).
template <typename T> bool rows_supported(int W, int K);
// Mixed-radix widths (round 6): 320, 384, 448, 480 = 16 waves x 20 / 24 / 28 / 30 points per thread
// (csc_rows_mr.hip).  Plain ConvBPDN options only: scalar weights, no NoBndryCross / AddMaskSim /
// Joint; the launchers refuse anything else and the API layer keeps such calls on the generic chain.
bool rows_mr_width(int W);
void launch_rows_fwd_mr(hipStream_t st, const RowsFwdArgs<float> &a);
int64_t launch_rows_inv_post_mr(hipStream_t st, const RowsPostArgs<float> &a);
int64_t launch_rows_inv_prox_fwd_mr(hipStream_t st, const RowsProxArgs<float> &a);
// Host table for RowsFwdArgs::twA (W entries: [wave][point of the in-register transform]).
template <typename T> void rows_twiddles(int W, cx<T> *twA);
template <typename T> void launch_rows_fwd(hipStream_t st, const RowsFwdArgs<T> &a);
// Returns the number of tiles (= rows of `partials` written).
template <typename T> int64_t launch_rows_inv_post(hipStream_t st, const RowsPostArgs<T> &a);

// ---------------------------------------------------------------------------------------------
// Small problems (a few million coefficients): a run of iterations of the device-driven solve as
// ONE launch.  At 256 x 256, K = 32, N = 1 the three kernels of an iteration take a few
// microseconds each and the iteration is the cost of its launches; here a grid that the device
// holds at once (one workgroup per CU at most) walks the tiles of the three passes -- the same
// device functions, the single-array state, the next row spectra always emitted -- with a
// barrier across the grid between passes, reduces the partial sums in the order of
// finalize_kernel and advances a control block per workgroup with admm_ctl_update_dev: the
// iterates and the per-iteration records are those of the launch-per-pass loop, bit for bit.
// Plain ConvBPDN options (scalar weights, NonNegCoef; no NoBndryCross / AddMaskSim / Joint /
// GradReg), square images of 128 or 256 pixels, K <= 64, float32.
// ---------------------------------------------------------------------------------------------
template <typename T> struct PersistIterArgs {   // what the passes read in iterations of one parity
    RowsFwdArgs<T> fwd;     // (ctl: filled in by the kernel, per workgroup)
    FusedColsArgs<T> cols;
    RowsPostArgs<T> post;   // V form in and out, emitting
};
template <typename T> struct AdmmPersistArgs {
    PersistIterArgs<T> iter[2];    // iteration index0 + i uses iter[(index0 + i) & 1]: the V buffers,
                                   // and the partial-sum buffers, alternate
    PersistIterArgs<T> *blk;       // scratch: 2 * grid copies of the above, one pair per workgroup
    AdmmCtl *ctl_blk;              // scratch: grid control blocks
    AdmmCtl *ctl;                  // the handle's control block: state in, state out
    AdmmRecord *rec;               // host-visible record of iteration index0 (then consecutive)
    int index0, max_iter;          // index of the first iteration of this launch in the run; how many
    unsigned *bar;                 // kPersistBarWords words, zero before the launch: [1] barrier
                                   // generation, [2] gave-up flag (a wait that did not complete:
                                   // results void), [3] iterations executed, [8..15] phase times of
                                   // measurement builds, [16..] arrival counters
    int n_row_tiles, n_col_tiles;  // rows of the two partial-sum buffers (8 and 1 doubles each)
    int want_dfid, want_sums;      // the data-fidelity sum; any sums at all (FastSolve: none)
    double dfid_scale;             // 1 / (H W)
};
constexpr int kPersistBarWords = 160;
template <typename T> bool admm_persist_supported(int H, int W, int K);
// workgroups of the launch (a multiple of 8, at most one per CU)
template <typename T> int admm_persist_grid(int H, int W, int K, int CN);
// ... and whether the device can hold that grid at once (occupancy of the kernel x compute units):
// the grid barrier inside relies on it
template <typename T> bool admm_persist_resident(int H, int W, int K, int CN);
template <typename T> void launch_admm_persist(hipStream_t st, const AdmmPersistArgs<T> &a, int grid);

}  // namespace sporco_amd
