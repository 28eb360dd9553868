// fft.h -- batched strided line FFTs for the (H, W, P) layout (P = C*N*K fastest).
//
// Replaces the reference's sporco.fft.rfftn / irfftn over axes (0, 1)
// (sporco/fft.py:257-314; numpy fallback :631-639).  The transform axes are
// the SLOWEST axes of the array, so every kernel here works on "lines with a
// contiguous batch": a workgroup owns all n points of `cols` adjacent batch
// columns, which makes each global access a contiguous >=128-byte segment.
#pragma once

#include "common.h"

namespace sporco_amd {

constexpr int kMaxRadixPasses = 24;

// Factorisation + twiddle tables of one transform length (host object).
struct FftPlan {
    int n = 0;
    int nrad = 0;
    int radix[kMaxRadixPasses] = {0};       // Stockham passes (fft_c2c / fft_r2c / fft_c2r): 8, 4, 2, 3, 5, 7, primes
    // passes of the in-place transform of fft_cols_sm: every 3 with a 4 or a 2 (radix 12, 6),
    // every 5 with a 2 (radix 10) -- prime-factor butterflies, one LDS round trip for two factors
    int nrad_ip = 0;
    int radix_ip[kMaxRadixPasses] = {0};
    cx<float> *tw32 = nullptr;   // device, W_n^t = exp(-2 pi i t / n), t in [0, n)
    cx<double> *tw64 = nullptr;  // device
    // device, n ints: the frequency held at position pos after the in-place decimation-in-
    // frequency passes radix_ip[0], radix_ip[1], ... (fft_cols_sm)
    int *drev = nullptr;
    void init(int n_);
    void destroy();
    template <typename T> const cx<T> *tw() const;
};

// Complex -> complex lines.  Element (o, i, c) of the input lives at
// in[o*in_outer + i*in_line + c] (units: complex elements), likewise for out.
template <typename T>
void fft_c2c(hipStream_t st, const FftPlan &plan, bool inverse, const cx<T> *in, cx<T> *out,
             int64_t n_outer, int64_t ncols, int64_t in_outer, int64_t in_line,
             int64_t out_outer, int64_t out_line, T scale);

// For both real transforms: with grp > 0 the complex-side column p is addressed as
// (p / grp) * grp_stride + p % grp (the tile-major layout of csc_fused.h).
// Real -> half-spectrum lines (forward).  Input element (o, i, p) at
// in[o*in_outer + i*in_line + p] (units: reals), p in [0, P).  If in2 != null
// the transformed signal is in - s2*in2 (fuses `YU = Y - U`,
// sporco/admm/cbpdn.py:271).  Output (o, f, p), f in [0, n/2], at
// out[o*out_outer + f*out_line + p] (units: complex).
// vf (optional): `in` is the ADMM iterate as the single array V = AX + U (csc_rows.h) and in2
// is null: the line transformed is Y - s2 U with Y = prox_l1(V; thr) (+ NonNegCoef), U = V - Y.
template <typename T> struct VformIn {
    T thr;
    bool nonneg;
};
template <typename T>
void fft_r2c(hipStream_t st, const FftPlan &plan, const T *in, const T *in2, T s2, cx<T> *out,
             int64_t n_outer, int64_t P, int64_t in_outer, int64_t in_line, int64_t out_outer,
             int64_t out_line, int64_t grp = 0, int64_t grp_stride = 0, int64_t bc_mod = 0,
             const VformIn<T> *vf = nullptr);
// (bc_mod > 0: `in` is (n_outer, n, bc_mod) and is broadcast over the P / bc_mod column blocks)

// Half-spectrum -> real lines (inverse, unnormalised times `scale`).  The
// imaginary parts of the DC (and Nyquist, n even) bins are ignored, as
// numpy.fft.irfft / FFTW c2r do.
template <typename T>
void fft_c2r(hipStream_t st, const FftPlan &plan, const cx<T> *in, T *out, int64_t n_outer,
             int64_t P, int64_t in_outer, int64_t in_line, int64_t out_outer, int64_t out_line,
             T scale, int64_t grp = 0, int64_t grp_stride = 0);

// The same pass fused with the ADMM epilogue of the single-array state (csc_kernels.h PostParams with
// v_in / v_out; csc_post_elem.h: relax_AX, ystep, ustep and the residual / objective sums of
// admm.py:877-885, cbpdn.py:614-620, admm.py:434-486; plain l1 term): X of an (n_outer, n, P) array is
// formed in the workgroup and never stored, V is read and V' written, every workgroup writes 8 partial
// sums (returns how many; fft_c2r_vpost_blocks: the same number, for sizing `partials`) -- and, when
// emit_out is set, Y' - U' is transformed forward again and stored there in the layout fft_r2c writes:
// the next iteration's row spectrum for an unchanged rho.  Four X-sized passes (three without the
// emission) in place of the c2r pass, the epilogue kernel and the next r2c pass (seven).  P even.
template <typename T> struct PostParams;
template <typename T> bool fft_c2r_vpost_supported(int64_t P);
template <typename T> int64_t fft_c2r_vpost_blocks(const FftPlan &plan, int64_t n_outer, int64_t P);
template <typename T>
int64_t fft_c2r_vpost(hipStream_t st, const FftPlan &plan, const cx<T> *in, int64_t n_outer, int64_t P,
                      int64_t in_outer, int64_t in_line, T scale, const PostParams<T> &post, cx<T> *emit_out,
                      int64_t emit_outer, int64_t emit_line, double *partials);

// out(H, W/2+1, P) = rfftn(in [- s2*in2], axes=(0,1)) for real in(H, W, P).
template <typename T>
void rfft2(hipStream_t st, const FftPlan &planW, const FftPlan &planH, const T *in, const T *in2,
           T s2, cx<T> *out, int H, int W, int64_t P);

// out(H, W, P) = irfftn(in(H, W/2+1, P), s=(H, W)); `tmp` receives the
// column-transformed spectrum (may alias `in` for an in-place column pass).
template <typename T>
void irfft2(hipStream_t st, const FftPlan &planW, const FftPlan &planH, const cx<T> *in,
            cx<T> *tmp, T *out, int H, int W, int64_t P);

// The column pass of the generic ADMM X-step in one kernel: forward transform along H, the
// Sherman-Morrison solve (sporco/linalg.py:232-297) and the inverse transform of one (wf, cn)
// tile of n = plan.n frequencies x K filters held in LDS.  xf (n, Wf, CN, K) is transformed in
// place (the c2r row pass is what remains of irfftn); partials (Wf * CN doubles) receive the
// Parseval-weighted sums of |Df.xf - Sf|^2 when want_obj.  fft_cols_sm_supported: radices <= 8 and
// K <= 64 with the tile in LDS, or K <= 256 in slabs of filters that fit (four passes then).  Returns the number of tiles.
// (force_slab > 0: the test switch SPORCO_AMD_COLS_SM_FORCE_SLAB, read once per handle -- slabs of
// that many filters even where the tile fits)
template <typename T> bool fft_cols_sm_supported(const FftPlan &plan, int K, int force_slab = 0);
template <typename T>
int64_t fft_cols_sm(hipStream_t st, const FftPlan &plan, cx<T> *xf, const cx<T> *df, const cx<T> *sf,
                    const T *gram, T rho, int Wf, int CN, int K, int W, bool want_obj, double *partials,
                    int force_slab = 0);

}  // namespace sporco_amd
