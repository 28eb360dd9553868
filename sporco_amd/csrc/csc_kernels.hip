// csc_kernels.hip -- the non-FFT kernels of the ConvBPDN iteration for gfx950.
//
// All of them are HBM-bound streaming kernels over (pixel, C, N, K) arrays with
// the filter index K fastest, so the mapping is always "consecutive lanes ->
// consecutive K", 16 bytes per lane where the shape allows, wave64 shuffles for
// the per-pixel K-length inner products, and double-precision block partials
// (summed in a fixed order by finalize_kernel => run-to-run deterministic).
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


// ---------------------------------------------------------------------------
// dictionary set-up
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads) pad_dict_kernel(const T *__restrict__ src,
                                                            T *__restrict__ dst, int H, int W,
                                                            int K, int dH, int dW, int Ksrc) {
    const int64_t n = (int64_t)H * W * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t pix = i / K;
        const int x = (int)(pix % W), h = (int)(pix / W);
        dst[i] = (h < dH && x < dW && k < Ksrc) ? src[((int64_t)h * dW + x) * Ksrc + k] : T(0);
    }
}

template <typename T>
void launch_pad_dict(hipStream_t st, const T *src, T *dst, int H, int W, int K, int dH, int dW,
                     int Ksrc) {
    const int64_t n = (int64_t)H * W * K;
    hipLaunchKernelGGL((pad_dict_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, src, dst, H,
                       W, K, dH, dW, Ksrc < 0 ? K : Ksrc);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) gram_kernel(const cx<T> *__restrict__ df,
                                                        T *__restrict__ gram, int64_t npix, int K) {
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < npix;
         pix += (int64_t)gridDim.x * blockDim.x) {
        T s = T(0);
        for (int k = 0; k < K; ++k) s += cabs2(df[pix * K + k]);
        gram[pix] = s;
    }
}

template <typename T>
void launch_gram(hipStream_t st, const cx<T> *df, T *gram, int64_t npix, int K) {
    hipLaunchKernelGGL((gram_kernel<T>), dim3(grid_for(npix)), dim3(kThreads), 0, st, df, gram,
                       npix, K);
    SA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------
// Sherman-Morrison solve
// ---------------------------------------------------------------------------
template <typename T> struct SmArgs {
    const cx<T> *yuf;
    cx<T> *xf;
    const cx<T> *df;
    const cx<T> *sf;
    const T *gram;
    T rho;
    int64_t npix;
    int CN, K, Wf, W;
    int want_obj, want_xrrs;
    double *partials;
    GradTerm<T> g;
    int per_grp;   // d > 0: df is (npix, CN / d, K) -- one system matrix per d consecutive systems of
                   // a pixel (d = 1: per (pixel, cn); d = Cd: the consensus update of a multi-channel
                   // dictionary, whose channels share the image's matrix); gram formed in the kernel
};

__device__ __forceinline__ double parseval_weight(int wf, int Wf, int W) {
    // weights 1, 2, ..., 2, (1 if W even else 2) over the half spectrum (fft.py:476-484)
    return (wf == 0 || ((W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
}

// Weighted gradient spectrum w_k * sum_i |G_i|^2 at (pixel, filter): GHGf of cbpdn.py:1141-1143
template <typename T>
__device__ __forceinline__ T grad_gh(const GradTerm<T> &g, int64_t pix, int Wf) {
    return g.ghh[pix / Wf] + g.ghw[pix % Wf];
}
template <typename T> __device__ __forceinline__ T grad_w(const GradTerm<T> &g, int k) {
    return g.wg ? g.wg[k] : T(1);
}

// Fast path: K even and G = K/2 a power of two <= 64.  Each lane owns two
// adjacent filters (one 16-byte access for f32), a group of G lanes owns one
// (pixel, c, n) system, and the K-length inner product is a log2(G)-step
// wave shuffle reduction.
//
// GRAD (ConvBPDNGradReg, cbpdn.py:1163-1175): the system diagonal is
// dd_k = mu w_k GHGf + rho instead of rho (linalg.solvedbd_sm, linalg.py:300-366):
//     coef = (Sf - rho sum_k Df yuf / dd) / (1 + sum_k |Df|^2 / dd)
//     xf   = (rho yuf + conj(Df) coef) / dd,        Df.xf - Sf = -coef
// and partial 4 is the Parseval-weighted sum of w_k GHGf |xf|^2 (obfn_reg, :1204-1214).
template <typename T, bool GRAD>
__global__ void __launch_bounds__(kThreads) sm_solve_wave_kernel(const SmArgs<T> a) {
    constexpr int NA = GRAD ? 5 : 4;
    const int G = a.K >> 1;
    const int64_t total = a.npix * a.CN * G;
    const int64_t total_pad = (total + kWave - 1) / kWave * kWave;
    double acc[NA] = {};
    const T rho = a.rho;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total_pad;
         t += (int64_t)gridDim.x * blockDim.x) {
        const bool valid = t < total;
        const int64_t grp = t / G;
        const int lg = (int)(t - grp * G);
        const int64_t pix = grp / a.CN;
        cxpair<T> yu, d;
        cx<T> s = mk<T>(T(0), T(0));
        T g = T(1);
        yu.a = yu.b = d.a = d.b = s;
        T gwa = T(0), gwb = T(0);
        if (valid) {
            yu = *reinterpret_cast<const cxpair<T> *>(a.yuf + 2 * t);
            d = *reinterpret_cast<const cxpair<T> *>(a.df + (a.per_grp ? grp / a.per_grp : pix) * a.K + 2 * lg);
            s = a.sf[grp];
            if constexpr (GRAD) {
                const T gh = grad_gh(a.g, pix, a.Wf);
                gwa = grad_w(a.g, 2 * lg) * gh;
                gwb = grad_w(a.g, 2 * lg + 1) * gh;
            } else if (!a.per_grp) {
                g = a.gram[pix];
            }
        }
        const T dda = GRAD ? a.g.mu * gwa + rho : rho, ddb = GRAD ? a.g.mu * gwb + rho : rho;
        const T ia = T(1) / dda, ib = T(1) / ddb;
        cx<T> q;
        T gs = T(0);
        if constexpr (GRAD) {
            q = cscale(cmul(d.a, yu.a), ia) + cscale(cmul(d.b, yu.b), ib);
            gs = cabs2(d.a) * ia + cabs2(d.b) * ib;
        } else {
            q = cmul(d.a, yu.a) + cmul(d.b, yu.b);
            // one system matrix per (pixel, cn): its gram is formed here, from the values
            // already loaded (a.gram may be null)
            if (a.per_grp) gs = cabs2(d.a) + cabs2(d.b);
        }
        for (int m = G >> 1; m > 0; m >>= 1) {
            q.re += __shfl_xor(q.re, m, kWave);
            q.im += __shfl_xor(q.im, m, kWave);
            if (GRAD || a.per_grp) gs += __shfl_xor(gs, m, kWave);
        }
        if (!GRAD && a.per_grp) g = gs;
        cx<T> coef;
        cxpair<T> x;
        if constexpr (GRAD) {
            coef = cscale(s - cscale(q, rho), T(1) / (T(1) + gs));
            x.a = cscale(cscale(yu.a, rho) + cmulc(d.a, coef), ia);
            x.b = cscale(cscale(yu.b, rho) + cmulc(d.b, coef), ib);
        } else {
            coef = cscale(s - q, T(1) / (g + rho));
            x.a = yu.a + cmulc(d.a, coef);
            x.b = yu.b + cmulc(d.b, coef);
        }
        if (valid) *reinterpret_cast<cxpair<T> *>(a.xf + 2 * t) = x;
        const double pw = parseval_weight((int)(pix % a.Wf), a.Wf, a.W);
        if (a.want_obj && valid && lg == 0) {
            // Df.xf - Sf = rho (q - Sf) / (gram + rho)   [GRAD: -coef]
            const double e2 = (double)cabs2(coef) * (GRAD ? 1.0 : (double)rho * (double)rho);
            acc[0] += pw * e2;
        }
        if constexpr (GRAD) {
            if (a.want_obj && valid)
                acc[4] += pw * ((double)gwa * (double)cabs2(x.a) + (double)gwb * (double)cabs2(x.b));
        }
        if (a.want_xrrs) {
            cx<T> dx = cmul(d.a, x.a) + cmul(d.b, x.b);
            for (int m = G >> 1; m > 0; m >>= 1) {
                dx.re += __shfl_xor(dx.re, m, kWave);
                dx.im += __shfl_xor(dx.im, m, kWave);
            }
            if (valid) {
                const cx<T> axa = cmulc(d.a, dx) + cscale(x.a, dda);
                const cx<T> axb = cmulc(d.b, dx) + cscale(x.b, ddb);
                const cx<T> ba = cmulc(d.a, s) + cscale(yu.a, rho);
                const cx<T> bb = cmulc(d.b, s) + cscale(yu.b, rho);
                acc[1] += (double)cabs2(axa - ba) + (double)cabs2(axb - bb);
                acc[2] += (double)cabs2(axa) + (double)cabs2(axb);
                acc[3] += (double)cabs2(ba) + (double)cabs2(bb);
            }
        }
    }
    block_sum_store<NA>(acc, dyn_lds<double>(), a.partials + (int64_t)blockIdx.x * NA);
}

// Generic path (any K): one thread per (pixel, c, n) system.
template <typename T, bool GRAD>
__global__ void __launch_bounds__(kThreads) sm_solve_generic_kernel(const SmArgs<T> a) {
    constexpr int NA = GRAD ? 5 : 4;
    const int64_t total = a.npix * a.CN;
    double acc[NA] = {};
    const T rho = a.rho;
    for (int64_t grp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; grp < total;
         grp += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = grp / a.CN;
        const cx<T> *d = a.df + (a.per_grp ? grp / a.per_grp : pix) * a.K;
        const cx<T> *yu = a.yuf + grp * a.K;
        cx<T> *x = a.xf + grp * a.K;
        const cx<T> s = a.sf[grp];
        const T gh = GRAD ? grad_gh(a.g, pix, a.Wf) : T(0);
        auto diag = [&](int k) -> T { return GRAD ? a.g.mu * (grad_w(a.g, k) * gh) + rho : rho; };
        cx<T> q = mk<T>(T(0), T(0));
        T gs = T(0);
        for (int k = 0; k < a.K; ++k) {
            if constexpr (GRAD) {
                const T inv = T(1) / diag(k);
                q = q + cscale(cmul(d[k], yu[k]), inv);
                gs += cabs2(d[k]) * inv;
            } else {
                q = q + cmul(d[k], yu[k]);
                if (a.per_grp) gs += cabs2(d[k]);
            }
        }
        const cx<T> coef = GRAD ? cscale(s - cscale(q, rho), T(1) / (T(1) + gs))
                                : cscale(s - q, T(1) / ((a.per_grp ? gs : a.gram[pix]) + rho));
        const double pw = parseval_weight((int)(pix % a.Wf), a.Wf, a.W);
        if (a.want_obj)
            acc[0] += pw * (double)cabs2(coef) * (GRAD ? 1.0 : (double)rho * (double)rho);
        cx<T> dx = mk<T>(T(0), T(0));
        double b2 = 0.0, rg = 0.0;
        for (int k = 0; k < a.K; ++k) {
            const cx<T> yk = yu[k];
            const cx<T> xk = GRAD ? cscale(cscale(yk, rho) + cmulc(d[k], coef), T(1) / diag(k))
                                  : yk + cmulc(d[k], coef);
            if (a.want_xrrs) {
                dx = dx + cmul(d[k], xk);
                b2 += (double)cabs2(cmulc(d[k], s) + cscale(yk, rho));
            }
            if constexpr (GRAD) rg += (double)(grad_w(a.g, k) * gh) * (double)cabs2(xk);
            x[k] = xk;
        }
        if constexpr (GRAD) {
            if (a.want_obj) acc[4] += pw * rg;
        }
        if (a.want_xrrs) {
            // b = ax + (b - ax):  recompute b from x: rho yu = dd x - conj(d) coef
            double d2 = 0.0, ax2 = 0.0;
            for (int k = 0; k < a.K; ++k) {
                const cx<T> xk = x[k];
                const cx<T> ry = GRAD ? cscale(xk, diag(k)) - cmulc(d[k], coef)
                                      : cscale(xk - cmulc(d[k], coef), rho);
                const cx<T> ax = cmulc(d[k], dx) + cscale(xk, diag(k));
                const cx<T> b = cmulc(d[k], s) + ry;
                d2 += (double)cabs2(ax - b);
                ax2 += (double)cabs2(ax);
            }
            acc[1] += d2;
            acc[2] += ax2;
            acc[3] += b2;
        }
    }
    block_sum_store<NA>(acc, dyn_lds<double>(), a.partials + (int64_t)blockIdx.x * NA);
}

static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

template <typename T>
int launch_sm_solve(hipStream_t st, const cx<T> *yuf, cx<T> *xf, const cx<T> *df,
                    const cx<T> *sf, const T *gram, T rho, int64_t npix, int CN, int K, int W,
                    bool want_obj, bool want_xrrs, double *partials, const GradTerm<T> *grad,
                    int per_grp) {
    SmArgs<T> a;
    a.per_grp = per_grp;
    a.yuf = yuf;
    a.xf = xf;
    a.df = df;
    a.sf = sf;
    a.gram = gram;
    a.rho = rho;
    a.npix = npix;
    a.CN = CN;
    a.K = K;
    a.W = W;
    a.Wf = W / 2 + 1;
    a.want_obj = want_obj;
    a.want_xrrs = want_xrrs;
    a.partials = partials;
    a.g = grad ? *grad : GradTerm<T>();
    const size_t lds = sizeof(double) * 5 * (kThreads / kWave);
    int grid;
    if (K % 2 == 0 && is_pow2(K / 2) && K / 2 <= kWave) {
        grid = grid_for(npix * CN * (K / 2));
        if (grad)
            hipLaunchKernelGGL((sm_solve_wave_kernel<T, true>), dim3(grid), dim3(kThreads), lds, st, a);
        else
            hipLaunchKernelGGL((sm_solve_wave_kernel<T, false>), dim3(grid), dim3(kThreads), lds, st, a);
    } else {
        grid = grid_for(npix * CN);
        if (grad)
            hipLaunchKernelGGL((sm_solve_generic_kernel<T, true>), dim3(grid), dim3(kThreads), lds, st, a);
        else
            hipLaunchKernelGGL((sm_solve_generic_kernel<T, false>), dim3(grid), dim3(kThreads), lds, st, a);
    }
    SA_HIP(hipGetLastError());
    return grid;
}

// partial[block] = Parseval-weighted sum of w_k GHGf |vf|^2 over (npix, CN, K): the
// gradient regulariser evaluated at an arbitrary spectrum (cbpdn.py:1204-1214 with
// fEvalX False)
template <typename T>
__global__ void __launch_bounds__(kThreads) grad_norm_kernel(const cx<T> *__restrict__ vf,
                                                             const GradTerm<T> g, int64_t npix,
                                                             int CN, int K, int W,
                                                             double *__restrict__ partials) {
    const int Wf = W / 2 + 1;
    const int64_t total = npix * CN * K;
    double acc[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t pix = i / ((int64_t)K * CN);
        acc[0] += parseval_weight((int)(pix % Wf), Wf, W) *
                  (double)(grad_w(g, k) * grad_gh(g, pix, Wf)) * (double)cabs2(vf[i]);
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + blockIdx.x);
}

template <typename T>
int launch_grad_norm(hipStream_t st, const cx<T> *vf, const GradTerm<T> &g, int64_t npix, int CN,
                     int K, int W, double *partials) {
    const int grid = grid_for(npix * CN * K);
    hipLaunchKernelGGL((grad_norm_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, vf, g, npix, CN, K, W, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// inner product over filters, half-spectrum norms
// ---------------------------------------------------------------------------
// L lanes share one output (consecutive lanes read consecutive filters; L = 1: a thread per output)
template <typename T, int L>
__global__ void __launch_bounds__(kThreads) inner_kernel(const cx<T> *__restrict__ df,
                                                         const cx<T> *__restrict__ v,
                                                         cx<T> *__restrict__ out, int64_t npix,
                                                         int CN, int K) {
    const int64_t total = npix * CN;
    const int sub = threadIdx.x % L;
    const int64_t per_blk = blockDim.x / L;
    // (every lane takes part in the shuffles: the loop bound is the same for a whole workgroup)
    for (int64_t base = (int64_t)blockIdx.x * per_blk; base < total; base += (int64_t)gridDim.x * per_blk) {
        const int64_t grp = base + threadIdx.x / L;
        cx<T> q = mk<T>(T(0), T(0));
        if (grp < total) {
            const cx<T> *d = df + (grp / CN) * K, *x = v + grp * K;
            for (int k = sub; k < K; k += L) q = q + cmul(d[k], x[k]);
        }
        if (L > 1) {
#pragma unroll
            for (int m = L / 2; m >= 1; m >>= 1) {
                q.re += __shfl_xor(q.re, m, kWave);
                q.im += __shfl_xor(q.im, m, kWave);
            }
        }
        if (grp < total && sub == 0) out[grp] = q;
    }
}

template <typename T>
void launch_inner(hipStream_t st, const cx<T> *df, const cx<T> *v, cx<T> *out, int64_t npix,
                  int CN, int K) {
    if (K >= 16) {
        hipLaunchKernelGGL((inner_kernel<T, 16>), dim3(grid_for(npix * CN * 16)), dim3(kThreads), 0, st, df,
                           v, out, npix, CN, K);
    } else {
        hipLaunchKernelGGL((inner_kernel<T, 1>), dim3(grid_for(npix * CN)), dim3(kThreads), 0, st, df, v,
                           out, npix, CN, K);
    }
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) rfl2norm2_kernel(const cx<T> *__restrict__ ef,
                                                             const cx<T> *__restrict__ sf,
                                                             int64_t npix, int64_t cols, int Wf,
                                                             int W, double *partials) {
    const int64_t total = npix * cols;
    double acc[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / cols;
        cx<T> e = ef[i];
        if (sf) e = e - sf[i];
        acc[0] += parseval_weight((int)(pix % Wf), Wf, W) * (double)cabs2(e);
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + blockIdx.x);
}

template <typename T>
int launch_rfl2norm2(hipStream_t st, const cx<T> *ef, const cx<T> *sf, int64_t npix, int64_t cols,
                     int W, double *partials) {
    const int grid = grid_for(npix * cols);
    hipLaunchKernelGGL((rfl2norm2_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, ef, sf, npix, cols, W / 2 + 1, W,
                       partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// ADMM epilogue (single-pass relax + prox + dual update + all sums)
// ---------------------------------------------------------------------------
// GENERAL = weight arrays and/or NoBndryCross need the 5-D index of every element.
template <typename T, int VEC, bool GENERAL>
__global__ void __launch_bounds__(kThreads) admm_post_kernel(const PostParams<T> p, int64_t nvec,
                                                             double *partials) {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const int64_t P = (int64_t)p.d.C * p.d.N * p.d.K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
         i += (int64_t)gridDim.x * blockDim.x) {
        const Vec<T, VEC> xv = reinterpret_cast<const Vec<T, VEC> *>(p.x)[i];
        Vec<T, VEC> yv = reinterpret_cast<const Vec<T, VEC> *>(p.y)[i];
        Vec<T, VEC> uv = reinterpret_cast<const Vec<T, VEC> *>(p.u)[i];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            admm_post_elem<T, GENERAL>(p, i * VEC + e, P, xv.v[e], yv.v[e], uv.v[e], acc);
        reinterpret_cast<Vec<T, VEC> *>(p.y)[i] = yv;
        reinterpret_cast<Vec<T, VEC> *>(p.u)[i] = uv;
    }
    block_sum_store<8>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 8);
}

// Joint l1 + l2,1 epilogue: one thread per (pixel, n, k), looping over the C
// channels that prox_l2 couples (prox/_lp.py:283-290 over axisC, cbpdn.py:790-793).
template <typename T>
__global__ void __launch_bounds__(kThreads) admm_post_joint_kernel(const PostParams<T> p,
                                                                   double *partials) {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const T a = p.rlx, oma = T(1) - p.rlx;
    const bool nonneg = p.flags & F_NONNEG, nob = p.flags & F_NOBNDRY, gy = p.flags & F_GEVAL_Y;
    const int C = p.d.C;
    const int64_t NK = (int64_t)p.d.N * p.d.K;
    const int64_t total = (int64_t)p.d.H * p.d.W * NK;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / NK;
        const int nk = (int)(t - pix * NK);
        const int k = nk % p.d.K, n = nk / p.d.K;
        const int xw = (int)(pix % p.d.W), h = (int)(pix / p.d.W);
        const int64_t base = pix * C * NK + nk;
        const bool kill = nob && in_bndry(h, xw, p.d.H, p.d.W, p.dH, p.dW);
        const bool ams = p.ams.ptr && is_ams(k, p.ams_k, p.ams_n);   // AddMaskSim slice, see admm_post_kernel
        // pass 1: l2 norm over channels of the soft-thresholded values
        T nrm2 = T(0);
        for (int c = 0; c < C; ++c) {
            const int64_t idx = base + c * NK;
            const T w = p.wl1.ptr ? weight_at(p.wl1, h, xw, c, n, k) : T(1);
            const T ax = a * p.x[idx] + oma * p.y[idx];
            const T sv = soft(ax + p.u_scale * p.u[idx], p.thr * w);
            nrm2 += sv * sv;
        }
        const T nrm = sqrt(nrm2);
        const T w21 = p.wl21.ptr ? weight_at(p.wl21, h, xw, 0, n, k) : T(1);
        T shrink = nrm - p.thr21 * w21;
        shrink = shrink > T(0) ? shrink : T(0);
        const T fac = (nrm != T(0)) ? shrink / nrm : T(0);  // array.zdivide, array.py:119-137
        // pass 2: outputs and sums
        double g2 = 0.0;
        for (int c = 0; c < C; ++c) {
            const int64_t idx = base + c * NK;
            const T w = p.wl1.ptr ? weight_at(p.wl1, h, xw, c, n, k) : T(1);
            const T x = p.x[idx], yo = p.y[idx], uo = p.u_scale * p.u[idx];
            const T ax = a * x + oma * yo;
            T yn = fac * soft(ax + uo, p.thr * w);
            if (nonneg && yn < T(0)) yn = T(0);
            if (kill) yn = T(0);
            if (ams) yn = weight_at(p.ams, h, xw, c, n, k - p.ams_k) != T(0) ? T(0) : ax + uo;
            const T un = uo + ax - yn;
            p.y[idx] = yn;
            p.u[idx] = un;
            const double dr = (double)(x - yn), ds = (double)(yn - yo);
            acc[0] += dr * dr;
            acc[1] += ds * ds;
            acc[2] += (double)x * (double)x;
            acc[3] += (double)yn * (double)yn;
            acc[4] += (double)un * (double)un;
            const T gvar = ams ? T(0) : (gy ? yn : x);
            const T gv = w * gvar;
            acc[5] += (double)(gv < T(0) ? -gv : gv);
            g2 += (double)gvar * (double)gvar;
        }
        acc[6] += (double)w21 * sqrt(g2);
    }
    block_sum_store<8>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 8);
}

// The same epilogue with the C channels of V adjacent filters held in registers, so that
// X, Y and U are read once (5 passes over an X-sized array instead of 8).  C = CC <= 4,
// K % V == 0; one thread per (pixel, n, group of V filters).
template <typename T, int CC, int V>
__global__ void __launch_bounds__(kThreads) admm_post_joint_reg_kernel(const PostParams<T> p,
                                                                       double *partials) {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const T a = p.rlx, oma = T(1) - p.rlx;
    const bool nonneg = p.flags & F_NONNEG, nob = p.flags & F_NOBNDRY, gy = p.flags & F_GEVAL_Y;
    const int KV = p.d.K / V;
    const int64_t NK = (int64_t)p.d.N * p.d.K, NKV = (int64_t)p.d.N * KV;
    const int64_t total = (int64_t)p.d.H * p.d.W * NKV;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / NKV;
        const int nkv = (int)(t - pix * NKV);
        const int k0 = (nkv % KV) * V, n = nkv / KV;
        const int xw = (int)(pix % p.d.W), h = (int)(pix / p.d.W);
        const int64_t base = pix * CC * NK + (int64_t)n * p.d.K + k0;
        const bool kill = nob && in_bndry(h, xw, p.d.H, p.d.W, p.dH, p.dW);
        Vec<T, V> xv[CC], yv[CC], uv[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            xv[c] = *reinterpret_cast<const Vec<T, V> *>(p.x + base + c * NK);
            yv[c] = *reinterpret_cast<const Vec<T, V> *>(p.y + base + c * NK);
            uv[c] = *reinterpret_cast<const Vec<T, V> *>(p.u + base + c * NK);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int k = k0 + e;
            const bool ams = p.ams.ptr && is_ams(k, p.ams_k, p.ams_n);
            T ax[CC], uo[CC], sv[CC], w[CC];
            T nrm2 = T(0);
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                w[c] = p.wl1.ptr ? weight_at(p.wl1, h, xw, c, n, k) : T(1);
                ax[c] = a * xv[c].v[e] + oma * yv[c].v[e];
                uo[c] = p.u_scale * uv[c].v[e];
                sv[c] = soft(ax[c] + uo[c], p.thr * w[c]);
                nrm2 += sv[c] * sv[c];
            }
            const T nrm = sqrt(nrm2);
            const T w21 = p.wl21.ptr ? weight_at(p.wl21, h, xw, 0, n, k) : T(1);
            T shrink = nrm - p.thr21 * w21;
            shrink = shrink > T(0) ? shrink : T(0);
            const T fac = (nrm != T(0)) ? shrink / nrm : T(0);
            double g2 = 0.0;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const T x = xv[c].v[e], yo = yv[c].v[e];
                T yn = fac * sv[c];
                if (nonneg && yn < T(0)) yn = T(0);
                if (kill) yn = T(0);
                if (ams) yn = weight_at(p.ams, h, xw, c, n, k - p.ams_k) != T(0) ? T(0) : ax[c] + uo[c];
                const T un = uo[c] + ax[c] - yn;
                yv[c].v[e] = yn;
                uv[c].v[e] = un;
                const double dr = (double)(x - yn), ds = (double)(yn - yo);
                acc[0] += dr * dr;
                acc[1] += ds * ds;
                acc[2] += (double)x * (double)x;
                acc[3] += (double)yn * (double)yn;
                acc[4] += (double)un * (double)un;
                const T gvar = ams ? T(0) : (gy ? yn : x);
                const T gv = w[c] * gvar;
                acc[5] += (double)(gv < T(0) ? -gv : gv);
                g2 += (double)gvar * (double)gvar;
            }
            acc[6] += (double)w21 * sqrt(g2);
        }
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            *reinterpret_cast<Vec<T, V> *>(p.y + base + c * NK) = yv[c];
            *reinterpret_cast<Vec<T, V> *>(p.u + base + c * NK) = uv[c];
        }
    }
    block_sum_store<8>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 8);
}

template <typename T, int V>
static int launch_post_joint_reg(hipStream_t st, const PostParams<T> &p, double *partials) {
    const int64_t E = (int64_t)p.d.H * p.d.W * p.d.C * p.d.N * p.d.K;
    const size_t lds = sizeof(double) * 8 * (kThreads / kWave);
    const int grid = grid_for(E / p.d.C / V);
    switch (p.d.C) {
    case 1: hipLaunchKernelGGL((admm_post_joint_reg_kernel<T, 1, V>), dim3(grid), dim3(kThreads), lds, st, p, partials); break;
    case 2: hipLaunchKernelGGL((admm_post_joint_reg_kernel<T, 2, V>), dim3(grid), dim3(kThreads), lds, st, p, partials); break;
    case 3: hipLaunchKernelGGL((admm_post_joint_reg_kernel<T, 3, V>), dim3(grid), dim3(kThreads), lds, st, p, partials); break;
    default: hipLaunchKernelGGL((admm_post_joint_reg_kernel<T, 4, V>), dim3(grid), dim3(kThreads), lds, st, p, partials); break;
    }
    return grid;
}

template <typename T> int launch_admm_post(hipStream_t st, const PostParams<T> &p, double *partials) {
    const int64_t E = (int64_t)p.d.H * p.d.W * p.d.C * p.d.N * p.d.K;
    const size_t lds = sizeof(double) * 8 * (kThreads / kWave);
    int grid;
    if (p.flags & F_JOINT) {
        constexpr int VJ = 16 / sizeof(T);
        if (p.d.C <= 4 && p.d.K % VJ == 0) {
            grid = launch_post_joint_reg<T, VJ>(st, p, partials);
        } else if (p.d.C <= 4) {
            grid = launch_post_joint_reg<T, 1>(st, p, partials);
        } else {
            grid = grid_for(E / p.d.C);
            hipLaunchKernelGGL((admm_post_joint_kernel<T>), dim3(grid), dim3(kThreads), lds, st, p,
                               partials);
        }
    } else {
        const bool general = p.wl1.ptr != nullptr || (p.flags & F_NOBNDRY) || p.ams.ptr;
        constexpr int V = 16 / sizeof(T);
        if (E % V == 0) {
            grid = grid_for(E / V);
            if (general)
                hipLaunchKernelGGL((admm_post_kernel<T, V, true>), dim3(grid), dim3(kThreads), lds,
                                   st, p, E / V, partials);
            else
                hipLaunchKernelGGL((admm_post_kernel<T, V, false>), dim3(grid), dim3(kThreads), lds,
                                   st, p, E / V, partials);
        } else {
            grid = grid_for(E);
            if (general)
                hipLaunchKernelGGL((admm_post_kernel<T, 1, true>), dim3(grid), dim3(kThreads), lds,
                                   st, p, E, partials);
            else
                hipLaunchKernelGGL((admm_post_kernel<T, 1, false>), dim3(grid), dim3(kThreads), lds,
                                   st, p, E, partials);
        }
    }
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// staged ADMM pieces
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads) relax_kernel(const T *__restrict__ x,
                                                         const T *__restrict__ y,
                                                         T *__restrict__ ax, T rlx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        ax[i] = rlx * x[i] + (T(1) - rlx) * y[i];
}

template <typename T>
void launch_relax(hipStream_t st, const T *x, const T *y, T *ax, T rlx, int64_t n) {
    hipLaunchKernelGGL((relax_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, x, y, ax, rlx, n);
    SA_HIP(hipGetLastError());
}

// (Y, U) of an iterate kept in the single-array form of csc_rows.h: Y = prox_l1(V; thr)
// (+ NonNeg), U = V - Y, element for element the operations of the row epilogue.  y or u may be
// null; u may alias v (each element is read, then written, by one thread).
template <typename T>
__global__ void __launch_bounds__(kThreads) vform_split_kernel(const T *v, T *y, T *u, T thr,
                                                               int nonneg, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const T vv = v[i];
        T yy = soft(vv, thr);
        if (nonneg && yy < T(0)) yy = T(0);
        if (y) y[i] = yy;
        if (u) u[i] = vv - yy;
    }
}

// The same for ConvBPDNJoint: Y = prox_sl1l2(V; thr, thr21) over the C <= 4 channels of each
// (pixel, image, filter) -- soft threshold, then the channel vector shrunk in l2 norm
// (cbpdn.py:785-794, prox/_l21.py:51-88), with the sum of squares taken in the order the row
// epilogue's cross-lane sum takes it, (s0 + s1) + (s2 + s3).  One thread per (pixel, n, k).
template <typename T>
__global__ void __launch_bounds__(kThreads) vform_split_joint_kernel(const T *v, T *y, T *u, T thr,
                                                                     T thr21, int nonneg, int C,
                                                                     int64_t NK, int64_t npixel) {
#pragma clang fp contract(off)
    const int64_t total = npixel * NK;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / NK, r = i - pix * NK;
        const int64_t base = pix * C * NK + r;
        T vv[4] = {T(0), T(0), T(0), T(0)}, sv[4], sq[4];
        for (int c = 0; c < 4; ++c) {
            if (c < C) vv[c] = v[base + c * NK];
            sv[c] = soft(vv[c], thr);
            sq[c] = sv[c] * sv[c];
        }
        const T q = (sq[0] + sq[1]) + (sq[2] + sq[3]);
        T fac = (T)sa_fma(-(float)thr21, sa_rsq((float)q), 1.f);
        fac = fac > T(0) ? fac : T(0);
        for (int c = 0; c < C; ++c) {
            T yy = fac * sv[c];
            if (nonneg && yy < T(0)) yy = T(0);
            if (y) y[base + c * NK] = yy;
            if (u) u[base + c * NK] = vv[c] - yy;
        }
    }
}

template <typename T>
void launch_vform_split_joint(hipStream_t st, const T *v, T *y, T *u, T thr, T thr21, bool nonneg,
                              int C, int64_t NK, int64_t npixel) {
    SA_REQUIRE(C >= 1 && C <= 4, "the joint V form serves up to four channels");
    hipLaunchKernelGGL((vform_split_joint_kernel<T>), dim3(grid_for(npixel * NK)), dim3(kThreads), 0,
                       st, v, y, u, thr, thr21, nonneg ? 1 : 0, C, NK, npixel);
    SA_HIP(hipGetLastError());
}

// ... under an L1Weight array / NoBndryCross / AddMaskSim: the per-element constants of the row
// epilogue (csc_rows.hip rows_inv_post_tile) -- weight (0 on the AddMaskSim impulse slice, which
// is neither shrunk nor clamped), the boundary band and the mask as multiplicative 0 / 1.
template <typename T> struct VsplitArgs {
    const T *v;
    T *y, *u;
    T thr;
    uint32_t flags;
    Dims5 d;
    int dH, dW;
    Weight<T> wl1, ams;
    int ams_k;
};
template <typename T>
__global__ void __launch_bounds__(kThreads) vform_split_general_kernel(const VsplitArgs<T> p) {
#pragma clang fp contract(off)
    const bool nonneg = p.flags & F_NONNEG, nob = p.flags & F_NOBNDRY;
    const int64_t P = (int64_t)p.d.C * p.d.N * p.d.K;
    const int64_t total = (int64_t)p.d.H * p.d.W * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / P;
        const int r = (int)(i - pix * P);
        const int k = r % p.d.K, n = (r / p.d.K) % p.d.N, c = r / (p.d.K * p.d.N);
        const int xw = (int)(pix % p.d.W), h = (int)(pix / p.d.W);
        const bool am = p.ams.ptr && k == p.ams_k;
        T wt = p.wl1.ptr ? weight_at(p.wl1, h, xw, c, n, k) : T(1);
        if (am) wt = T(0);
        const T keep = (nob && in_bndry(h, xw, p.d.H, p.d.W, p.dH, p.dW)) ? T(0) : T(1);
        const T mkeep = (am && weight_at(p.ams, h, xw, c, n, 0) != T(0)) ? T(0) : T(1);
        const T vv = p.v[i];
        T yy = soft(vv, p.thr * wt);
        if (nonneg && !am && yy < T(0)) yy = T(0);
        yy *= am ? mkeep : keep;
        if (p.y) p.y[i] = yy;
        if (p.u) p.u[i] = vv - yy;
    }
}
template <typename T>
void launch_vform_split_general(hipStream_t st, const T *v, T *y, T *u, T thr, uint32_t flags,
                                Dims5 d, int dH, int dW, Weight<T> wl1, Weight<T> ams, int ams_k) {
    VsplitArgs<T> p;
    p.v = v;
    p.y = y;
    p.u = u;
    p.thr = thr;
    p.flags = flags;
    p.d = d;
    p.dH = dH;
    p.dW = dW;
    p.wl1 = wl1;
    p.ams = ams;
    p.ams_k = ams_k;
    const int64_t total = (int64_t)d.H * d.W * d.C * d.N * d.K;
    hipLaunchKernelGGL((vform_split_general_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, st, p);
    SA_HIP(hipGetLastError());
}

template <typename T>
void launch_vform_split(hipStream_t st, const T *v, T *y, T *u, T thr, bool nonneg, int64_t n) {
    hipLaunchKernelGGL((vform_split_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, v, y, u, thr,
                       nonneg ? 1 : 0, n);
    SA_HIP(hipGetLastError());
}

template <typename T> struct YstepArgs {
    const T *ax;
    const T *u;
    T *y;
    T thr, thr21, u_scale;
    uint32_t flags;
    Dims5 d;
    int dH, dW;
    Weight<T> wl1, wl21, ams;
    int ams_k;
    int ams_n = 1;   // number of impulse filters from ams_k on
};

template <typename T>
__global__ void __launch_bounds__(kThreads) ystep_kernel(const YstepArgs<T> p) {
    const bool nonneg = p.flags & F_NONNEG, nob = p.flags & F_NOBNDRY, joint = p.flags & F_JOINT;
    const int C = p.d.C;
    const int64_t NK = (int64_t)p.d.N * p.d.K;
    const int64_t total = (int64_t)p.d.H * p.d.W * NK;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / NK;
        const int nk = (int)(t - pix * NK);
        const int k = nk % p.d.K, n = nk / p.d.K;
        const int xw = (int)(pix % p.d.W), h = (int)(pix / p.d.W);
        const int64_t base = pix * C * NK + nk;
        const bool kill = nob && in_bndry(h, xw, p.d.H, p.d.W, p.dH, p.dW);
        T fac = T(1);
        if (joint) {
            T nrm2 = T(0);
            for (int c = 0; c < C; ++c) {
                const int64_t idx = base + c * NK;
                const T w = p.wl1.ptr ? weight_at(p.wl1, h, xw, c, n, k) : T(1);
                const T sv = soft(p.ax[idx] + p.u_scale * p.u[idx], p.thr * w);
                nrm2 += sv * sv;
            }
            const T nrm = sqrt(nrm2);
            const T w21 = p.wl21.ptr ? weight_at(p.wl21, h, xw, 0, n, k) : T(1);
            T shrink = nrm - p.thr21 * w21;
            shrink = shrink > T(0) ? shrink : T(0);
            fac = (nrm != T(0)) ? shrink / nrm : T(0);
        }
        for (int c = 0; c < C; ++c) {
            const int64_t idx = base + c * NK;
            const T w = p.wl1.ptr ? weight_at(p.wl1, h, xw, c, n, k) : T(1);
            const T v = p.ax[idx] + p.u_scale * p.u[idx];
            T yn = fac * soft(v, p.thr * w);
            if (nonneg && yn < T(0)) yn = T(0);
            if (kill) yn = T(0);
            if (p.ams.ptr && is_ams(k, p.ams_k, p.ams_n))   // AddMaskSim slice (cbpdn.py:2378-2394)
                yn = weight_at(p.ams, h, xw, c, n, k - p.ams_k) != T(0) ? T(0) : v;
            p.y[idx] = yn;
        }
    }
}

template <typename T>
void launch_ystep(hipStream_t st, const T *ax, const T *u, T *y, T thr, T thr21, T u_scale,
                  uint32_t flags, Dims5 d, int dH, int dW, Weight<T> wl1, Weight<T> wl21,
                  Weight<T> ams, int ams_k, int ams_n) {
    YstepArgs<T> p;
    p.ams = ams;
    p.ams_k = ams_k;
    p.ams_n = ams_n;
    p.ax = ax;
    p.u = u;
    p.y = y;
    p.thr = thr;
    p.thr21 = thr21;
    p.u_scale = u_scale;
    p.flags = flags;
    p.d = d;
    p.dH = dH;
    p.dW = dW;
    p.wl1 = wl1;
    p.wl21 = wl21;
    const int64_t total = (int64_t)d.H * d.W * d.N * d.K;
    hipLaunchKernelGGL((ystep_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, st, p);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) ustep_kernel(const T *__restrict__ ax,
                                                         const T *__restrict__ y,
                                                         T *__restrict__ u, T u_scale, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        u[i] = u_scale * u[i] + ax[i] - y[i];
}

template <typename T>
void launch_ustep(hipStream_t st, const T *ax, const T *y, T *u, T u_scale, int64_t n) {
    hipLaunchKernelGGL((ustep_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, ax, y, u,
                       u_scale, n);
    SA_HIP(hipGetLastError());
}

template <typename T> struct StatsArgs {
    const T *x;
    const T *y;
    const T *yprev;
    const T *u;
    uint32_t flags;
    Dims5 d;
    Weight<T> wl1, wl21;
    int ams_k;   // filter index of the AddMaskSim slice, or -1
    int ams_n = 1;   // number of impulse filters from ams_k on
};

template <typename T>
__global__ void __launch_bounds__(kThreads) admm_stats_kernel(const StatsArgs<T> p,
                                                              double *partials) {
    double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const bool gy = p.flags & F_GEVAL_Y, joint = p.flags & F_JOINT;
    const int C = p.d.C;
    const int64_t NK = (int64_t)p.d.N * p.d.K;
    const int64_t total = (int64_t)p.d.H * p.d.W * NK;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / NK;
        const int nk = (int)(t - pix * NK);
        const int k = nk % p.d.K, n = nk / p.d.K;
        const int xw = (int)(pix % p.d.W), h = (int)(pix / p.d.W);
        const int64_t base = pix * C * NK + nk;
        double g2 = 0.0;
        for (int c = 0; c < C; ++c) {
            const int64_t idx = base + c * NK;
            const T w = p.wl1.ptr ? weight_at(p.wl1, h, xw, c, n, k) : T(1);
            const T x = p.x[idx], y = p.y[idx], yo = p.yprev[idx], u = p.u[idx];
            const double dr = (double)(x - y), ds = (double)(y - yo);
            acc[0] += dr * dr;
            acc[1] += ds * ds;
            acc[2] += (double)x * (double)x;
            acc[3] += (double)y * (double)y;
            acc[4] += (double)u * (double)u;
            // (the regularisers do not see the AddMaskSim slice, cbpdn.py:2398-2412)
            const T gvar = is_ams(k, p.ams_k, p.ams_n) ? T(0) : (gy ? y : x);
            const T gv = w * gvar;
            acc[5] += (double)(gv < T(0) ? -gv : gv);
            g2 += (double)gvar * (double)gvar;
        }
        if (joint) {
            const T w21 = p.wl21.ptr ? weight_at(p.wl21, h, xw, 0, n, k) : T(1);
            acc[6] += (double)w21 * sqrt(g2);
        }
    }
    block_sum_store<8>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 8);
}

template <typename T>
int launch_admm_stats(hipStream_t st, const T *x, const T *y, const T *yprev, const T *u,
                      uint32_t flags, Dims5 d, Weight<T> wl1, Weight<T> wl21, int ams_k,
                      double *partials, int ams_n) {
    StatsArgs<T> p;
    p.ams_k = ams_k;
    p.ams_n = ams_n;
    p.x = x;
    p.y = y;
    p.yprev = yprev;
    p.u = u;
    p.flags = flags;
    p.d = d;
    p.wl1 = wl1;
    p.wl21 = wl21;
    const int grid = grid_for((int64_t)d.H * d.W * d.N * d.K);
    hipLaunchKernelGGL((admm_stats_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * 8 * (kThreads / kWave), st, p, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

template <typename T>
__global__ void __launch_bounds__(kThreads) scale_kernel(T *__restrict__ v, T s, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        v[i] *= s;
}

template <typename T> void launch_scale(hipStream_t st, T *v, T s, int64_t n) {
    hipLaunchKernelGGL((scale_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, v, s, n);
    SA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------
// proximal operators
// ---------------------------------------------------------------------------
template <typename T> struct ProxArgs {
    const T *v;
    T *out;
    T thr;
    uint32_t flags;
    Dims5 d;
    int dH, dW;
    Weight<T> wl1;
};

template <typename T>
__global__ void __launch_bounds__(kThreads) prox_l1_kernel(const ProxArgs<T> p, double *partials) {
    double acc[1] = {0.0};
    const bool nonneg = p.flags & F_NONNEG, nob = p.flags & F_NOBNDRY;
    const bool general = nob || p.wl1.ptr != nullptr;
    const int64_t P = (int64_t)p.d.C * p.d.N * p.d.K;
    const int64_t total = (int64_t)p.d.H * p.d.W * P;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        T w = T(1);
        bool kill = false;
        if (general) {
            const int64_t pix = idx / P;
            const int r = (int)(idx - pix * P);
            const int k = r % p.d.K, n = (r / p.d.K) % p.d.N, c = r / (p.d.K * p.d.N);
            const int xw = (int)(pix % p.d.W), h = (int)(pix / p.d.W);
            if (p.wl1.ptr) w = weight_at(p.wl1, h, xw, c, n, k);
            kill = nob && in_bndry(h, xw, p.d.H, p.d.W, p.dH, p.dW);
        }
        T o = soft(p.v[idx], p.thr * w);
        if (nonneg && o < T(0)) o = T(0);
        if (kill) o = T(0);
        p.out[idx] = o;
        const T gv = w * o;
        acc[0] += (double)(gv < T(0) ? -gv : gv);
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + blockIdx.x);
}

template <typename T>
int launch_prox_l1(hipStream_t st, const T *v, T *out, T thr, uint32_t flags, Dims5 d, int dH,
                   int dW, Weight<T> wl1, double *partials) {
    ProxArgs<T> p;
    p.v = v;
    p.out = out;
    p.thr = thr;
    p.flags = flags;
    p.d = d;
    p.dH = dH;
    p.dW = dW;
    p.wl1 = wl1;
    const int grid = grid_for((int64_t)d.H * d.W * d.C * d.N * d.K);
    hipLaunchKernelGGL((prox_l1_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, p, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

template <typename T>
__global__ void __launch_bounds__(kThreads) prox_sl1l2_kernel(const T *__restrict__ v,
                                                              T *__restrict__ out, T alpha, T beta,
                                                              int64_t outer, int C, int64_t inner) {
    const int64_t total = outer * inner;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t o = t / inner, in = t - o * inner;
        const int64_t base = o * C * inner + in;
        T nrm2 = T(0);
        for (int c = 0; c < C; ++c) {
            const T sv = soft(v[base + c * inner], alpha);
            nrm2 += sv * sv;
        }
        const T nrm = sqrt(nrm2);
        T shrink = nrm - beta;
        shrink = shrink > T(0) ? shrink : T(0);
        const T fac = (nrm != T(0)) ? shrink / nrm : T(0);
        for (int c = 0; c < C; ++c) out[base + c * inner] = fac * soft(v[base + c * inner], alpha);
    }
}

template <typename T>
void launch_prox_sl1l2(hipStream_t st, const T *v, T *out, T alpha, T beta, int64_t outer, int C,
                       int64_t inner) {
    hipLaunchKernelGGL((prox_sl1l2_kernel<T>), dim3(grid_for(outer * inner)), dim3(kThreads), 0, st,
                       v, out, alpha, beta, outer, C, inner);
    SA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------
// PGM kernels
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads) pgm_grad_kernel(const cx<T> *__restrict__ v,
                                                            const cx<T> *__restrict__ df,
                                                            const cx<T> *__restrict__ sf,
                                                            cx<T> *__restrict__ gf, int64_t npix,
                                                            int CN, int K, int Wf, int W,
                                                            double *partials) {
    // wave-cooperative when K is a power of two <= 64 (lane = filter), else per-thread loop
    double acc[2] = {0.0, 0.0};
    const bool coop = K <= kWave && (K & (K - 1)) == 0;
    if (coop) {
        const int64_t total = npix * CN * K;
        const int64_t total_pad = (total + kWave - 1) / kWave * kWave;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total_pad;
             t += (int64_t)gridDim.x * blockDim.x) {
            const bool valid = t < total;
            const int64_t grp = t / K;
            const int k = (int)(t - grp * K);
            const int64_t pix = grp / CN;
            cx<T> d = mk<T>(T(0), T(0)), x = d, s = d;
            if (valid) {
                d = df[pix * K + k];
                x = v[t];
                s = sf[grp];
            }
            cx<T> q = cmul(d, x);
            for (int m = K >> 1; m > 0; m >>= 1) {
                q.re += __shfl_xor(q.re, m, kWave);
                q.im += __shfl_xor(q.im, m, kWave);
            }
            const cx<T> r = q - s;
            if (valid) {
                gf[t] = cmulc(d, r);
                if (k == 0) {
                    const double r2 = (double)cabs2(r);
                    acc[0] += r2;
                    acc[1] += parseval_weight((int)(pix % Wf), Wf, W) * r2;
                }
            }
        }
    } else {
        const int64_t total = npix * CN;
        for (int64_t grp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; grp < total;
             grp += (int64_t)gridDim.x * blockDim.x) {
            const int64_t pix = grp / CN;
            cx<T> q = mk<T>(T(0), T(0));
            for (int k = 0; k < K; ++k) q = q + cmul(df[pix * K + k], v[grp * K + k]);
            const cx<T> r = q - sf[grp];
            for (int k = 0; k < K; ++k) gf[grp * K + k] = cmulc(df[pix * K + k], r);
            const double r2 = (double)cabs2(r);
            acc[0] += r2;
            acc[1] += parseval_weight((int)(pix % Wf), Wf, W) * r2;
        }
    }
    block_sum_store<2>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 2);
}

template <typename T>
int launch_pgm_grad(hipStream_t st, const cx<T> *v, const cx<T> *df, const cx<T> *sf, cx<T> *gf,
                    int64_t npix, int CN, int K, int W, double *partials) {
    const bool coop = K <= kWave && (K & (K - 1)) == 0;
    const int grid = grid_for(coop ? npix * CN * K : npix * CN);
    hipLaunchKernelGGL((pgm_grad_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * 2 * (kThreads / kWave), st, v, df, sf, gf, npix, CN, K,
                       W / 2 + 1, W, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

template <typename T>
__global__ void __launch_bounds__(kThreads) axpy_c_kernel(const cx<T> *__restrict__ y,
                                                          const cx<T> *__restrict__ g,
                                                          cx<T> *__restrict__ out, T a, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = y[i] + cscale(g[i], a);
}

template <typename T>
void launch_axpy_c(hipStream_t st, const cx<T> *y, const cx<T> *g, cx<T> *out, T a, int64_t n) {
    hipLaunchKernelGGL((axpy_c_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, y, g, out, a, n);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) lincomb_kernel(cx<T> *__restrict__ dst, T a,
                                                           const cx<T> *va, T b, const cx<T> *vb,
                                                           T c, const cx<T> *vc, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        cx<T> r = cscale(va[i], a);
        if (vb) r = r + cscale(vb[i], b);
        if (vc) r = r + cscale(vc[i], c);
        dst[i] = r;
    }
}

template <typename T>
void launch_lincomb(hipStream_t st, cx<T> *dst, T a, const cx<T> *va, T b, const cx<T> *vb, T c,
                    const cx<T> *vc, int64_t n) {
    hipLaunchKernelGGL((lincomb_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, dst, a, va, b,
                       vb, c, vc, n);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) pair_stats_kernel(const cx<T> *__restrict__ a,
                                                              const cx<T> *__restrict__ b,
                                                              const cx<T> *__restrict__ g,
                                                              int64_t npix, int64_t cols, int Wf,
                                                              int W, double *partials) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t total = npix * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / cols;
        cx<T> dlt = a[i];
        if (b) dlt = dlt - b[i];
        const double d2 = (double)cabs2(dlt);
        acc[0] += parseval_weight((int)(pix % Wf), Wf, W) * d2;
        acc[2] += d2;
        if (g) {
            const cx<T> gg = g[i];
            acc[1] += (double)dlt.re * (double)gg.re + (double)dlt.im * (double)gg.im;
            acc[3] += (double)cabs2(gg);
        }
    }
    block_sum_store<4>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 4);
}

template <typename T>
int launch_pair_stats(hipStream_t st, const cx<T> *a, const cx<T> *b, const cx<T> *g, int64_t npix,
                      int64_t cols, int W, double *partials) {
    const int grid = grid_for(npix * cols);
    hipLaunchKernelGGL((pair_stats_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * 4 * (kThreads / kWave), st, a, b, g, npix, cols, W / 2 + 1,
                       W, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

template <typename T>
__global__ void __launch_bounds__(kThreads) dhs_absmax_kernel(const cx<T> *__restrict__ df,
                                                              const cx<T> *__restrict__ sf,
                                                              int64_t npix, int CN, int K,
                                                              double *partials) {
    double m = 0.0;
    const int64_t total = npix * CN * K;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t grp = t / K;
        const int k = (int)(t - grp * K);
        const int64_t pix = grp / CN;
        const double v = (double)cabs2(cmulc(df[pix * K + k], sf[grp]));
        m = v > m ? v : m;
    }
    // block max through LDS
    double *scratch = dyn_lds<double>();
    for (int s = kWave / 2; s > 0; s >>= 1) {
        const double o = __shfl_xor(m, s, kWave);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) scratch[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int j = 0; j < (int)(blockDim.x / kWave); ++j) r = scratch[j] > r ? scratch[j] : r;
        partials[blockIdx.x] = r;
    }
}

template <typename T>
int launch_dhs_absmax(hipStream_t st, const cx<T> *df, const cx<T> *sf, int64_t npix, int CN,
                      int K, double *partials) {
    const int grid = grid_for(npix * CN * K);
    hipLaunchKernelGGL((dhs_absmax_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, df, sf, npix, CN, K, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// dictionary update (D-step) kernels
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ cx<T> wave_sum_cx(cx<T> v) {
#pragma unroll
    for (int m = kWave / 2; m > 0; m >>= 1) {
        v.re += __shfl_xor(v.re, m, kWave);
        v.im += __shfl_xor(v.im, m, kWave);
    }
    return v;
}

// One workgroup per pixel (grid-stride): phase 1 gives every wave whole rows of
// zf[n, :] (lanes = filters, coalesced) and reduces them to r[n] in LDS; phase 2
// re-walks the same rows (now L1/L2 resident) accumulating conj(zf) r per filter.
template <typename T>
__global__ void __launch_bounds__(kThreads) ccmod_grad_kernel(const cx<T> *__restrict__ zf,
                                                              const cx<T> *__restrict__ d,
                                                              const cx<T> *__restrict__ sf,
                                                              cx<T> *__restrict__ gf, int64_t npix,
                                                              int CN, int K, int Wf, int W, int Cd,
                                                              double *partials, int zch) {
    // Cd > 1 (multi-channel dictionary): d, gf are (npix, Cd, K), sf is (npix, Cd, CN); the
    // channels are independent least-squares problems sharing zf (pgm/ccmod.py:295-317) -- or,
    // zch, each with its own coefficient maps: zf is (npix, CN, Cd, K)
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int nwave = blockDim.x / kWave;
    cx<T> *r = dyn_lds<cx<T>>();               // [CN]
    cx<T> *gpart = r + CN;                     // [nwave][K]
    double *red = reinterpret_cast<double *>(gpart + (size_t)nwave * K);  // [3 * nwave]
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t pc = blockIdx.x; pc < npix * Cd; pc += gridDim.x) {
        const int64_t pix = pc / Cd;
        const int64_t zs = zch ? (int64_t)Cd * K : K;      // stride of the images in zf
        const cx<T> *zp = zf + pix * CN * zs + (zch ? (pc - pix * Cd) * K : 0);
        const cx<T> *dp = d + pc * K;
        for (int n = wave; n < CN; n += nwave) {
            cx<T> q = mk<T>(T(0), T(0));
            for (int k = lane; k < K; k += kWave) q = q + cmul(zp[(int64_t)n * zs + k], dp[k]);
            q = wave_sum_cx(q);
            if (lane == 0) {
                const cx<T> rr = q - sf[pc * CN + n];
                r[n] = rr;
                const double r2 = (double)cabs2(rr);
                acc[0] += r2;
                acc[1] += parseval_weight((int)(pix % Wf), Wf, W) * r2;
                acc[2] += (double)cabs2(q);
            }
        }
        __syncthreads();
        if (gf) {
            for (int k = lane; k < K; k += kWave) {
                cx<T> g = mk<T>(T(0), T(0));
                for (int n = wave; n < CN; n += nwave) g = g + cmulc(zp[(int64_t)n * zs + k], r[n]);
                gpart[wave * K + k] = g;
            }
            __syncthreads();
            for (int k = threadIdx.x; k < K; k += blockDim.x) {
                cx<T> g = gpart[k];
                for (int w = 1; w < nwave; ++w) g = g + gpart[w * K + k];
                gf[pc * K + k] = g;
            }
        }
        __syncthreads();
    }
    block_sum_store<3>(acc, red, partials + (int64_t)blockIdx.x * 3);
}

// The same for a handful of images: a wave per (frequency, channel) walks the images itself --
// filters on the lanes, the gradient in registers, no workgroup barrier (K <= 256).
template <typename T>
__global__ void __launch_bounds__(kThreads) ccmod_grad_wave_kernel(const cx<T> *__restrict__ zf,
                                                                   const cx<T> *__restrict__ d,
                                                                   const cx<T> *__restrict__ sf,
                                                                   cx<T> *__restrict__ gf, int64_t npix,
                                                                   int CN, int K, int Wf, int W, int Cd,
                                                                   double *partials, int zch) {
    constexpr int KR = 4;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t pc = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; pc < npix * Cd;
         pc += nwaves) {
        const int64_t pix = pc / Cd;
        const int64_t zs = zch ? (int64_t)Cd * K : K;
        const cx<T> *zp = zf + pix * CN * zs + (zch ? (pc - pix * Cd) * K : 0);
        const cx<T> *dp = d + pc * K;
        const double pw = parseval_weight((int)(pix % Wf), Wf, W);
        cx<T> dk[KR], g[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + kWave * j;
            dk[j] = k < K ? dp[k] : mk<T>(T(0), T(0));
            g[j] = mk<T>(T(0), T(0));
        }
        for (int n = 0; n < CN; ++n) {
            cx<T> zk[KR], q = mk<T>(T(0), T(0));
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                zk[j] = k < K ? zp[(int64_t)n * zs + k] : mk<T>(T(0), T(0));
                if (k < K) q = q + cmul(zk[j], dk[j]);
            }
            q = wave_sum_cx(q);
            const cx<T> rr = q - sf[pc * CN + n];
#pragma unroll
            for (int j = 0; j < KR; ++j) g[j] = g[j] + cmulc(zk[j], rr);
            if (lane == 0) {
                const double r2 = (double)cabs2(rr);
                acc[0] += r2;
                acc[1] += pw * r2;
                acc[2] += (double)cabs2(q);
            }
        }
        if (gf) {
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) gf[pc * K + k] = g[j];
            }
        }
    }
    block_sum_store<3>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 3);
}

template <typename T>
int launch_ccmod_grad(hipStream_t st, const cx<T> *zf, const cx<T> *d, const cx<T> *sf, cx<T> *gf,
                      int64_t npix, int CN, int K, int W, double *partials, int Cd, int zch) {
    if (CN <= 16 && K <= 256) {
        const int grid = std::min(grid_for(npix * Cd * kWave), kMaxPartialBlocks);
        hipLaunchKernelGGL((ccmod_grad_wave_kernel<T>), dim3(grid), dim3(kThreads),
                           sizeof(double) * 3 * (kThreads / kWave), st, zf, d, sf, gf, npix, CN, K, W / 2 + 1,
                           W, Cd, partials, zch);
        SA_HIP(hipGetLastError());
        return grid;
    }
    int grid = (int)(npix * Cd < kMaxPartialBlocks ? npix * Cd : kMaxPartialBlocks);
    const int nwave = kThreads / kWave;
    size_t lds = sizeof(cx<T>) * ((size_t)CN + (size_t)nwave * K) + sizeof(double) * 3 * nwave;
    lds = (lds + 15) / 16 * 16;
    SA_REQUIRE(lds <= 64 * 1024, "too many images x filters for the D-step gradient kernel");
    hipLaunchKernelGGL((ccmod_grad_kernel<T>), dim3(grid), dim3(kThreads), lds, st, zf, d, sf, gf,
                       npix, CN, K, W / 2 + 1, W, Cd, partials, zch);
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// masked data fidelity (pgm.cbpdn.ConvBPDNMask, pgm/cbpdn.py:387-506; pgm.ccmod.ConvCnstrMODMask,
// pgm/ccmod.py:408-604): the residual goes to the spatial domain, is weighted, and comes back
// ---------------------------------------------------------------------------
// r(H, W, C, N) <- w^p r (p = 1 or 2), w broadcastable (H, W, C, N, 1); partial[block] = sum (w r)^2
// of the INPUT r (so that both powers report the weighted residual energy)
template <typename T>
__global__ void __launch_bounds__(kThreads) mask_apply_kernel(T *__restrict__ r, const Weight<T> w,
                                                              int squared, int W_, int C, int N,
                                                              int64_t n, double *partials) {
    double acc[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int nn = (int)(i % N);
        const int c = (int)((i / N) % C);
        const int64_t pix = i / ((int64_t)N * C);
        const int x = (int)(pix % W_), h = (int)(pix / W_);
        const T wv = w.ptr ? weight_at(w, h, x, c, nn, 0) : T(1);
        const T v = r[i], wr = wv * v;
        acc[0] += (double)wr * (double)wr;
        r[i] = squared ? wv * wr : wr;
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + blockIdx.x);
}

template <typename T>
int launch_mask_apply(hipStream_t st, T *r, const Weight<T> &w, bool squared, int H, int W, int C,
                      int N, double *partials) {
    const int64_t n = (int64_t)H * W * C * N;
    const int grid = grid_for(n);
    hipLaunchKernelGGL((mask_apply_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, r, w, squared ? 1 : 0, W, C, N, n,
                       partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// ConvBPDNMaskDcpl: the signal-sized block (Y0, U0) of the two-block constraint
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads) md_pre_kernel(const T *__restrict__ y0,
                                                          const T *__restrict__ u0,
                                                          const T *__restrict__ s,
                                                          T *__restrict__ out, T us, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = y0[i] - us * u0[i] + s[i];
}

template <typename T>
void launch_md_pre(hipStream_t st, const T *y0, const T *u0, const T *s, T *out, T us, int64_t n) {
    hipLaunchKernelGGL((md_pre_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, y0, u0, s, out,
                       us, n);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) md_y0step_kernel(const MdY0Args<T> a, int64_t n,
                                                             double *partials) {
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int nn = (int)(i % a.N);
        const int c = (int)((i / a.N) % a.C);
        const int64_t pix = i / ((int64_t)a.N * a.C);
        const int x = (int)(pix % a.W), h = (int)(pix / a.W);
        const T wv = a.w.ptr ? weight_at(a.w, h, x, c, nn, 0) : T(1);
        const T axnr = a.ax0nr[i], sv = a.s[i], yo = a.y0[i], uo = a.us * a.u0[i];
        const T ax = a.rlx == T(1) ? axnr : a.rlx * axnr + (T(1) - a.rlx) * (yo + sv);
        const T yn = (a.rho * (ax + uo - sv)) / (wv * wv + a.rho);
        const T un = uo + (ax - (yn + sv));
        a.y0[i] = yn;
        a.u0[i] = un;
        const double r = (double)(axnr - (yn + sv));
        const double g = (double)(wv * (a.geval_y ? yn : axnr - sv));
        acc[0] += r * r;
        acc[1] += (double)axnr * (double)axnr;
        acc[2] += (double)yn * (double)yn;
        acc[3] += (double)un * (double)un;
        acc[4] += g * g;
    }
    block_sum_store<5>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 5);
}

template <typename T>
int launch_md_y0step(hipStream_t st, const MdY0Args<T> &a, double *partials) {
    const int64_t n = (int64_t)a.H * a.W * a.C * a.N;
    const int grid = grid_for(n);
    hipLaunchKernelGGL((md_y0step_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * 5 * (kThreads / kWave), st, a, n, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// gf[pix, cn, k] = conj(df[pix, k]) r[pix, cn]      (D^H applied to a signal-sized spectrum)
template <typename T>
__global__ void __launch_bounds__(kThreads) conj_outer_kernel(const cx<T> *__restrict__ df,
                                                              const cx<T> *__restrict__ r,
                                                              cx<T> *__restrict__ gf, int64_t npix,
                                                              int CN, int K) {
    const int64_t total = npix * CN * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t grp = i / K;
        gf[i] = cmulc(df[(grp / CN) * K + k], r[grp]);
    }
}

template <typename T>
void launch_conj_outer(hipStream_t st, const cx<T> *df, const cx<T> *r, cx<T> *gf, int64_t npix,
                       int CN, int K) {
    hipLaunchKernelGGL((conj_outer_kernel<T>), dim3(grid_for(npix * CN * K)), dim3(kThreads), 0, st,
                       df, r, gf, npix, CN, K);
    SA_HIP(hipGetLastError());
}

// gf[pix, k] = sum_n conj(zf[pix, n, k]) r[pix, n]   (inner over the image axis, pgm/ccmod.py:570)
template <typename T>
__global__ void __launch_bounds__(kThreads) zf_adjoint_kernel(const cx<T> *__restrict__ zf,
                                                              const cx<T> *__restrict__ r,
                                                              cx<T> *__restrict__ gf, int64_t npix,
                                                              int CN, int K) {
    const int64_t total = npix * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t pix = i / K;
        cx<T> g = mk<T>(T(0), T(0));
        for (int n = 0; n < CN; ++n) g = g + cmulc(zf[(pix * CN + n) * K + k], r[pix * CN + n]);
        gf[i] = g;
    }
}

// Multi-channel dictionary: gf[pix, c, k] = sum_n conj(zf[pix, n, k]) r[pix, c, n]
// (zch: zf[pix, n, c, k])
template <typename T>
__global__ void __launch_bounds__(kThreads) mc_zf_adjoint_kernel(const cx<T> *__restrict__ zf,
                                                                 const cx<T> *__restrict__ r,
                                                                 cx<T> *__restrict__ gf, int64_t npix,
                                                                 int Cd, int N, int K, int zch) {
    const int64_t total = npix * Cd * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K), c = (int)((i / K) % Cd);
        const int64_t pix = i / ((int64_t)K * Cd);
        cx<T> acc = mk<T>(T(0), T(0));
        for (int n = 0; n < N; ++n)
            acc = acc + cmulc(zf[(zch ? (pix * N + n) * Cd + c : pix * N + n) * K + k],
                              r[(pix * Cd + c) * N + n]);
        gf[i] = acc;
    }
}
template <typename T>
void launch_mc_zf_adjoint(hipStream_t st, const cx<T> *zf, const cx<T> *r, cx<T> *gf, int64_t npix,
                          int Cd, int N, int K, int zch) {
    hipLaunchKernelGGL((mc_zf_adjoint_kernel<T>), dim3(grid_for(npix * Cd * K)), dim3(kThreads), 0, st,
                       zf, r, gf, npix, Cd, N, K, zch);
    SA_HIP(hipGetLastError());
}

template <typename T>
void launch_zf_adjoint(hipStream_t st, const cx<T> *zf, const cx<T> *r, cx<T> *gf, int64_t npix,
                       int CN, int K) {
    hipLaunchKernelGGL((zf_adjoint_kernel<T>), dim3(grid_for(npix * K)), dim3(kThreads), 0, st, zf,
                       r, gf, npix, CN, K);
    SA_HIP(hipGetLastError());
}

// Residual per frequency on the TILE-MAJOR layout of the fused kernels (csc_fused.h):
// out[tile][h] = sum_k dft[wf][h][k] v[tile][h][k] - sft[tile][h]   (eval_Rf, pgm/cbpdn.py:281-286),
// one wave per (tile, h) row of K filters (rows Ks apart).
template <typename T>
__global__ void __launch_bounds__(kThreads) tiled_resid_kernel(const cx<T> *__restrict__ v,
                                                               const cx<T> *__restrict__ dft,
                                                               const cx<T> *__restrict__ sft,
                                                               cx<T> *__restrict__ out, int64_t nrows,
                                                               int H, int K, int Ks, int CN) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    for (int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; row < nrows;
         row += nwaves) {
        const int64_t tile = row / H;
        const int h = (int)(row - tile * H);
        const int64_t wf = tile / CN;
        const cx<T> *d = dft + (wf * H + h) * Ks, *x = v + row * Ks;
        cx<T> s = mk<T>(T(0), T(0));
        for (int k = lane; k < K; k += kWave) s = s + cmul(d[k], x[k]);
        s = wave_sum_cx(s);
        if (lane == 0) out[row] = s - sft[row];
    }
}
template <typename T>
void launch_tiled_resid(hipStream_t st, const cx<T> *v, const cx<T> *dft, const cx<T> *sft, cx<T> *out,
                        int64_t ntiles, int H, int K, int Ks, int CN) {
    hipLaunchKernelGGL((tiled_resid_kernel<T>), dim3(grid_for(ntiles * H * kWave)), dim3(kThreads), 0, st,
                       v, dft, sft, out, ntiles * H, H, K, Ks ? Ks : K, CN);
    SA_HIP(hipGetLastError());
}

// Dual residual of the mask-decoupled X-step on the TILE-MAJOR layout of the fused kernels
// (csc_fused.h): sum over (tile, h, k) of pw(wf) |conj(dft[wf][h][k]) u0t[tile][h] + t[tile][h][k]|^2
// with t the 2-D spectrum of u1 and u0t that of u0 -- rho^2 ||A^T u||^2 H W without its factors
// (cbpdn.py:1814-1818).  One wave per (tile, h) row of K filters (rows Ks apart).
template <typename T>
__global__ void __launch_bounds__(kThreads) md_dualres_tiled_kernel(const cx<T> *__restrict__ t,
                                                                    const cx<T> *__restrict__ dft,
                                                                    const cx<T> *__restrict__ u0t,
                                                                    int64_t nrows, int H, int K, int Ks,
                                                                    int CN, int Wf, int W,
                                                                    double *partials) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    double acc[1] = {0.0};
    for (int64_t row = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; row < nrows;
         row += nwaves) {
        const int64_t tile = row / H;
        const int h = (int)(row - tile * H);
        const int wf = (int)(tile / CN);
        const cx<T> u0 = u0t[row];
        const cx<T> *d = dft + ((int64_t)wf * H + h) * Ks, *x = t + row * Ks;
        double s = 0.0;
        for (int k = lane; k < K; k += kWave) s += (double)cabs2(cmulc(d[k], u0) + x[k]);
        acc[0] += parseval_weight(wf, Wf, W) * s;
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + blockIdx.x);
}
template <typename T>
int launch_md_dualres_tiled(hipStream_t st, const cx<T> *t, const cx<T> *dft, const cx<T> *u0t,
                            int64_t ntiles, int H, int K, int Ks, int CN, int W, double *partials) {
    const int grid = std::min(grid_for(ntiles * H * kWave), kMaxPartialBlocks);
    hipLaunchKernelGGL((md_dualres_tiled_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, t, dft, u0t, ntiles * H, H, K, Ks ? Ks : K, CN,
                       W / 2 + 1, W, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// ADMM consensus dictionary update (admm/ccmod.py:605-908 on admm/admm.py:1441-1707):
// one dictionary copy X_n (and dual U_n) per image, consensus variable Y (H, W, K)
// ---------------------------------------------------------------------------
// out[pix, n, k] = y[pix, k] - s * u[pix, n, k]        (ccmod.py:768: Y[..., newaxis] - U)
template <typename T>
__global__ void __launch_bounds__(kThreads) cns_yu_kernel(const T *__restrict__ y,
                                                          const T *__restrict__ u,
                                                          T *__restrict__ out, T s, int64_t npixr,
                                                          int CN, int K) {
    const int64_t total = npixr * CN * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t pix = i / ((int64_t)K * CN);
        out[i] = y[pix * K + k] - s * u[i];
    }
}

// m[pix, k] = mean_n (a x + (1 - a) y + s u)    (relax_AX admm.py:1608-1616, ystep :1585-1591)
template <typename T>
__global__ void __launch_bounds__(kThreads) cns_mean_kernel(const T *__restrict__ x,
                                                            const T *__restrict__ u,
                                                            const T *__restrict__ y,
                                                            T *__restrict__ m, T a, T s,
                                                            int64_t npixr, int CN, int K) {
    const int64_t total = npixr * K;
    const T inv = T(1) / (T)CN;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t pix = i / K;
        const T yo = (T(1) - a) * y[i];
        T acc = T(0);
        for (int n = 0; n < CN; ++n) {
            const int64_t j = (pix * CN + n) * K + k;
            acc += a * x[j] + yo + s * u[j];
        }
        m[i] = acc * inv;
    }
}

// u = s u + (a x + (1 - a) yold) - ynew   (ustep, admm.py:434-437 with rsdl_r :1673-1676);
// partials per block (4): sum (x - ynew)^2, sum x^2, sum u_new^2, unused
template <typename T>
__global__ void __launch_bounds__(kThreads) cns_ustep_kernel(const T *__restrict__ x,
                                                             T *__restrict__ u,
                                                             const T *__restrict__ yold,
                                                             const T *__restrict__ ynew, T a, T s,
                                                             int64_t npixr, int CN, int K,
                                                             double *partials) {
    const int64_t total = npixr * CN * K;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int64_t pix = i / ((int64_t)K * CN);
        const T xv = x[i], yn = ynew[pix * K + k];
        const T un = s * u[i] + (a * xv + (T(1) - a) * yold[pix * K + k]) - yn;
        u[i] = un;
        const double dr = (double)(xv - yn);
        acc[0] += dr * dr;
        acc[1] += (double)xv * (double)xv;
        acc[2] += (double)un * (double)un;
    }
    block_sum_store<4>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 4);
}

// partials per block (2): sum (ynew - yold)^2, sum ynew^2
template <typename T>
__global__ void __launch_bounds__(kThreads) cns_ystats_kernel(const T *__restrict__ yold,
                                                              const T *__restrict__ ynew,
                                                              int64_t n, double *partials) {
    double acc[2] = {0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const double d = (double)(ynew[i] - yold[i]);
        acc[0] += d * d;
        acc[1] += (double)ynew[i] * (double)ynew[i];
    }
    block_sum_store<2>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 2);
}

template <typename T>
void launch_cns_yu(hipStream_t st, const T *y, const T *u, T *out, T s, int64_t npixr, int CN,
                   int K) {
    hipLaunchKernelGGL((cns_yu_kernel<T>), dim3(grid_for(npixr * CN * K)), dim3(kThreads), 0, st, y,
                       u, out, s, npixr, CN, K);
    SA_HIP(hipGetLastError());
}
template <typename T>
void launch_cns_mean(hipStream_t st, const T *x, const T *u, const T *y, T *m, T a, T s,
                     int64_t npixr, int CN, int K) {
    hipLaunchKernelGGL((cns_mean_kernel<T>), dim3(grid_for(npixr * K)), dim3(kThreads), 0, st, x, u,
                       y, m, a, s, npixr, CN, K);
    SA_HIP(hipGetLastError());
}
template <typename T>
int launch_cns_ustep(hipStream_t st, const T *x, T *u, const T *yold, const T *ynew, T a, T s,
                     int64_t npixr, int CN, int K, double *partials) {
    const int grid = grid_for(npixr * CN * K);
    hipLaunchKernelGGL((cns_ustep_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * 4 * (kThreads / kWave), st, x, u, yold, ynew, a, s, npixr,
                       CN, K, partials);
    SA_HIP(hipGetLastError());
    return grid;
}
template <typename T>
int launch_cns_ystats(hipStream_t st, const T *yold, const T *ynew, int64_t n, double *partials) {
    const int grid = grid_for(n);
    hipLaunchKernelGGL((cns_ystats_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * 2 * (kThreads / kWave), st, yold, ynew, n, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// multi-channel dictionaries (Cd > 1): iterated Sherman-Morrison, linalg.solvemdbi_ism
// (linalg.py:370-444) as called by GenericConvBPDN.xstep (cbpdn.py:277-279)
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ cx<T> cdivide(cx<T> a, cx<T> b) {
    const T s = T(1) / cabs2(b);
    return cscale(cmulc(b, a), s);   // a conj(b) / |b|^2
}

// gam[pix, c, :], del[pix, c] and mm[pix, c, l]: the vectors gamma_c and scalars delta_c of
// the recursion (linalg.py:418-441) and the products M_cl = <ah_c, gamma_l>; they depend on
// Df and rho only.  One wave per frequency, lane = filter (KR chunks of 64: K <= 64 KR).
template <typename T, int KR>
__global__ void __launch_bounds__(kThreads) ism_setup_kernel(const cx<T> *__restrict__ df,
                                                             cx<T> *__restrict__ gam,
                                                             cx<T> *__restrict__ del,
                                                             cx<T> *__restrict__ mm, int64_t npix,
                                                             int Cd, int K, T rho, GradTerm<T> gt,
                                                             int Wf) {
    constexpr int CMAX = 8;
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    const T irho = T(1) / rho;
    for (int64_t pix = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
         pix < npix; pix += nwaves) {
        const cx<T> *d = df + pix * Cd * K;
        cx<T> *g = gam + pix * Cd * K;
        cx<T> dl[CMAX];
        // the identity term is rho, or the diagonal mu wg GHGf + rho of ConvBPDNGradReg
        // (cbpdn.py:1181-1184: solvemdbi_ism with an array for `rho`)
        T idg[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + kWave * j;
            idg[j] = irho;
            if (gt.ghh && k < K) idg[j] = T(1) / (gt.mu * (grad_w(gt, k) * grad_gh(gt, pix, Wf)) + rho);
        }
        // (loops unrolled over the CMAX possible terms: delta and gamma stay in registers)
        cx<T> gr[CMAX][KR];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < Cd) {
                cx<T> al[KR], dc[KR];
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    const int k = lane + kWave * j;
                    dc[j] = k < K ? d[c * K + k] : mk<T>(T(0), T(0));
                    al[j] = k < K ? cscale(cconj(dc[j]), idg[j]) : mk<T>(T(0), T(0));
                }
#pragma unroll
                for (int l = 0; l < CMAX; ++l) {
                    if (l < c) {
                        cx<T> t = mk<T>(T(0), T(0));
#pragma unroll
                        for (int j = 0; j < KR; ++j) {
                            const int k = lane + kWave * j;
                            if (k < K) t = t + cmul(d[l * K + k], al[j]);
                        }
                        const cx<T> f = cdivide(wave_sum_cx(t), dl[l]);
#pragma unroll
                        for (int j = 0; j < KR; ++j) al[j] = al[j] - cmul(gr[l][j], f);
                    }
                }
                cx<T> t = mk<T>(T(0), T(0));
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    const int k = lane + kWave * j;
                    gr[c][j] = al[j];
                    if (k < K) {
                        g[c * K + k] = al[j];
                        t = t + cmul(dc[j], al[j]);
                    }
                }
                t = wave_sum_cx(t);
                dl[c] = mk<T>(T(1) + t.re, t.im);
                if (lane == 0) del[pix * Cd + c] = dl[c];
            }
        }
        // M_cl = sum_k d_c[k] gamma_l[k] for every pair (the solve kernel then needs only the
        // Cd inner products with b / rho, taken together)
        // (a row of M at a time: its up to CMAX reductions are independent and overlap)
        for (int c = 0; c < Cd; ++c) {
            cx<T> dc[KR], t[CMAX];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                dc[j] = k < K ? d[c * K + k] : mk<T>(T(0), T(0));
            }
#pragma unroll
            for (int l = 0; l < CMAX; ++l) {
                t[l] = mk<T>(T(0), T(0));
                if (l < Cd) {
#pragma unroll
                    for (int j = 0; j < KR; ++j) {
                        const int k = lane + kWave * j;
                        if (k < K) t[l] = t[l] + cmul(dc[j], gr[l][j]);
                    }
                }
            }
#pragma unroll
            for (int l = 0; l < CMAX; ++l)
                if (l < Cd) t[l] = wave_sum_cx(t[l]);
#pragma unroll
            for (int l = 0; l < CMAX; ++l)
                if (l < Cd && lane == 0) mm[(pix * Cd + c) * Cd + l] = t[l];
        }
    }
}

template <typename T> struct IsmArgs {
    const cx<T> *yuf;   // (npix, N, K): rfftn(Y - U)
    cx<T> *xf;          // out (may alias yuf)
    const cx<T> *df;    // (npix, Cd, K)
    const cx<T> *sf;    // (npix, Cd, N)
    const cx<T> *gam;   // (npix, Cd, K)
    const cx<T> *del;   // (npix, Cd)
    const cx<T> *mm;    // (npix, Cd, Cd)
    T rho;
    int64_t npix;
    int Cd, N, K, W;
    int want_obj, want_xrrs;
    double *partials;   // 4 doubles per block (5 with the gradient term), as launch_sm_solve
    GradTerm<T> g;      // GRAD instantiations: the diagonal is mu wg GHGf + rho
};

// xf = solvemdbi_ism(Df, rho, sum_c conj(Df) Sf + rho yuf): one workgroup per frequency, its
// waves take the images in turn, lane = filter.  With beta0 = b / rho and t_c = <ah_c, beta0> (the only reductions,
// taken together), the recursion of linalg.py:425-441 unrolls to
//     f_c = (t_c - sum_{l<c} M_cl f_l) / delta_c,     x = beta0 - sum_c gamma_c f_c,
//     (D x)_c = t_c - sum_l M_cl f_l.
// CC: compile-time channel count (2..4), or 0 for a run-time Cd <= 8.
__host__ __device__ inline int ism_waves_per_pixel(int nrhs, int wpb) {
    return nrhs >= 3 ? wpb : (nrhs == 2 ? 2 : 1);
}
template <typename T, int KR, int CC, bool GRAD>
__global__ void __launch_bounds__(kThreads) ism_solve_kernel(const IsmArgs<T> a) {
    constexpr int CM = CC ? CC : 8;
    constexpr int NA = GRAD ? 5 : 4;
    const int lane = threadIdx.x & (kWave - 1);
    const int Wf = a.W / 2 + 1, K = a.K, Cd = CC ? CC : a.Cd;
    const T rho = a.rho, irho = T(1) / a.rho;
    // The kThreads / kWave waves of a workgroup share one frequency and take its images in
    // turn: Df, gamma, M and delta of the frequency are loaded once per wave, into registers.
    // With fewer right-hand sides than waves (the dictionary update has one: the images are
    // the rank-one terms there) the waves spread over neighbouring frequencies instead.
    constexpr int WPB = kThreads / kWave;
    const int wpp = ism_waves_per_pixel(a.N, WPB);
    const int wv = (threadIdx.x / kWave) % wpp, psub = (threadIdx.x / kWave) / wpp, ppb = WPB / wpp;
    double acc[NA] = {};
    for (int64_t pix = (int64_t)blockIdx.x * ppb + psub; pix < a.npix; pix += (int64_t)gridDim.x * ppb) {
        const cx<T> *dp = a.df + pix * Cd * K;
        const cx<T> *gp = a.gam + pix * Cd * K;
        const cx<T> *M = a.mm + pix * Cd * Cd;
        const double pw = parseval_weight((int)(pix % Wf), Wf, a.W);
        cx<T> d[CM][KR], g[CM][KR], dl[CM];
        T dg[KR], gwh[KR];   // GRAD: the diagonal and wg GHGf of this lane's filters
        if constexpr (GRAD) {
            const T gh = grad_gh(a.g, pix, Wf);
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                gwh[j] = k < K ? grad_w(a.g, k) * gh : T(0);
                dg[j] = a.g.mu * gwh[j] + rho;
            }
        }
#pragma unroll
        for (int c = 0; c < CM; ++c)
            if (c < Cd) {
                dl[c] = a.del[pix * Cd + c];
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    const int k = lane + kWave * j;
                    d[c][j] = g[c][j] = mk<T>(T(0), T(0));
                    if (k < K) {
                        d[c][j] = dp[c * K + k];
                        g[c][j] = gp[c * K + k];
                    }
                }
            }
        for (int n = wv; n < a.N; n += wpp) {
            const int64_t sys = pix * a.N + n;
            cx<T> sc[CM], t[CM], f[CM], dx[CM];
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                t[c] = mk<T>(T(0), T(0));
                if (c < Cd) sc[c] = a.sf[(pix * Cd + c) * a.N + n];
            }
            cx<T> be[KR];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                be[j] = mk<T>(T(0), T(0));
                if (k < K) {
                    cx<T> v = a.yuf[sys * K + k];
                    if constexpr (GRAD) {
                        v = cscale(v, rho);                      // b = rho yuf + sum_c conj(d_c) s_c
#pragma unroll
                        for (int c = 0; c < CM; ++c)
                            if (c < Cd) v = v + cmulc(d[c][j], sc[c]);
                        v = cscale(v, T(1) / dg[j]);
                    } else {
#pragma unroll
                        for (int c = 0; c < CM; ++c)
                            if (c < Cd) v = v + cscale(cmulc(d[c][j], sc[c]), irho);
                    }
                    be[j] = v;                                   // beta0 = b / rho
#pragma unroll
                    for (int c = 0; c < CM; ++c)
                        if (c < Cd) t[c] = t[c] + cmul(d[c][j], v);
                }
            }
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < Cd) t[c] = wave_sum_cx(t[c]);
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < Cd) {
                    cx<T> r = t[c];
#pragma unroll
                    for (int l = 0; l < CM; ++l)
                        if (l < c) r = r - cmul(M[c * Cd + l], f[l]);
                    f[c] = cdivide(r, dl[c]);
                }
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < Cd) {
                    cx<T> r = t[c];
#pragma unroll
                    for (int l = 0; l < CM; ++l)
                        if (l < Cd) r = r - cmul(M[c * Cd + l], f[l]);
                    dx[c] = r;                                   // (D x)_c
                    if (a.want_obj && lane == 0) acc[0] += pw * (double)cabs2(r - sc[c]);
                }
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) {
                    cx<T> x = be[j];
#pragma unroll
                    for (int c = 0; c < CM; ++c)
                        if (c < Cd) x = x - cmul(g[c][j], f[c]);
                    a.xf[sys * K + k] = x;
                    if constexpr (GRAD) {
                        if (a.want_obj) acc[4] += pw * (double)gwh[j] * (double)cabs2(x);
                    }
                    if (a.want_xrrs) {
                        cx<T> ax = cscale(x, GRAD ? dg[j] : rho);
#pragma unroll
                        for (int c = 0; c < CM; ++c)
                            if (c < Cd) ax = ax + cmulc(d[c][j], dx[c]);
                        const cx<T> b = cscale(be[j], GRAD ? dg[j] : rho);
                        acc[1] += (double)cabs2(ax - b);
                        acc[2] += (double)cabs2(ax);
                        acc[3] += (double)cabs2(b);
                    }
                }
            }
        }
    }
    block_sum_store<NA>(acc, dyn_lds<double>(), a.partials + (int64_t)blockIdx.x * NA);
}

// ---- LinSolveCheck of the consensus dictionary update (csc_kernels.h) -----------------------
// one wave per frequency, lane = filter (KR of them per lane)
template <typename T, int KR>
__global__ void __launch_bounds__(kThreads) cns_xrrs_rhs_kernel(const cx<T> *__restrict__ zf,
                                                                const cx<T> *__restrict__ sf,
                                                                const cx<T> *__restrict__ yuf, T rho,
                                                                cx<T> *__restrict__ bsum, int64_t npix,
                                                                int CN, int K, int Cd, int zch) {
    // (Cd > 1: the CN systems of a pixel are (image, channel) pairs, channel fastest, sharing the
    // image's zf row -- or, zch, each with its own row of a (npix, N, Cd, K) zf; one wave per
    // (pixel, channel), bsum (npix, Cd, K))
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    const int NI = CN / Cd;
    for (int64_t pc = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; pc < npix * Cd;
         pc += nwaves) {
        const int64_t pix = pc / Cd;
        const int c = (int)(pc - pix * Cd);
        cx<T> b[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) b[j] = mk<T>(T(0), T(0));
        for (int n = 0; n < NI; ++n) {
            const int64_t row = ((pix * NI + n) * Cd + c) * K, zrow = zch ? row : (pix * NI + n) * K;
            const cx<T> s = sf[(pix * NI + n) * Cd + c];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) b[j] = b[j] + cmulc(zf[zrow + k], s) + cscale(yuf[row + k], rho);
            }
        }
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + kWave * j;
            if (k < K) bsum[pc * K + k] = b[j];
        }
    }
}

template <typename T, int KR>
__global__ void __launch_bounds__(kThreads) cns_xrrs_fin_kernel(const cx<T> *__restrict__ zf,
                                                                const cx<T> *__restrict__ xf, T rho,
                                                                const cx<T> *__restrict__ bsum,
                                                                int64_t npix, int CN, int K,
                                                                double *partials, int Cd, int zch) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    const int NI = CN / Cd;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t pc = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; pc < npix * Cd;
         pc += nwaves) {
        const int64_t pix = pc / Cd;
        const int c = (int)(pc - pix * Cd);
        cx<T> a[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) a[j] = mk<T>(T(0), T(0));
        for (int n = 0; n < NI; ++n) {
            const int64_t row = ((pix * NI + n) * Cd + c) * K, zrow = zch ? row : (pix * NI + n) * K;
            cx<T> q = mk<T>(T(0), T(0));
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) q = q + cmul(zf[zrow + k], xf[row + k]);
            }
            q = wave_sum_cx(q);
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) a[j] = a[j] + cmulc(zf[zrow + k], q) + cscale(xf[row + k], rho);
            }
        }
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + kWave * j;
            if (k < K) {
                const cx<T> b = bsum[pc * K + k];
                acc[0] += (double)cabs2(a[j] - b);
                acc[1] += (double)cabs2(a[j]);
                acc[2] += (double)cabs2(b);
            }
        }
    }
    block_sum_store<3>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 3);
}

// ---- any number of rank-one terms (the dictionary update's iterated solve over more than 8
// images x channels, admm/ccmod.py:433-604): the same recursions with the per-term scalars in
// LDS (one slice per wave) and the term vectors re-read from memory instead of held in
// registers.  One wave per frequency (setup) / per system (solve), lane = filter.
template <typename T, int KR>
__global__ void __launch_bounds__(kThreads) ism_setup_big_kernel(const cx<T> *__restrict__ df,
                                                                 cx<T> *__restrict__ gam,
                                                                 cx<T> *__restrict__ del,
                                                                 cx<T> *__restrict__ mm, int64_t npix,
                                                                 int Cd, int K, T rho) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    cx<T> *dl = dyn_lds<cx<T>>() + (size_t)(threadIdx.x / kWave) * Cd;   // delta_c of this wave's frequency
    const T irho = T(1) / rho;
    for (int64_t pix = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
         pix < npix; pix += nwaves) {
        const cx<T> *d = df + pix * Cd * K;
        cx<T> *g = gam + pix * Cd * K;
        for (int c = 0; c < Cd; ++c) {
            cx<T> al[KR];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                al[j] = k < K ? cscale(cconj(d[c * K + k]), irho) : mk<T>(T(0), T(0));
            }
            for (int l = 0; l < c; ++l) {
                cx<T> t = mk<T>(T(0), T(0));
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    const int k = lane + kWave * j;
                    if (k < K) t = t + cmul(d[l * K + k], al[j]);
                }
                const cx<T> f = cdivide(wave_sum_cx(t), dl[l]);
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    const int k = lane + kWave * j;
                    if (k < K) al[j] = al[j] - cmul(g[l * K + k], f);
                }
            }
            cx<T> t = mk<T>(T(0), T(0));
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) {
                    g[c * K + k] = al[j];
                    t = t + cmul(d[c * K + k], al[j]);
                }
            }
            t = wave_sum_cx(t);
            const cx<T> dc = mk<T>(T(1) + t.re, t.im);
            dl[c] = dc;      // (every lane holds the same value and stores it: no lane waits for another)
            if (lane == 0) del[pix * Cd + c] = dc;
        }
        // (gamma of this frequency was written by this wave's own lanes, element by element the
        // lane that reads it back below)
        for (int c = 0; c < Cd; ++c)
            for (int l = 0; l < Cd; ++l) {
                cx<T> t = mk<T>(T(0), T(0));
#pragma unroll
                for (int j = 0; j < KR; ++j) {
                    const int k = lane + kWave * j;
                    if (k < K) t = t + cmul(d[c * K + k], g[l * K + k]);
                }
                t = wave_sum_cx(t);
                if (lane == 0) mm[(pix * Cd + c) * Cd + l] = t;
            }
    }
}

template <typename T, int KR>
__global__ void __launch_bounds__(kThreads) ism_solve_big_kernel(const IsmArgs<T> a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int K = a.K, Cd = a.Cd;
    const T irho = T(1) / a.rho;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    cx<T> *fw = dyn_lds<cx<T>>() + (size_t)(threadIdx.x / kWave) * 2 * Cd;   // t_c, then f_c
    const int64_t nsys = a.npix * a.N;
    for (int64_t sys = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; sys < nsys;
         sys += nwaves) {
        const int64_t pix = sys / a.N;
        const int n = (int)(sys - pix * a.N);
        const cx<T> *dp = a.df + pix * Cd * K;
        const cx<T> *gp = a.gam + pix * Cd * K;
        const cx<T> *M = a.mm + pix * Cd * Cd;
        cx<T> be[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + kWave * j;
            be[j] = k < K ? a.yuf[sys * K + k] : mk<T>(T(0), T(0));
        }
        for (int c = 0; c < Cd; ++c) {        // beta0 = yuf + sum_c conj(d_c) s_c / rho
            const cx<T> sc = a.sf[(pix * Cd + c) * a.N + n];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) be[j] = be[j] + cscale(cmulc(dp[c * K + k], sc), irho);
            }
        }
        for (int c = 0; c < Cd; ++c) {        // t_c = <ah_c, beta0>
            cx<T> t = mk<T>(T(0), T(0));
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) t = t + cmul(dp[c * K + k], be[j]);
            }
            fw[c] = wave_sum_cx(t);      // (all lanes store the same value, as above)
        }
        for (int c = 0; c < Cd; ++c) {        // f_c = (t_c - sum_{l<c} M_cl f_l) / delta_c
            cx<T> r = fw[c];
            for (int l = 0; l < c; ++l) r = r - cmul(M[c * Cd + l], fw[Cd + l]);
            fw[Cd + c] = cdivide(r, a.del[pix * Cd + c]);
        }
        for (int c = 0; c < Cd; ++c) {        // x = beta0 - sum_c gamma_c f_c
            const cx<T> f = fw[Cd + c];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) be[j] = be[j] - cmul(gp[c * K + k], f);
            }
        }
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + kWave * j;
            if (k < K) a.xf[sys * K + k] = be[j];
        }
    }
}

template <typename T, typename F> static void ism_dispatch_kr(int K, F &&f) {
    if (K <= 64) f(std::integral_constant<int, 1>{});
    else if (K <= 128) f(std::integral_constant<int, 2>{});
    else if (K <= 256) f(std::integral_constant<int, 4>{});
    else throw Error(-1, "multi-channel dictionaries are handled for K <= 256 filters");
}

template <typename T>
void launch_cns_xrrs_rhs(hipStream_t st, const cx<T> *zf, const cx<T> *sf, const cx<T> *yuf, T rho,
                         cx<T> *bsum, int64_t npix, int CN, int K, int Cd, int zch) {
    const int grid = grid_for(npix * Cd * kWave);
    ism_dispatch_kr<T>(K, [&](auto kr) {
        constexpr int KR = decltype(kr)::value;
        hipLaunchKernelGGL((cns_xrrs_rhs_kernel<T, KR>), dim3(grid), dim3(kThreads), 0, st, zf, sf, yuf, rho,
                           bsum, npix, CN, K, Cd, zch);
    });
    SA_HIP(hipGetLastError());
}
template <typename T>
int launch_cns_xrrs_fin(hipStream_t st, const cx<T> *zf, const cx<T> *xf, T rho, const cx<T> *bsum,
                        int64_t npix, int CN, int K, double *partials, int Cd, int zch) {
    const int grid = std::min(grid_for(npix * Cd * kWave), kMaxPartialBlocks);
    ism_dispatch_kr<T>(K, [&](auto kr) {
        constexpr int KR = decltype(kr)::value;
        hipLaunchKernelGGL((cns_xrrs_fin_kernel<T, KR>), dim3(grid), dim3(kThreads),
                           sizeof(double) * 3 * (kThreads / kWave), st, zf, xf, rho, bsum, npix, CN, K,
                           partials, Cd, zch);
    });
    SA_HIP(hipGetLastError());
    return grid;
}

// dst[r, b, a] = src[r, a, b]: the two inner axes of a small array swapped (a signal spectrum
// (npix, Cd, N) -> (npix, N, Cd))
template <typename T>
__global__ void __launch_bounds__(kThreads) swap_inner_kernel(const cx<T> *__restrict__ src,
                                                              cx<T> *__restrict__ dst, int64_t rows,
                                                              int A, int B) {
    const int64_t total = rows * A * B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i % B), a = (int)((i / B) % A);
        const int64_t r = i / ((int64_t)A * B);
        dst[(r * B + b) * A + a] = src[i];
    }
}
template <typename T>
void launch_swap_inner(hipStream_t st, const cx<T> *src, cx<T> *dst, int64_t rows, int A, int B) {
    hipLaunchKernelGGL((swap_inner_kernel<T>), dim3(grid_for(rows * A * B)), dim3(kThreads), 0, st, src, dst,
                       rows, A, B);
    SA_HIP(hipGetLastError());
}

// dst[(pix, c), n, k] = zch ? src[pix, n, c, k] : src[pix, n, k]: the coefficient spectrum of a
// multi-channel dictionary update seen as one single-channel update per (frequency, channel)
// -- the same matrix for every channel of a frequency (linalg.solvemdbi_ism / _cg with a
// broadcast channel axis, admm/ccmod.py:481-487), or a matrix per channel when the maps carry
// the channels themselves.
template <typename T>
__global__ void __launch_bounds__(kThreads) zf_per_channel_kernel(const cx<T> *__restrict__ src,
                                                                  cx<T> *__restrict__ dst, int64_t npix,
                                                                  int N, int Cd, int K, int zch) {
    const int64_t total = npix * Cd * N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int n = (int)((i / K) % N);
        const int c = (int)((i / ((int64_t)K * N)) % Cd);
        const int64_t pix = i / ((int64_t)K * N * Cd);
        dst[i] = zch ? src[((pix * N + n) * Cd + c) * K + k] : src[(pix * N + n) * K + k];
    }
}
template <typename T>
void launch_zf_per_channel(hipStream_t st, const cx<T> *src, cx<T> *dst, int64_t npix, int N, int Cd,
                           int K, int zch) {
    hipLaunchKernelGGL((zf_per_channel_kernel<T>), dim3(grid_for(npix * Cd * N * K)), dim3(kThreads), 0, st,
                       src, dst, npix, N, Cd, K, zch);
    SA_HIP(hipGetLastError());
}

template <typename T>
void launch_ism_setup(hipStream_t st, const cx<T> *df, cx<T> *gam, cx<T> *del, cx<T> *mm,
                      int64_t npix, int Cd, int K, T rho, const GradTerm<T> *grad, int W) {
    const int grid = grid_for(npix * kWave);
    if (Cd > 8) {
        if (grad) throw Error(-1, "the gradient-regularised iterated solve takes up to 8 channels");
        ism_dispatch_kr<T>(K, [&](auto kr) {
            constexpr int KR = decltype(kr)::value;
            hipLaunchKernelGGL((ism_setup_big_kernel<T, KR>), dim3(grid), dim3(kThreads),
                               sizeof(cx<T>) * (kThreads / kWave) * Cd, st, df, gam, del, mm, npix, Cd, K, rho);
        });
        SA_HIP(hipGetLastError());
        return;
    }
    const GradTerm<T> gt = grad ? *grad : GradTerm<T>();
    ism_dispatch_kr<T>(K, [&](auto kr) {
        constexpr int KR = decltype(kr)::value;
        hipLaunchKernelGGL((ism_setup_kernel<T, KR>), dim3(grid), dim3(kThreads), 0, st, df, gam,
                           del, mm, npix, Cd, K, rho, gt, W / 2 + 1);
    });
    SA_HIP(hipGetLastError());
}

template <typename T>
int launch_ism_solve(hipStream_t st, const cx<T> *yuf, cx<T> *xf, const cx<T> *df,
                     const cx<T> *sf, const cx<T> *gam, const cx<T> *del, const cx<T> *mm, T rho,
                     int64_t npix, int Cd, int N, int K, int W, bool want_obj, bool want_xrrs,
                     double *partials, const GradTerm<T> *grad) {
    IsmArgs<T> a;
    if (grad) a.g = *grad;
    a.yuf = yuf;
    a.xf = xf;
    a.df = df;
    a.sf = sf;
    a.gam = gam;
    a.del = del;
    a.mm = mm;
    a.rho = rho;
    a.npix = npix;
    a.Cd = Cd;
    a.N = N;
    a.K = K;
    a.W = W;
    a.want_obj = want_obj;
    a.want_xrrs = want_xrrs;
    a.partials = partials;
    if (Cd > 8) {
        if (grad || want_obj || want_xrrs)
            throw Error(-1, "more than 8 rank-one terms: the plain solve only (the dictionary update)");
        const int gridb = grid_for(npix * N * kWave);
        ism_dispatch_kr<T>(K, [&](auto kr) {
            constexpr int KR = decltype(kr)::value;
            hipLaunchKernelGGL((ism_solve_big_kernel<T, KR>), dim3(gridb), dim3(kThreads),
                               sizeof(cx<T>) * (kThreads / kWave) * 2 * Cd, st, a);
        });
        SA_HIP(hipGetLastError());
        return 0;
    }
    const int ppb = (kThreads / kWave) / ism_waves_per_pixel(N, kThreads / kWave);
    const int grid = (int)std::min<int64_t>(ceil_div(npix, (int64_t)ppb), kMaxPartialBlocks);
    const size_t lds = sizeof(double) * 5 * (kThreads / kWave);
    ism_dispatch_kr<T>(K, [&](auto kr) {
        constexpr int KR = decltype(kr)::value;
        if (grad) {
            // (one instantiation with a run-time channel count: not a path measured in it/s)
            hipLaunchKernelGGL((ism_solve_kernel<T, KR, 0, true>), dim3(grid), dim3(kThreads), lds, st, a);
            return;
        }
        switch (Cd) {
        case 2: hipLaunchKernelGGL((ism_solve_kernel<T, KR, 2, false>), dim3(grid), dim3(kThreads), lds, st, a); break;
        case 3: hipLaunchKernelGGL((ism_solve_kernel<T, KR, 3, false>), dim3(grid), dim3(kThreads), lds, st, a); break;
        case 4: hipLaunchKernelGGL((ism_solve_kernel<T, KR, 4, false>), dim3(grid), dim3(kThreads), lds, st, a); break;
        default: hipLaunchKernelGGL((ism_solve_kernel<T, KR, 0, false>), dim3(grid), dim3(kThreads), lds, st, a); break;
        }
    });
    SA_HIP(hipGetLastError());
    return grid;
}

// PGM gradient for a multi-channel dictionary (pgm/cbpdn.py:263-279):
// gf[pix, n, k] = sum_c conj(df[pix, c, k]) (sum_m df[pix, c, m] v[pix, n, m] - sf[pix, c, n]);
// one wave per (pix, n) system, lane = filter.  Partials (2): sum_c |r_c|^2 unweighted and
// Parseval-weighted.
template <typename T, int KR>
__global__ void __launch_bounds__(kThreads) mc_pgm_grad_kernel(const cx<T> *__restrict__ v,
                                                               const cx<T> *__restrict__ df,
                                                               const cx<T> *__restrict__ sf,
                                                               cx<T> *__restrict__ gf, int64_t npix,
                                                               int Cd, int N, int K, int W,
                                                               double *partials) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    const int Wf = W / 2 + 1;
    const int64_t nsys = npix * N;
    double acc[2] = {0.0, 0.0};
    for (int64_t sys = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
         sys < nsys; sys += nwaves) {
        const int64_t pix = sys / N;
        const int n = (int)(sys - pix * N);
        const cx<T> *d = df + pix * Cd * K;
        cx<T> x[KR], g[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + kWave * j;
            x[j] = k < K ? v[sys * K + k] : mk<T>(T(0), T(0));
            g[j] = mk<T>(T(0), T(0));
        }
        for (int c = 0; c < Cd; ++c) {
            cx<T> t = mk<T>(T(0), T(0));
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) t = t + cmul(d[c * K + k], x[j]);
            }
            const cx<T> r = wave_sum_cx(t) - sf[(pix * Cd + c) * N + n];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int k = lane + kWave * j;
                if (k < K) g[j] = g[j] + cmulc(d[c * K + k], r);
            }
            if (lane == 0) {
                const double r2 = (double)cabs2(r);
                acc[0] += r2;
                acc[1] += parseval_weight((int)(pix % Wf), Wf, W) * r2;
            }
        }
#pragma unroll
        for (int j = 0; j < KR; ++j) {
            const int k = lane + kWave * j;
            if (k < K) gf[sys * K + k] = g[j];
        }
    }
    block_sum_store<2>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 2);
}

template <typename T>
int launch_mc_pgm_grad(hipStream_t st, const cx<T> *v, const cx<T> *df, const cx<T> *sf, cx<T> *gf,
                       int64_t npix, int Cd, int N, int K, int W, double *partials) {
    const int grid = grid_for(npix * N * kWave);
    const size_t lds = sizeof(double) * 2 * (kThreads / kWave);
    ism_dispatch_kr<T>(K, [&](auto kr) {
        constexpr int KR = decltype(kr)::value;
        hipLaunchKernelGGL((mc_pgm_grad_kernel<T, KR>), dim3(grid), dim3(kThreads), lds, st, v, df, sf,
                           gf, npix, Cd, N, K, W, partials);
    });
    SA_HIP(hipGetLastError());
    return grid;
}

// out[pix, c, n] = sum_k df[pix, c, k] v[pix, n, k]: linalg.inner over the filter axis for a
// multi-channel dictionary (the Cd = 1 case is launch_inner)
template <typename T, int L>    // (L lanes per output, as inner_kernel)
__global__ void __launch_bounds__(kThreads) mc_inner_kernel(const cx<T> *__restrict__ df,
                                                            const cx<T> *__restrict__ v,
                                                            cx<T> *__restrict__ out, int64_t npix,
                                                            int Cd, int N, int K, int vch) {
    // (vch: v has a channel axis of its own, (npix, N, Cd, K))
    const int64_t total = npix * Cd * N;
    const int sub = threadIdx.x % L;
    const int64_t per_blk = blockDim.x / L;
    for (int64_t base = (int64_t)blockIdx.x * per_blk; base < total; base += (int64_t)gridDim.x * per_blk) {
        const int64_t i = base + threadIdx.x / L;
        cx<T> q = mk<T>(T(0), T(0));
        if (i < total) {
            const int n = (int)(i % N);
            const int c = (int)((i / N) % Cd);
            const int64_t pix = i / ((int64_t)N * Cd);
            const cx<T> *d = df + (pix * Cd + c) * K;
            const cx<T> *x = v + (vch ? (pix * N + n) * Cd + c : pix * N + n) * K;
            for (int k = sub; k < K; k += L) q = q + cmul(d[k], x[k]);
        }
        if (L > 1) {
#pragma unroll
            for (int m = L / 2; m >= 1; m >>= 1) {
                q.re += __shfl_xor(q.re, m, kWave);
                q.im += __shfl_xor(q.im, m, kWave);
            }
        }
        if (i < total && sub == 0) out[i] = q;
    }
}

template <typename T>
void launch_mc_inner(hipStream_t st, const cx<T> *df, const cx<T> *v, cx<T> *out, int64_t npix,
                     int Cd, int N, int K, int vch) {
    if (K >= 16) {
        hipLaunchKernelGGL((mc_inner_kernel<T, 16>), dim3(grid_for(npix * Cd * N * 16)), dim3(kThreads), 0,
                           st, df, v, out, npix, Cd, N, K, vch);
    } else {
        hipLaunchKernelGGL((mc_inner_kernel<T, 1>), dim3(grid_for(npix * Cd * N)), dim3(kThreads), 0, st,
                           df, v, out, npix, Cd, N, K, vch);
    }
    SA_HIP(hipGetLastError());
}

// gf[pix, n, k] (+)= sum_c conj(df[pix, c, k]) r[pix, c, n]: the adjoint of mc_inner (A_0^T of the
// mask-decoupling constraint with a multi-channel dictionary, cbpdn.py:1762-1770; `add`: onto gf)
template <typename T>
__global__ void __launch_bounds__(kThreads) mc_conj_outer_kernel(const cx<T> *__restrict__ df,
                                                                 const cx<T> *__restrict__ r,
                                                                 cx<T> *gf, int64_t npix, int Cd,
                                                                 int N, int K, int add) {
    const int64_t total = npix * N * K;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const int n = (int)((i / K) % N);
        const int64_t pix = i / ((int64_t)N * K);
        cx<T> g = add ? gf[i] : mk<T>(T(0), T(0));
        for (int c = 0; c < Cd; ++c)
            g = g + cmulc(df[(pix * Cd + c) * K + k], r[(pix * Cd + c) * N + n]);
        gf[i] = g;
    }
}

template <typename T>
void launch_mc_conj_outer(hipStream_t st, const cx<T> *df, const cx<T> *r, cx<T> *gf, int64_t npix,
                          int Cd, int N, int K, bool add) {
    hipLaunchKernelGGL((mc_conj_outer_kernel<T>), dim3(grid_for(npix * N * K)), dim3(kThreads), 0, st,
                       df, r, gf, npix, Cd, N, K, add ? 1 : 0);
    SA_HIP(hipGetLastError());
}

// max |conj(df[pix, c, k]) sf[pix, c, n]| (cbpdn.py:573-578 without the channel sum)
template <typename T>
__global__ void __launch_bounds__(kThreads) mc_dhs_absmax_kernel(const cx<T> *__restrict__ df,
                                                                 const cx<T> *__restrict__ sf,
                                                                 int64_t npix, int Cd, int N,
                                                                 int K, double *partials) {
    double m = 0.0;
    const int64_t total = npix * Cd * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pc = i / N;   // (pix, c)
        const double s2 = (double)cabs2(sf[i]);
        for (int k = 0; k < K; ++k) {
            const double v = (double)cabs2(df[pc * K + k]) * s2;
            m = v > m ? v : m;
        }
    }
    double *scratch = dyn_lds<double>();
    for (int s = kWave / 2; s > 0; s >>= 1) {
        const double o = __shfl_xor(m, s, kWave);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) scratch[threadIdx.x / kWave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int j = 0; j < (int)(blockDim.x / kWave); ++j) r = scratch[j] > r ? scratch[j] : r;
        partials[blockIdx.x] = r;
    }
}

template <typename T>
int launch_mc_dhs_absmax(hipStream_t st, const cx<T> *df, const cx<T> *sf, int64_t npix, int Cd,
                         int N, int K, double *partials) {
    const int grid = grid_for(npix * Cd * N);
    hipLaunchKernelGGL((mc_dhs_absmax_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, df, sf, npix, Cd, N, K, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

template <typename T>
__global__ void __launch_bounds__(kThreads) pcn_stats_kernel(const T *__restrict__ v,
                                                             T *__restrict__ stats, int H, int W,
                                                             int K, int dH_, int dW_, int zm, int Cd,
                                                             FilterSizes fs) {
    // v is (H, W, Cd, K); mean per (channel, filter) over the support (cnvrep.zeromean,
    // cnvrep.py:609-670), norm per filter over support and channels (cnvrep.normalise with
    // dimN + dimC axes, cnvrep.py:696-700).  stats[2 (c K + k)] = mean, stats[2k + 1] = 1/norm.
    // One wave per filter, the lanes share the support.
    const int lane = threadIdx.x & (kWave - 1);
    const int nwaves = gridDim.x * (blockDim.x / kWave);
    for (int k = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; k < K; k += nwaves) {
        // (multi-scale dictionary: every filter has its own support, cnvrep.py:634-662, :778-812)
        const int dH = fs.h ? fs.h[k] : dH_, dW = fs.w ? fs.w[k] : dW_;
        const int np = dH * dW;
        T n2 = T(0);
        for (int c = 0; c < Cd; ++c) {
            T mean = T(0);
            if (zm) {
                T s = T(0);
                for (int i = lane; i < np; i += kWave)
                    s += v[(((int64_t)(i / dW) * W + i % dW) * Cd + c) * K + k];
                mean = (T)(wave_sum((double)s) / (double)np);
            }
            for (int i = lane; i < np; i += kWave) {
                const T e = v[(((int64_t)(i / dW) * W + i % dW) * Cd + c) * K + k] - mean;
                n2 += e * e;
            }
            if (lane == 0) stats[2 * (c * K + k)] = mean;
        }
        const T nrm = (T)sqrt(wave_sum((double)n2));
        if (lane == 0) stats[2 * k + 1] = nrm == T(0) ? T(1) : T(1) / nrm;
    }
}

template <typename T>
void launch_pcn_stats(hipStream_t st, const T *v, T *stats, int H, int W, int K, int dH, int dW,
                      bool zm, int Cd, FilterSizes fs) {
    hipLaunchKernelGGL((pcn_stats_kernel<T>), dim3(grid_for((int64_t)K * kWave)), dim3(kThreads), 0, st, v,
                       stats, H, W, K, dH, dW, zm ? 1 : 0, Cd, fs);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) pcn_apply_kernel(const T *__restrict__ v,
                                                             const T *__restrict__ stats, T *out,
                                                             int H, int W, int K, int dH_, int dW_,
                                                             int Kvalid, int Cd, double *partials,
                                                             FilterSizes fs) {
    double acc[1] = {0.0};
    const int64_t KD = (int64_t)Cd * K;
    const int64_t n = (int64_t)H * W * KD;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ck = (int)(i % KD);
        const int k = ck % K;
        const int64_t pix = i / KD;
        const int x = (int)(pix % W), h = (int)(pix / W);
        const int dH = fs.h ? fs.h[k] : dH_, dW = fs.w ? fs.w[k] : dW_;
        const T vi = v[i];
        // v / vn as in cnvrep.normalise (cnvrep.py:696-700): 1/norm is applied by division
        // (filters >= Kvalid are the handle's zero padding: rounding noise must not be
        // normalised up to a unit-norm filter)
        const T o = (h < dH && x < dW && k < Kvalid) ? (vi - stats[2 * ck]) * stats[2 * k + 1] : T(0);
        if (out) out[i] = o;
        const double df = (double)(o - vi);
        acc[0] += df * df;
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + blockIdx.x);
}

template <typename T>
int launch_pcn_apply(hipStream_t st, const T *v, const T *stats, T *out, int H, int W, int K,
                     int dH, int dW, double *partials, int Kvalid, int Cd, FilterSizes fs) {
    const int grid = grid_for((int64_t)H * W * Cd * K);
    hipLaunchKernelGGL((pcn_apply_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, v, stats, out, H, W, K, dH, dW,
                       Kvalid < 0 ? K : Kvalid, Cd, partials, fs);
    SA_HIP(hipGetLastError());
    return grid;
}

template <typename T>
__global__ void __launch_bounds__(kThreads) asum_kernel(const T *__restrict__ v, int64_t n,
                                                        double *partials) {
    double acc[1] = {0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const T x = v[i];
        acc[0] += (double)(x < T(0) ? -x : x);
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + blockIdx.x);
}

template <typename T> int launch_asum(hipStream_t st, const T *v, int64_t n, double *partials) {
    const int grid = grid_for(n);
    hipLaunchKernelGGL((asum_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * (kThreads / kWave), st, v, n, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// ---------------------------------------------------------------------------
// fixed-order final reduction of block partials
// ---------------------------------------------------------------------------
struct FinalizeGroup {
    const double *partials;
    int nblocks, stride, nvals;
    int slots[8];
    double scales[8];
};
struct FinalizeArgs {
    FinalizeGroup g[2];
    int ngroups, is_max;
    double *out;
};

// One workgroup per output value: thread t sums blocks t, t+256, ... of its value,
// then a fixed-shape LDS tree (same order on every run).
__global__ void __launch_bounds__(kThreads) finalize_kernel(const FinalizeArgs a) {
    double *scratch = dyn_lds<double>();
    int i = blockIdx.x;
    const FinalizeGroup *gp = &a.g[0];
    if (i >= a.g[0].nvals) {
        i -= a.g[0].nvals;
        gp = &a.g[1];
    }
    const FinalizeGroup &g = *gp;
    double s = 0.0;
    for (int b = threadIdx.x; b < g.nblocks; b += blockDim.x) {
        const double v = g.partials[(int64_t)b * g.stride + i];
        s = a.is_max ? (v > s ? v : s) : s + v;
    }
    scratch[threadIdx.x] = s;
    __syncthreads();
    for (int w = blockDim.x / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            const double o = scratch[threadIdx.x + w];
            scratch[threadIdx.x] = a.is_max ? (o > scratch[threadIdx.x] ? o : scratch[threadIdx.x])
                                            : scratch[threadIdx.x] + o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) a.out[g.slots[i]] = scratch[0] * g.scales[i];
}

static void fill_group(FinalizeGroup &g, const double *partials, int nblocks, int stride, int nvals,
                       const int *slots, const double *scales) {
    g.partials = partials;
    g.nblocks = nblocks;
    g.stride = stride;
    g.nvals = nvals;
    for (int i = 0; i < 8; ++i) {
        g.slots[i] = i < nvals ? slots[i] : 0;
        g.scales[i] = i < nvals ? scales[i] : 0.0;
    }
}

void launch_finalize(hipStream_t st, const double *partials, int nblocks, int stride, int nvals,
                     const int *slots, const double *scales, bool is_max, double *out) {
    FinalizeArgs a;
    fill_group(a.g[0], partials, nblocks, stride, nvals, slots, scales);
    fill_group(a.g[1], partials, 0, 1, 0, slots, scales);
    a.ngroups = 1;
    a.is_max = is_max ? 1 : 0;
    a.out = out;
    if (nvals <= 0) return;
    hipLaunchKernelGGL(finalize_kernel, dim3(nvals), dim3(kThreads), sizeof(double) * kThreads, st, a);
    SA_HIP(hipGetLastError());
}

void launch_finalize2(hipStream_t st, const double *pa, int nblocks_a, int stride_a, int nvals_a,
                      const int *slots_a, const double *scales_a, const double *pb, int nblocks_b,
                      int stride_b, int nvals_b, const int *slots_b, const double *scales_b,
                      double *out) {
    FinalizeArgs a;
    fill_group(a.g[0], pa, nblocks_a, stride_a, nvals_a, slots_a, scales_a);
    fill_group(a.g[1], pb, nblocks_b, stride_b, nvals_b, slots_b, scales_b);
    a.ngroups = 2;
    a.is_max = 0;
    a.out = out;
    if (nvals_a + nvals_b <= 0) return;
    hipLaunchKernelGGL(finalize_kernel, dim3(nvals_a + nvals_b), dim3(kThreads),
                       sizeof(double) * kThreads, st, a);
    SA_HIP(hipGetLastError());
}

// ---------------------------------------------------------------------------
// pre / post-processing on device arrays (csc_kernels.h)
// ---------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kThreads) sympad_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                          int H, int W, int64_t P, int npd) {
    const int Hp = H + 2 * npd, Wp = W + 2 * npd;
    const int64_t total = (int64_t)Hp * Wp * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % P, pix = i / P;
        int w = (int)(pix % Wp) - npd, h = (int)(pix / Wp) - npd;
        // 'symmetric': reflect about the edge, edge sample repeated; period 2n
        auto refl = [](int v, int n) {
            const int m = 2 * n;
            v = ((v % m) + m) % m;
            return v < n ? v : m - 1 - v;
        };
        h = refl(h, H);
        w = refl(w, W);
        out[i] = in[((int64_t)h * W + w) * P + p];
    }
}
template <typename T>
void launch_sympad(hipStream_t st, const T *in, T *out, int H, int W, int64_t P, int npd) {
    const int64_t total = (int64_t)(H + 2 * npd) * (W + 2 * npd) * P;
    hipLaunchKernelGGL((sympad_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, st, in, out, H, W,
                       P, npd);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) tikhonov_divide_kernel(cx<T> *__restrict__ spf, int Hp,
                                                                   int Wp, int64_t P, double lmbda) {
    const int Wf = Wp / 2 + 1;
    const int64_t total = (int64_t)Hp * Wf * P;
    const double two_pi = 6.283185307179586476925286766559;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / P;
        const int wf = (int)(pix % Wf), h = (int)(pix / Wf);
        const double a = 1.0 + lmbda * ((2.0 - 2.0 * cos(two_pi * h / Hp)) +
                                        (2.0 - 2.0 * cos(two_pi * wf / Wp)));
        const cx<T> v = spf[i];
        spf[i] = mk<T>((T)((double)v.re / a), (T)((double)v.im / a));
    }
}
template <typename T>
void launch_tikhonov_divide(hipStream_t st, cx<T> *spf, int Hp, int Wp, int64_t P, double lmbda) {
    const int64_t total = (int64_t)Hp * (Wp / 2 + 1) * P;
    hipLaunchKernelGGL((tikhonov_divide_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, st, spf,
                       Hp, Wp, P, lmbda);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) crop_highpass_kernel(const T *__restrict__ sp,
                                                                 const T *__restrict__ s,
                                                                 T *__restrict__ slp,
                                                                 T *__restrict__ shp, int H, int W,
                                                                 int64_t P, int npd) {
    const int Wp = W + 2 * npd;
    const int64_t total = (int64_t)H * W * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % P, pix = i / P;
        const int w = (int)(pix % W), h = (int)(pix / W);
        const T lo = sp[((int64_t)(h + npd) * Wp + (w + npd)) * P + p];
        slp[i] = lo;
        shp[i] = s[i] - lo;
    }
}
template <typename T>
void launch_crop_highpass(hipStream_t st, const T *sp, const T *s, T *slp, T *shp, int H, int W,
                          int64_t P, int npd) {
    hipLaunchKernelGGL((crop_highpass_kernel<T>), dim3(grid_for((int64_t)H * W * P)), dim3(kThreads), 0,
                       st, sp, s, slp, shp, H, W, P, npd);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) zeropad2_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                            int h, int w, int H, int W, int64_t P) {
    const int64_t total = (int64_t)H * W * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % P, pix = i / P;
        const int x = (int)(pix % W), y = (int)(pix / W);
        out[i] = (y < h && x < w) ? in[((int64_t)y * w + x) * P + p] : T(0);
    }
}
template <typename T>
void launch_zeropad2(hipStream_t st, const T *in, T *out, int h, int w, int H, int W, int64_t P) {
    hipLaunchKernelGGL((zeropad2_kernel<T>), dim3(grid_for((int64_t)H * W * P)), dim3(kThreads), 0, st,
                       in, out, h, w, H, W, P);
    SA_HIP(hipGetLastError());
}

struct Bcast3 {
    int64_t d[3], sa[3], sb[3], pa, pb;
};
template <typename T>
__global__ void __launch_bounds__(kThreads) cmul_bcast_kernel(const cx<T> *__restrict__ a,
                                                              const cx<T> *__restrict__ b,
                                                              cx<T> *__restrict__ out, int64_t npix,
                                                              const Bcast3 bc) {
    const int64_t po = bc.d[0] * bc.d[1] * bc.d[2];
    const int64_t total = npix * po;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / po;
        int64_t r = i - pix * po;
        const int64_t i2 = r % bc.d[2];
        r /= bc.d[2];
        const int64_t i1 = r % bc.d[1], i0 = r / bc.d[1];
        const cx<T> x = a[pix * bc.pa + i0 * bc.sa[0] + i1 * bc.sa[1] + i2 * bc.sa[2]];
        const cx<T> y = b[pix * bc.pb + i0 * bc.sb[0] + i1 * bc.sb[1] + i2 * bc.sb[2]];
        out[i] = cmul(x, y);
    }
}
template <typename T>
void launch_cmul_bcast(hipStream_t st, const cx<T> *a, const cx<T> *b, cx<T> *out, int64_t npix,
                       const int64_t d[3], const int64_t sa[3], const int64_t sb[3], int64_t pa,
                       int64_t pb) {
    Bcast3 bc;
    for (int i = 0; i < 3; ++i) {
        bc.d[i] = d[i];
        bc.sa[i] = sa[i];
        bc.sb[i] = sb[i];
    }
    bc.pa = pa;
    bc.pb = pb;
    hipLaunchKernelGGL((cmul_bcast_kernel<T>), dim3(grid_for(npix * d[0] * d[1] * d[2])), dim3(kThreads),
                       0, st, a, b, out, npix, bc);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) roll2_kernel(const T *__restrict__ in, T *__restrict__ out,
                                                         int H, int W, int64_t P, int oh, int ow) {
    const int64_t total = (int64_t)H * W * P;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % P, pix = i / P;
        const int w = (int)(pix % W), h = (int)(pix / W);
        const int hs = (((h + oh) % H) + H) % H, ws = (((w + ow) % W) + W) % W;
        out[i] = in[((int64_t)hs * W + ws) * P + p];
    }
}
template <typename T>
void launch_roll2(hipStream_t st, const T *in, T *out, int H, int W, int64_t P, int oh, int ow) {
    hipLaunchKernelGGL((roll2_kernel<T>), dim3(grid_for((int64_t)H * W * P)), dim3(kThreads), 0, st, in,
                       out, H, W, P, oh, ow);
    SA_HIP(hipGetLastError());
}

template <typename T>
__global__ void __launch_bounds__(kThreads) axpby_kernel(T a, const T *__restrict__ x, T b,
                                                         const T *__restrict__ y, T *__restrict__ out,
                                                         int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = y ? a * x[i] + b * y[i] : a * x[i];
}
template <typename T>
void launch_axpby(hipStream_t st, T a, const T *x, T b, const T *y, T *out, int64_t n) {
    hipLaunchKernelGGL((axpby_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, a, x, b, y, out, n);
    SA_HIP(hipGetLastError());
}
#define SA_INST_PREPOST(T)                                                                          \
    template void launch_sympad<T>(hipStream_t, const T *, T *, int, int, int64_t, int);            \
    template void launch_tikhonov_divide<T>(hipStream_t, cx<T> *, int, int, int64_t, double);       \
    template void launch_crop_highpass<T>(hipStream_t, const T *, const T *, T *, T *, int, int,    \
                                          int64_t, int);                                            \
    template void launch_zeropad2<T>(hipStream_t, const T *, T *, int, int, int, int, int64_t);     \
    template void launch_cmul_bcast<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t, \
                                       const int64_t[3], const int64_t[3], const int64_t[3],        \
                                       int64_t, int64_t);                                           \
    template void launch_roll2<T>(hipStream_t, const T *, T *, int, int, int64_t, int, int);        \
    template void launch_axpby<T>(hipStream_t, T, const T *, T, const T *, T *, int64_t);
SA_INST_PREPOST(float)
SA_INST_PREPOST(double)

// ---------------------------------------------------------------------------
// conjugate gradients with device-side scalars (csc_kernels.h)
// ---------------------------------------------------------------------------
__global__ void cg_init_kernel(CgCtl *c, CgPinned *pin, double atol, int maxit) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    c->rr = c->rr_prev = c->pq = 0.0;
    c->atol = atol;
    c->alpha = c->beta = 0.0;
    c->done = c->it = 0;
    c->rr2[0] = c->rr2[1] = 0.0;
    c->info = maxit;
    c->maxit = maxit;
    (void)pin;     // reset by the host before this launch (csc_api.hip dstep_iter)
}
void launch_cg_init(hipStream_t st, CgCtl *c, CgPinned *pin, double atol, int maxit) {
    hipLaunchKernelGGL(cg_init_kernel, dim3(1), dim3(64), 0, st, c, pin, atol, maxit);
    SA_HIP(hipGetLastError());
}

// The scalar step of CG on the sums of the block partials (csc_kernels.h CgCtl), run by one
// whole workgroup: thread t adds rows t, t + 256, ..., then a fixed-shape tree (the order of
// launch_finalize, which the host-driven loop reads its sums through).
template <typename T, int PHASE>
__device__ __forceinline__ void cg_scalar_step(const double *partials, int nb, CgCtl *c, CgPinned *pin,
                                               double *cgout, double *scratch) {
#pragma clang fp contract(off)
    constexpr int idx = PHASE == 0 ? 2 : 1;
    // (kThreads summation slots whatever the size of this workgroup)
    for (int t = threadIdx.x; t < kThreads; t += blockDim.x) {
        double s = 0.0;
        for (int b = t; b < nb; b += kThreads) s += partials[(int64_t)b * 4 + idx];
        scratch[t] = s;
    }
    __syncthreads();
    for (int w = kThreads / 2; w > 0; w >>= 1) {
        for (int t = threadIdx.x; t < w; t += blockDim.x) scratch[t] += scratch[t + w];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const double v = scratch[0];
    if (PHASE == 0) {
        int done = 0, info = c->maxit;
        if (c->it >= c->maxit) {
            done = 1;
        } else {
            c->rr = v;
            if (sqrt(v) < c->atol) {
                done = 1;
                info = 0;
            } else {
                c->beta = c->it == 0 ? 0.0 : (double)(T)(v / c->rr_prev);
            }
        }
        if (done) {
            c->done = 1;
            c->info = info;
            cgout[0] = (double)info;
            cgout[1] = (double)c->it;
            pin->done = 1;
            pin->it = c->it;
            pin->info = info;
        }
        sa_fence_system();
        pin->seq = pin->seq + 1;
        sa_fence_system();
    } else {
        c->pq = v;
        c->alpha = (double)(T)(c->rr / v);
        c->rr_prev = c->rr;
        c->it = c->it + 1;
    }
}
template <typename T, int PHASE>
__global__ void __launch_bounds__(kThreads) cg_ctl_kernel(const double *partials, int nb, CgCtl *c,
                                                          CgPinned *pin, double *cgout) {
    if (c->done) return;
    cg_scalar_step<T, PHASE>(partials, nb, c, pin, cgout, dyn_lds<double>());
}
template <typename T>
void launch_cg_ctl(hipStream_t st, int phase, const double *partials, int nb, CgCtl *c, CgPinned *pin,
                   double *cgout) {
    if (phase == 0)
        hipLaunchKernelGGL((cg_ctl_kernel<T, 0>), dim3(1), dim3(kThreads), sizeof(double) * kThreads, st,
                           partials, nb, c, pin, cgout);
    else
        hipLaunchKernelGGL((cg_ctl_kernel<T, 1>), dim3(1), dim3(kThreads), sizeof(double) * kThreads, st,
                           partials, nb, c, pin, cgout);
    SA_HIP(hipGetLastError());
}

// (T)(a / b) as the scalar steps form alpha and beta: a float64 quotient rounded to T
template <typename T> __device__ __forceinline__ T cg_ratio(double a, double b) {
#pragma clang fp contract(off)
    return (T)(a / b);
}
// Sum of column `idx` of `nb` partial rows in the order of launch_finalize, delivered to every
// thread of the workgroup (CgSelf).  scratch: kThreads doubles.
__device__ __forceinline__ double cg_rows_sum(const double *partials, int nb, int idx, double *scratch) {
    for (int t = threadIdx.x; t < kThreads; t += blockDim.x) {
        double s = 0.0;
        for (int b = t; b < nb; b += kThreads) s += partials[(int64_t)b * 4 + idx];
        scratch[t] = s;
    }
    __syncthreads();
    for (int w = kThreads / 2; w > 0; w >>= 1) {
        for (int t = threadIdx.x; t < w; t += blockDim.x) scratch[t] += scratch[t + w];
        __syncthreads();
    }
    const double v = scratch[0];
    __syncthreads();
    return v;
}

// (the CG kernels: grid-stride over at most four workgroups per CU)
// (1024 workgroups: four per CU; measured 411 outer it/s at the bench shape against 393 with 2048
// and 388 with 512 -- every workgroup sums the partial rows of the preceding launch, CgSelf)
constexpr int kCgMaxBlocks = 1024;
template <typename T, int JM>     // JM filters per lane: K <= 64 JM
__global__ void __launch_bounds__(kThreads) cg_op_kernel(const CgCtl *ctl, int with_update,
                                                         const cx<T> *__restrict__ zf,
                                                         const cx<T> *__restrict__ r,
                                                         cx<T> *__restrict__ p, cx<T> *__restrict__ q,
                                                         T rho, int64_t npix, int CN, int K,
                                                         double *partials, const CgSelf self) {
    if (ctl && ctl->done) return;
    constexpr int NB = 4;                       // images whose spectra are in flight together
    const int lane = threadIdx.x & (kWave - 1);
    const int wpb = blockDim.x / kWave;
    T beta = (ctl && with_update) ? (T)ctl->beta : T(0);
    if (self.c) {
        // top of CG iteration `iter` (cg_scalar_step, phase 0): <r, r>, stopping test, beta
        CgCtl *c = self.c;
        const double rr = cg_rows_sum(self.prev, self.prev_nb, 2, dyn_lds<double>());
        const bool out_of_iters = self.iter >= c->maxit;
        const bool converged = !out_of_iters && sqrt(rr) < c->atol;
        beta = (self.iter == 0 || out_of_iters || converged) ? T(0)
                                                               : cg_ratio<T>(rr, c->rr2[(self.iter - 1) & 1]);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (!out_of_iters) c->rr2[self.iter & 1] = rr;
            if (out_of_iters || converged) {
                const int info = converged ? 0 : c->maxit;
                c->done = 1;
                c->info = info;
                c->it = self.iter;
                self.cgout[0] = (double)info;
                self.cgout[1] = (double)self.iter;
                self.pin->done = 1;
                self.pin->it = self.iter;
                self.pin->info = info;
            }
            sa_fence_system();
            self.pin->seq = self.iter + 1;
            sa_fence_system();
        }
        if (out_of_iters || converged) return;
    }
    double acc[1] = {0.0};                      // <p, q>, slot 1 of the block's partial row
    for (int64_t pix = (int64_t)blockIdx.x * wpb + threadIdx.x / kWave; pix < npix;
         pix += (int64_t)gridDim.x * wpb) {
        cx<T> pk[JM], qk[JM];
#pragma unroll
        for (int j = 0; j < JM; ++j) {
            const int k = lane + kWave * j;
            pk[j] = mk<T>(T(0), T(0));
            qk[j] = mk<T>(T(0), T(0));
            if (k < K) {
                if (with_update) {
                    const cx<T> rv = r[pix * K + k];
                    pk[j] = beta == T(0) ? rv : cscale(rv, T(1)) + cscale(p[pix * K + k], beta);
                    p[pix * K + k] = pk[j];
                } else {
                    pk[j] = p[pix * K + k];
                }
            }
        }
        for (int n0 = 0; n0 < CN; n0 += NB) {
            cx<T> zk[NB][JM], t[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const bool have = n0 + b < CN;
                const cx<T> *zrow = zf + (pix * CN + (have ? n0 + b : n0)) * K;
#pragma unroll
                for (int j = 0; j < JM; ++j) {
                    const int k = lane + kWave * j;
                    zk[b][j] = (have && k < K) ? zrow[k] : mk<T>(T(0), T(0));
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                t[b] = mk<T>(T(0), T(0));
#pragma unroll
                for (int j = 0; j < JM; ++j) t[b] = t[b] + cmul(zk[b][j], pk[j]);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) sa_wave_allreduce2(t[b].re, t[b].im);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int j = 0; j < JM; ++j) qk[j] = qk[j] + cmulc(zk[b][j], t[b]);
            }
        }
#pragma unroll
        for (int j = 0; j < JM; ++j) {
            const int k = lane + kWave * j;
            if (k < K) {
                const cx<T> qq = cscale(qk[j], T(1)) + cscale(pk[j], rho);
                q[pix * K + k] = qq;
                acc[0] += (double)pk[j].re * (double)qq.re + (double)pk[j].im * (double)qq.im;
            }
        }
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 4 + 1);
}
template <typename T>
int launch_cg_op(hipStream_t st, const CgCtl *ctl, bool with_update, const cx<T> *zf,
                 const cx<T> *r, cx<T> *p, cx<T> *q, T rho, int64_t npix, int CN, int K,
                 double *partials, const CgSelf &self) {
    SA_REQUIRE(K <= 4 * kWave, "cg_op: at most 256 filters");
#ifdef SPORCO_AMD_HOSTSIM
    const int threads = kWave;                    // (the simulator's scheduler walks the whole block)
    const int cap = kMaxPartialBlocks;
#else
    const int threads = kThreads;
    const int cap = kCgMaxBlocks;
#endif
    // one wave per pixel, grid-stride beyond the cap on workgroups
    const int grid = std::min(grid_for(npix * kWave, threads), cap);
    const size_t lds = sizeof(double) * kThreads;
    const int wu = with_update ? 1 : 0;
    if (K <= kWave)
        hipLaunchKernelGGL((cg_op_kernel<T, 1>), dim3(grid), dim3(threads), lds, st, ctl, wu, zf, r, p, q,
                           rho, npix, CN, K, partials, self);
    else if (K <= 2 * kWave)
        hipLaunchKernelGGL((cg_op_kernel<T, 2>), dim3(grid), dim3(threads), lds, st, ctl, wu, zf, r, p, q,
                           rho, npix, CN, K, partials, self);
    else
        hipLaunchKernelGGL((cg_op_kernel<T, 4>), dim3(grid), dim3(threads), lds, st, ctl, wu, zf, r, p, q,
                           rho, npix, CN, K, partials, self);
    SA_HIP(hipGetLastError());
    return grid;
}

template <typename T>
__global__ void __launch_bounds__(kThreads) cg_update_p_kernel(const CgCtl *c, const cx<T> *__restrict__ r,
                                                               cx<T> *__restrict__ p, int64_t n) {
    if (c->done) return;
    const T beta = (T)c->beta;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        // (lincomb form of the host-driven loop: 1 r + beta p)
        p[i] = beta == T(0) ? r[i] : cscale(r[i], T(1)) + cscale(p[i], beta);
    }
}
template <typename T>
void launch_cg_update_p(hipStream_t st, const CgCtl *c, const cx<T> *r, cx<T> *p, int64_t n) {
    hipLaunchKernelGGL((cg_update_p_kernel<T>), dim3(grid_for(n)), dim3(kThreads), 0, st, c, r, p, n);
    SA_HIP(hipGetLastError());
}
template <typename T>
__global__ void __launch_bounds__(kThreads) cg_update_xr_kernel(const CgCtl *c, T alpha_host,
                                                                cx<T> *__restrict__ x,
                                                                cx<T> *__restrict__ r,
                                                                const cx<T> *__restrict__ p,
                                                                const cx<T> *__restrict__ q, int64_t n,
                                                                double *partials, const CgSelf self) {
    if (c && c->done) return;
    T alpha = c ? (T)c->alpha : alpha_host;
    if (self.c) {
        // (cg_scalar_step, phase 1): <p, q> of the operator launch before this one, alpha
        const double pq = cg_rows_sum(self.prev, self.prev_nb, 1, dyn_lds<double>());
        alpha = cg_ratio<T>(self.c->rr2[self.iter & 1], pq);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            self.c->pq = pq;
            self.c->it = self.iter + 1;
        }
    }
    double acc[1] = {0.0};                      // <r, r>, slot 2 of the block's partial row
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // two grid-stride steps per trip (eight loads in flight per thread); the thread's elements
    // enter its sum in the order of the plain loop
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += 2 * stride) {
        const int64_t i1 = i + stride;
        const bool two = i1 < n;
        const cx<T> x0 = x[i], r0 = r[i], p0 = p[i], q0 = q[i];
        cx<T> x1 = x0, r1 = r0, p1 = p0, q1 = q0;
        if (two) {
            x1 = x[i1];
            r1 = r[i1];
            p1 = p[i1];
            q1 = q[i1];
        }
        x[i] = cscale(x0, T(1)) + cscale(p0, alpha);
        const cx<T> rn0 = cscale(r0, T(1)) + cscale(q0, -alpha);
        r[i] = rn0;
        acc[0] += (double)cabs2(rn0);
        if (two) {
            x[i1] = cscale(x1, T(1)) + cscale(p1, alpha);
            const cx<T> rn1 = cscale(r1, T(1)) + cscale(q1, -alpha);
            r[i1] = rn1;
            acc[0] += (double)cabs2(rn1);
        }
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 4 + 2);
}
template <typename T>
int launch_cg_update_xr(hipStream_t st, const CgCtl *c, T alpha_host, cx<T> *x, cx<T> *r,
                        const cx<T> *p, const cx<T> *q, int64_t n, double *partials, const CgSelf &self) {
    const int grid = std::min(grid_for(n), kCgMaxBlocks);
    hipLaunchKernelGGL((cg_update_xr_kernel<T>), dim3(grid), dim3(kThreads), sizeof(double) * kThreads, st,
                       c, alpha_host, x, r, p, q, n, partials, self);
    SA_HIP(hipGetLastError());
    return grid;
}
#define SA_INST_CG(T)                                                                               \
    template int launch_cg_op<T>(hipStream_t, const CgCtl *, bool, const cx<T> *, const cx<T> *,    \
                                 cx<T> *, cx<T> *, T, int64_t, int, int, double *, const CgSelf &); \
    template void launch_cg_ctl<T>(hipStream_t, int, const double *, int, CgCtl *, CgPinned *,      \
                                   double *);                                                       \
    template void launch_cg_update_p<T>(hipStream_t, const CgCtl *, const cx<T> *, cx<T> *, int64_t); \
    template int launch_cg_update_xr<T>(hipStream_t, const CgCtl *, T, cx<T> *, cx<T> *,            \
                                        const cx<T> *, const cx<T> *, int64_t, double *,            \
                                        const CgSelf &);
SA_INST_CG(float)
SA_INST_CG(double)

// ---------------------------------------------------------------------------
// device-resident ADMM control (csc_kernels.h)
// ---------------------------------------------------------------------------
__global__ void admm_ctl_init_kernel(AdmmCtl *c, const AdmmCtlInit in) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    c->rho = in.rho;
    c->u_scale = in.u_scale;
    c->lmbda = in.lmbda;
    c->abstol = in.abstol;
    c->reltol = in.reltol;
    c->sqrt_nc = in.sqrt_nc;
    c->sqrt_nx = in.sqrt_nx;
    c->tau = in.tau;
    c->mu = in.mu;
    c->xi = in.xi;
    c->mu21 = in.mu21;
    c->k = in.k;
    c->stable_run = in.stable_run;
    c->emitted = in.emitted;
    c->is_f32 = in.is_f32;
    c->autorho = in.autorho;
    c->period = in.period;
    c->autoscaling = in.autoscaling;
    c->stdres = in.stdres;
    c->need_resid = in.need_resid;
    c->no_speculation = in.no_speculation;
    c->stop = 0;
    c->t0 = sa_wall_clock();
    admm_ctl_derive(c);
    c->thr_prev_f = in.thr_prev;
    c->thr21_prev_f = in.thr21_prev;
}

template <typename T>
__global__ void admm_ctl_update_kernel(AdmmCtl *c, const double *sums, AdmmRecord *rec, int index) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    admm_ctl_update_dev<T>(c, sums, rec, index);
}

void launch_admm_ctl_init(hipStream_t st, AdmmCtl *ctl, const AdmmCtlInit &in) {
    hipLaunchKernelGGL(admm_ctl_init_kernel, dim3(1), dim3(64), 0, st, ctl, in);
    SA_HIP(hipGetLastError());
}
void launch_admm_ctl_update(hipStream_t st, AdmmCtl *ctl, const double *sums, AdmmRecord *rec,
                            int index, bool f32) {
    if (f32)
        hipLaunchKernelGGL(admm_ctl_update_kernel<float>, dim3(1), dim3(64), 0, st, ctl, sums, rec, index);
    else
        hipLaunchKernelGGL(admm_ctl_update_kernel<double>, dim3(1), dim3(64), 0, st, ctl, sums, rec,
                           index);
    SA_HIP(hipGetLastError());
}

#define SA_INST(T)                                                                                 \
    template void launch_pad_dict<T>(hipStream_t, const T *, T *, int, int, int, int, int, int);        \
    template void launch_gram<T>(hipStream_t, const cx<T> *, T *, int64_t, int);                   \
    template int launch_grad_norm<T>(hipStream_t, const cx<T> *, const GradTerm<T> &, int64_t, int, \
                                     int, int, double *);                                          \
    template int launch_sm_solve<T>(hipStream_t, const cx<T> *, cx<T> *, const cx<T> *,            \
                                    const cx<T> *, const T *, T, int64_t, int, int, int, bool,     \
                                    bool, double *, const GradTerm<T> *, int);                     \
    template void launch_inner<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t,     \
                                  int, int);                                                       \
    template int launch_rfl2norm2<T>(hipStream_t, const cx<T> *, const cx<T> *, int64_t, int64_t,  \
                                     int, double *);                                               \
    template int launch_admm_post<T>(hipStream_t, const PostParams<T> &, double *);                \
    template void launch_relax<T>(hipStream_t, const T *, const T *, T *, T, int64_t);             \
    template void launch_vform_split<T>(hipStream_t, const T *, T *, T *, T, bool, int64_t);       \
    template void launch_vform_split_general<T>(hipStream_t, const T *, T *, T *, T, uint32_t, Dims5, \
                                                int, int, Weight<T>, Weight<T>, int);            \
    template void launch_vform_split_joint<T>(hipStream_t, const T *, T *, T *, T, T, bool, int,   \
                                              int64_t, int64_t);                                \
    template void launch_ystep<T>(hipStream_t, const T *, const T *, T *, T, T, T, uint32_t,       \
                                  Dims5, int, int, Weight<T>, Weight<T>, Weight<T>, int, int);     \
    template void launch_ustep<T>(hipStream_t, const T *, const T *, T *, T, int64_t);             \
    template int launch_admm_stats<T>(hipStream_t, const T *, const T *, const T *, const T *,     \
                                      uint32_t, Dims5, Weight<T>, Weight<T>, int, double *, int);  \
    template void launch_scale<T>(hipStream_t, T *, T, int64_t);                                   \
    template int launch_prox_l1<T>(hipStream_t, const T *, T *, T, uint32_t, Dims5, int, int,      \
                                   Weight<T>, double *);                                           \
    template void launch_prox_sl1l2<T>(hipStream_t, const T *, T *, T, T, int64_t, int, int64_t);  \
    template int launch_pgm_grad<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *,      \
                                    cx<T> *, int64_t, int, int, int, double *);                    \
    template void launch_axpy_c<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, T,          \
                                   int64_t);                                                       \
    template void launch_lincomb<T>(hipStream_t, cx<T> *, T, const cx<T> *, T, const cx<T> *, T,   \
                                    const cx<T> *, int64_t);                                       \
    template int launch_pair_stats<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *,    \
                                      int64_t, int64_t, int, double *);                            \
    template int launch_dhs_absmax<T>(hipStream_t, const cx<T> *, const cx<T> *, int64_t, int,     \
                                      int, double *);                                              \
    template int launch_ccmod_grad<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *,    \
                                      cx<T> *, int64_t, int, int, int, double *, int, int);        \
    template void launch_pcn_stats<T>(hipStream_t, const T *, T *, int, int, int, int, int, bool,  \
                                      int, FilterSizes);                                           \
    template int launch_pcn_apply<T>(hipStream_t, const T *, const T *, T *, int, int, int, int,   \
                                     int, double *, int, int, FilterSizes);                        \
    template int launch_asum<T>(hipStream_t, const T *, int64_t, double *);                        \
    template int launch_mask_apply<T>(hipStream_t, T *, const Weight<T> &, bool, int, int, int,    \
                                      int, double *);                                              \
    template void launch_md_pre<T>(hipStream_t, const T *, const T *, const T *, T *, T, int64_t);    \
    template int launch_md_y0step<T>(hipStream_t, const MdY0Args<T> &, double *);                  \
    template void launch_conj_outer<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *,         \
                                       int64_t, int, int);                                         \
    template void launch_zf_adjoint<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *,         \
                                       int64_t, int, int);                                         \
    template void launch_mc_zf_adjoint<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *,      \
                                          int64_t, int, int, int, int);                            \
    template void launch_cns_yu<T>(hipStream_t, const T *, const T *, T *, T, int64_t, int, int);  \
    template void launch_cns_mean<T>(hipStream_t, const T *, const T *, const T *, T *, T, T,      \
                                     int64_t, int, int);                                           \
    template int launch_cns_ustep<T>(hipStream_t, const T *, T *, const T *, const T *, T, T,      \
                                     int64_t, int, int, double *);                                 \
    template int launch_cns_ystats<T>(hipStream_t, const T *, const T *, int64_t, double *);       \
    template void launch_cns_xrrs_rhs<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, \
                                         T, cx<T> *, int64_t, int, int, int, int);                \
    template int launch_cns_xrrs_fin<T>(hipStream_t, const cx<T> *, const cx<T> *, T,              \
                                        const cx<T> *, int64_t, int, int, double *, int, int);     \
    template void launch_swap_inner<T>(hipStream_t, const cx<T> *, cx<T> *, int64_t, int, int);    \
    template void launch_tiled_resid<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *,  \
                                        cx<T> *, int64_t, int, int, int, int);                     \
    template int launch_md_dualres_tiled<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, \
                                            int64_t, int, int, int, int, int, double *);           \
    template void launch_zf_per_channel<T>(hipStream_t, const cx<T> *, cx<T> *, int64_t, int, int, \
                                           int, int);                                              \
    template void launch_ism_setup<T>(hipStream_t, const cx<T> *, cx<T> *, cx<T> *, cx<T> *,       \
                                      int64_t, int, int, T, const GradTerm<T> *, int);             \
    template int launch_ism_solve<T>(hipStream_t, const cx<T> *, cx<T> *, const cx<T> *,           \
                                     const cx<T> *, const cx<T> *, const cx<T> *, const cx<T> *,   \
                                     T, int64_t, int, int, int, int, bool, bool, double *,         \
                                     const GradTerm<T> *);                                         \
    template int launch_mc_pgm_grad<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *,   \
                                       cx<T> *, int64_t, int, int, int, int, double *);            \
    template void launch_mc_inner<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t,  \
                                     int, int, int, int);                                          \
    template void launch_mc_conj_outer<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *,      \
                                          int64_t, int, int, int, bool);                          \
    template int launch_mc_dhs_absmax<T>(hipStream_t, const cx<T> *, const cx<T> *, int64_t, int,  \
                                         int, int, double *);
SA_INST(float)
SA_INST(double)

}  // namespace sporco_amd
