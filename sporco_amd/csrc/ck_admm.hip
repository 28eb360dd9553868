// ck_admm.hip -- generic (any shape, float32 / float64) kernels of libsporco_amd.so, declared in
// csc_kernels.h: dictionary set-up, Sherman-Morrison solves, inner products and norms, the ADMM epilogue and its staged pieces, proximal operators.
//
// All of them are HBM-bound streaming kernels over (pixel, C, N, K) arrays with the filter index
// K fastest: consecutive lanes -> consecutive K, 16 bytes per lane where the shape allows, wave64
// shuffles for the per-pixel K-length inner products, double-precision block partials summed in
// a fixed order by finalize_kernel (run-to-run deterministic).
#include "csc_kernels_dev.h"

namespace sporco_amd {

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

// ---------------------------------------------------------------------------
// Complex-valued signals and dictionaries (sporco/admm/cbpdn.py:209-217: the reference switches
// to fftn / ifftn).  Here the real and imaginary parts of every complex array are two CHANNELS of
// the real machinery (channels c and c + Cc of 2 Cc), so all transforms stay real ones: with
// A = rfftn(Re z), B = rfftn(Im z) at a stored half-spectrum frequency f,
//     P = A + i B = fftn(z)(f),        M = A - i B = conj(fftn(z)(-f)),
// and the system of frequency -f, conjugated, is the system of f with the dictionary
// conj(Df(-f)) = DA - i DB and the signal SA - i SB.  One stored element therefore takes two
// Sherman-Morrison solves (linalg.solvedbi_sm, sporco/linalg.py:232-297), for P with
// (DA + i DB, SA + i SB) and for M with (DA - i DB, SA - i SB), and goes back as
// A = (XP + XM) / 2, B = (XP - XM) / (2 i).  The data-fidelity term of both frequencies is
// pw/2 rho^2 (|coefP|^2 + |coefM|^2) with the half-spectrum weight pw of the stored element.
// (The complex soft threshold is the l2 shrinkage over the channel pair: the caller runs the
// ConvBPDNJoint y step with lambda_l1 = 0, mu = lambda.)
// One thread per (pixel, complex channel, image) system.
template <typename T> __device__ __forceinline__ cx<T> mul_i(cx<T> z) { return mk<T>(-z.im, z.re); }

template <typename T>
__global__ void __launch_bounds__(kThreads) sm_cplx_kernel(cx<T> *__restrict__ xf,
                                                           const cx<T> *__restrict__ dfa,
                                                           const cx<T> *__restrict__ dfb,
                                                           const cx<T> *__restrict__ sf, int64_t npix,
                                                           int Cc, int N, int K, T rho, int W,
                                                           int want_obj, double *partials) {
    double acc[1] = {0.0};
    const int Wf = W / 2 + 1;
    const int64_t half = (int64_t)Cc * N, total = npix * half;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / half, r = t - pix * half;
        const int64_t ga = pix * 2 * half + r, gb = ga + half;
        const cx<T> *da = dfa + pix * K, *db = dfb + pix * K;
        cx<T> *xa = xf + ga * K, *xb = xf + gb * K;
        cx<T> qp = mk<T>(T(0), T(0)), qm = qp;
        T gp = T(0), gm = T(0);
        for (int k = 0; k < K; ++k) {
            const cx<T> ib = mul_i(db[k]), dp = da[k] + ib, dm = da[k] - ib;
            const cx<T> iy = mul_i(xb[k]), p = xa[k] + iy, m = xa[k] - iy;
            qp = qp + cmul(dp, p);
            qm = qm + cmul(dm, m);
            gp += cabs2(dp);
            gm += cabs2(dm);
        }
        const cx<T> is = mul_i(sf[gb]);
        const cx<T> cp = cscale(sf[ga] + is - qp, T(1) / (gp + rho));
        const cx<T> cm = cscale(sf[ga] - is - qm, T(1) / (gm + rho));
        for (int k = 0; k < K; ++k) {
            const cx<T> ib = mul_i(db[k]), dp = da[k] + ib, dm = da[k] - ib;
            const cx<T> iy = mul_i(xb[k]);
            const cx<T> xp = xa[k] + iy + cmulc(dp, cp), xm = xa[k] - iy + cmulc(dm, cm);
            const cx<T> df = xp - xm;
            xa[k] = cscale(xp + xm, T(0.5));
            xb[k] = mk<T>(T(0.5) * df.im, T(-0.5) * df.re);       // (XP - XM) / (2 i)
        }
        if (want_obj)
            acc[0] += 0.5 * parseval_weight((int)(pix % Wf), Wf, W) * (double)rho * (double)rho *
                      ((double)cabs2(cp) + (double)cabs2(cm));
    }
    block_sum_store<1>(acc, dyn_lds<double>(), partials + blockIdx.x);
}

template <typename T>
int launch_sm_cplx(hipStream_t st, cx<T> *xf, const cx<T> *dfa, const cx<T> *dfb, const cx<T> *sf,
                   int64_t npix, int Cc, int N, int K, T rho, int W, bool want_obj, double *partials) {
    const int grid = grid_for(npix * Cc * N);
    hipLaunchKernelGGL((sm_cplx_kernel<T>), dim3(grid), dim3(kThreads), sizeof(double) * (kThreads / kWave),
                       st, xf, dfa, dfb, sf, npix, Cc, N, K, rho, W, want_obj ? 1 : 0, partials);
    SA_HIP(hipGetLastError());
    return grid;
}

// out(npix, 2 Cc N) = the channel pair of sum_k Df vf for a complex dictionary and complex maps
// (linalg.inner over the filter axis at f and -f, see sm_cplx_kernel)
template <typename T>
__global__ void __launch_bounds__(kThreads) inner_cplx_kernel(const cx<T> *__restrict__ dfa,
                                                              const cx<T> *__restrict__ dfb,
                                                              const cx<T> *__restrict__ vf,
                                                              cx<T> *__restrict__ out, int64_t npix,
                                                              int Cc, int N, int K) {
    const int64_t half = (int64_t)Cc * N, total = npix * half;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / half, r = t - pix * half;
        const int64_t ga = pix * 2 * half + r, gb = ga + half;
        const cx<T> *da = dfa + pix * K, *db = dfb + pix * K;
        const cx<T> *xa = vf + ga * K, *xb = vf + gb * K;
        cx<T> qp = mk<T>(T(0), T(0)), qm = qp;
        for (int k = 0; k < K; ++k) {
            const cx<T> ib = mul_i(db[k]), iy = mul_i(xb[k]);
            qp = qp + cmul(da[k] + ib, xa[k] + iy);
            qm = qm + cmul(da[k] - ib, xa[k] - iy);
        }
        const cx<T> df = qp - qm;
        out[ga] = cscale(qp + qm, T(0.5));
        out[gb] = mk<T>(T(0.5) * df.im, T(-0.5) * df.re);
    }
}

template <typename T>
void launch_inner_cplx(hipStream_t st, const cx<T> *dfa, const cx<T> *dfb, const cx<T> *vf, cx<T> *out,
                       int64_t npix, int Cc, int N, int K) {
    hipLaunchKernelGGL((inner_cplx_kernel<T>), dim3(grid_for(npix * Cc * N)), dim3(kThreads), 0, st, dfa,
                       dfb, vf, out, npix, Cc, N, K);
    SA_HIP(hipGetLastError());
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
        Vec<T, VEC> yv, uv, vv;
        if (!GENERAL && p.v_in) {
            // single-array state: (Y, U) of the iterate follow from V as the epilogue that
            // stored it derived them
            vv = reinterpret_cast<const Vec<T, VEC> *>(p.v_in)[i];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                T y = soft(vv.v[e], p.thr_prev);
                if ((p.flags & F_NONNEG) && y < T(0)) y = T(0);
                yv.v[e] = y;
                uv.v[e] = vv.v[e] - y;
            }
        } else {
            yv = reinterpret_cast<const Vec<T, VEC> *>(p.y)[i];
            uv = reinterpret_cast<const Vec<T, VEC> *>(p.u)[i];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            admm_post_elem<T, GENERAL>(p, i * VEC + e, P, xv.v[e], yv.v[e], uv.v[e], acc, &vv.v[e]);
        if (!GENERAL && p.v_out) {
            reinterpret_cast<Vec<T, VEC> *>(p.v_out)[i] = vv;
        } else {
            reinterpret_cast<Vec<T, VEC> *>(p.y)[i] = yv;
            reinterpret_cast<Vec<T, VEC> *>(p.u)[i] = uv;
        }
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

#define SA_INST(T) \
    template void launch_pad_dict<T>(hipStream_t, const T *, T *, int, int, int, int, int, int); \
    template void launch_gram<T>(hipStream_t, const cx<T> *, T *, int64_t, int); \
    template int launch_grad_norm<T>(hipStream_t, const cx<T> *, const GradTerm<T> &, int64_t, int, int, int, double *); \
    template int launch_sm_solve<T>(hipStream_t, const cx<T> *, cx<T> *, const cx<T> *, const cx<T> *, const T *, T, int64_t, int, int, int, bool, bool, double *, const GradTerm<T> *, int); \
    template int launch_sm_cplx<T>(hipStream_t, cx<T> *, const cx<T> *, const cx<T> *, const cx<T> *, int64_t, int, int, int, T, int, bool, double *); \
    template void launch_inner_cplx<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int, int); \
    template void launch_inner<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int); \
    template int launch_rfl2norm2<T>(hipStream_t, const cx<T> *, const cx<T> *, int64_t, int64_t, int, double *); \
    template int launch_admm_post<T>(hipStream_t, const PostParams<T> &, double *); \
    template void launch_relax<T>(hipStream_t, const T *, const T *, T *, T, int64_t); \
    template void launch_vform_split<T>(hipStream_t, const T *, T *, T *, T, bool, int64_t); \
    template void launch_vform_split_general<T>(hipStream_t, const T *, T *, T *, T, uint32_t, Dims5, int, int, Weight<T>, Weight<T>, int); \
    template void launch_vform_split_joint<T>(hipStream_t, const T *, T *, T *, T, T, bool, int, int64_t, int64_t); \
    template void launch_ystep<T>(hipStream_t, const T *, const T *, T *, T, T, T, uint32_t, Dims5, int, int, Weight<T>, Weight<T>, Weight<T>, int, int); \
    template void launch_ustep<T>(hipStream_t, const T *, const T *, T *, T, int64_t); \
    template int launch_admm_stats<T>(hipStream_t, const T *, const T *, const T *, const T *, uint32_t, Dims5, Weight<T>, Weight<T>, int, double *, int); \
    template void launch_scale<T>(hipStream_t, T *, T, int64_t); \
    template int launch_prox_l1<T>(hipStream_t, const T *, T *, T, uint32_t, Dims5, int, int, Weight<T>, double *); \
    template void launch_prox_sl1l2<T>(hipStream_t, const T *, T *, T, T, int64_t, int, int64_t);
SA_INST(float)
SA_INST(double)

}  // namespace sporco_amd
