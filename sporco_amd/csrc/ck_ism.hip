// ck_ism.hip -- generic (any shape, float32 / float64) kernels of libsporco_amd.so, declared in
// csc_kernels.h: multi-channel dictionaries: iterated Sherman-Morrison (any number of terms), LinSolveCheck of the consensus update.
//
// All of them are HBM-bound streaming kernels over (pixel, C, N, K) arrays with the filter index
// K fastest: consecutive lanes -> consecutive K, 16 bytes per lane where the shape allows, wave64
// shuffles for the per-pixel K-length inner products, double-precision block partials summed in
// a fixed order by finalize_kernel (run-to-run deterministic).
#include "csc_kernels_dev.h"

namespace sporco_amd {

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

// Complex-valued signals on a real handle (csc_kernels.h launch_pm_butterfly): the channel pair
// (A, B) = half spectra of the real and the imaginary part, array (npix, mid, 2, inner).
template <typename T>
__global__ void __launch_bounds__(kThreads) pm_butterfly_kernel(const cx<T> *src, cx<T> *dst, int64_t npix,
                                                                int mid, int inner, int Wf, int W,
                                                                int mode) {
    const int64_t total = npix * mid * inner;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t grp = t / inner;                 // (pix, j)
        const int i = (int)(t - grp * inner);
        const int64_t pix = grp / mid;
        const int64_t ia = grp * 2 * inner + i, ib = ia + inner;
        const cx<T> a = src[ia], b = src[ib];
        // s^2 = half the Parseval weight of the row frequency: 1/2 on the two self-conjugate
        // columns, which the + and the - array both hold in full, 1 elsewhere
        const T s = (mode != 0 && parseval_weight((int)(pix % Wf), Wf, W) == 1.0) ? (T)0.70710678118654752440
                                                                                   : T(1);
        if (mode != 2) {
            // (A + iB, A - iB): i times (re, im) = (-im, re)
            dst[ia] = cscale(mk<T>(a.re - b.im, a.im + b.re), s);
            dst[ib] = cscale(mk<T>(a.re + b.im, a.im - b.re), s);
        } else {
            // A = (P + M) / 2, B = (P - M) / (2i), with the scaling taken out
            const T h = T(0.5) / s;
            dst[ia] = cscale(mk<T>(a.re + b.re, a.im + b.im), h);
            dst[ib] = cscale(mk<T>(a.im - b.im, b.re - a.re), h);
        }
    }
}
template <typename T>
void launch_pm_butterfly(hipStream_t st, const cx<T> *src, cx<T> *dst, int64_t npix, int mid, int inner,
                         int W, int mode) {
    hipLaunchKernelGGL((pm_butterfly_kernel<T>), dim3(grid_for(npix * mid * inner)), dim3(kThreads), 0, st,
                       src, dst, npix, mid, inner, W / 2 + 1, W, mode);
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
        const int np = fs.dD * dH * dW;
        // support element i -> row of the (folded) array: slab i / (dH dW), height (i / dW) % dH
        auto row_of = [&](int i) -> int64_t {
            const int r = i / dW;
            return fs.Hs ? (int64_t)(r / dH) * fs.Hs + r % dH : r;
        };
        T n2 = T(0);
        for (int c = 0; c < Cd; ++c) {
            T mean = T(0);
            if (zm) {
                T s = T(0);
                for (int i = lane; i < np; i += kWave)
                    s += v[((row_of(i) * W + i % dW) * Cd + c) * K + k];
                mean = (T)(wave_sum((double)s) / (double)np);
            }
            for (int i = lane; i < np; i += kWave) {
                const T e = v[((row_of(i) * W + i % dW) * Cd + c) * K + k] - mean;
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
        const bool in_rows = fs.Hs ? (h / fs.Hs < fs.dD && h % fs.Hs < dH) : h < dH;
        const T o = (in_rows && x < dW && k < Kvalid) ? (vi - stats[2 * ck]) * stats[2 * k + 1] : T(0);
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

#define SA_INST(T) \
    template void launch_pcn_stats<T>(hipStream_t, const T *, T *, int, int, int, int, int, bool, int, FilterSizes); \
    template int launch_pcn_apply<T>(hipStream_t, const T *, const T *, T *, int, int, int, int, int, double *, int, int, FilterSizes); \
    template int launch_asum<T>(hipStream_t, const T *, int64_t, double *); \
    template void launch_cns_xrrs_rhs<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, T, cx<T> *, int64_t, int, int, int, int); \
    template int launch_cns_xrrs_fin<T>(hipStream_t, const cx<T> *, const cx<T> *, T, const cx<T> *, int64_t, int, int, double *, int, int); \
    template void launch_swap_inner<T>(hipStream_t, const cx<T> *, cx<T> *, int64_t, int, int); \
    template void launch_pm_butterfly<T>(hipStream_t, const cx<T> *, cx<T> *, int64_t, int, int, int, int); \
    template void launch_zf_per_channel<T>(hipStream_t, const cx<T> *, cx<T> *, int64_t, int, int, int, int); \
    template void launch_ism_setup<T>(hipStream_t, const cx<T> *, cx<T> *, cx<T> *, cx<T> *, int64_t, int, int, T, const GradTerm<T> *, int); \
    template int launch_ism_solve<T>(hipStream_t, const cx<T> *, cx<T> *, const cx<T> *, const cx<T> *, const cx<T> *, const cx<T> *, const cx<T> *, T, int64_t, int, int, int, int, bool, bool, double *, const GradTerm<T> *); \
    template int launch_mc_pgm_grad<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int, int, int, double *); \
    template void launch_mc_inner<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int, int, int); \
    template void launch_mc_conj_outer<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int, int, bool); \
    template int launch_mc_dhs_absmax<T>(hipStream_t, const cx<T> *, const cx<T> *, int64_t, int, int, int, double *);
SA_INST(float)
SA_INST(double)

}  // namespace sporco_amd
