// ck_dict.hip -- generic (any shape, float32 / float64) kernels of libsporco_amd.so, declared in
// csc_kernels.h: PGM sparse-coding kernels, dictionary-update kernels, masked data fidelity, mask decoupling, the ADMM consensus update.
//
// All of them are HBM-bound streaming kernels over (pixel, C, N, K) arrays with the filter index
// K fastest: consecutive lanes -> consecutive K, 16 bytes per lane where the shape allows, wave64
// shuffles for the per-pixel K-length inner products, double-precision block partials summed in
// a fixed order by finalize_kernel (run-to-run deterministic).
#include "csc_kernels_dev.h"

namespace sporco_amd {

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
                                                              int W, double *partials, int64_t wf_div) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int64_t total = npix * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        // (wf_div: the operands are tile-major (Wf, CN, H, K) -- the row frequency is the slowest
        // index, wf_div = CN H K elements apart)
        const int wf = wf_div ? (int)(i / wf_div) : (int)((i / cols) % Wf);
        cx<T> dlt = a[i];
        if (b) dlt = dlt - b[i];
        const double d2 = (double)cabs2(dlt);
        acc[0] += parseval_weight(wf, Wf, W) * d2;
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
                      int64_t cols, int W, double *partials, int64_t wf_div) {
    const int grid = grid_for(npix * cols);
    hipLaunchKernelGGL((pair_stats_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * 4 * (kThreads / kWave), st, a, b, g, npix, cols, W / 2 + 1,
                       W, partials, wf_div);
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

// Sums over signal-sized residual spectra e = sum_k Df v - Sf in the tile-major layout
// (Wf CN, H) -- what the step-size policies need of the gradient g = conj(Df) e without forming it
// (sporco/pgm/stepsize.py:67-145): with d1 = a - b, d2 = c - d (b, d may be null; c null: d2 = d1)
// and G = sum_k |Df|^2 of the frequency,
//   [0] sum G |d1|^2  = <g1, g1>      [1] sum G^2 |d1|^2 = <g1, hessian_f(g1)>
//   [2] sum Re(conj(d2) d1) = <x2, g1> for a difference of iterates x2 whose residual difference is d2
// over the half-spectrum array, unweighted (the policies sum the rfftn arrays as they are).
template <typename T>
__global__ void __launch_bounds__(kThreads) resid_stats_kernel(const cx<T> *__restrict__ a,
                                                               const cx<T> *__restrict__ b,
                                                               const cx<T> *__restrict__ c,
                                                               const cx<T> *__restrict__ d,
                                                               const T *__restrict__ gramt, int64_t nrows,
                                                               int H, int CN, double *partials) {
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < nrows;
         row += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tile = row / H;
        const int h = (int)(row - tile * H);
        const double g = (double)gramt[(tile / CN) * H + h];
        cx<T> d1 = a[row];
        if (b) d1 = d1 - b[row];
        cx<T> d2 = d1;
        if (c) {
            d2 = c[row];
            if (d) d2 = d2 - d[row];
        }
        const double m = (double)cabs2(d1);
        acc[0] += g * m;
        acc[1] += g * g * m;
        acc[2] += (double)d2.re * (double)d1.re + (double)d2.im * (double)d1.im;
    }
    block_sum_store<3>(acc, dyn_lds<double>(), partials + (int64_t)blockIdx.x * 3);
}
template <typename T>
int launch_resid_stats(hipStream_t st, const cx<T> *a, const cx<T> *b, const cx<T> *c, const cx<T> *d,
                       const T *gramt, int64_t ntiles, int H, int CN, double *partials) {
    const int grid = grid_for(ntiles * H);
    hipLaunchKernelGGL((resid_stats_kernel<T>), dim3(grid), dim3(kThreads),
                       sizeof(double) * 3 * (kThreads / kWave), st, a, b, c, d, gramt, ntiles * H, H, CN,
                       partials);
    SA_HIP(hipGetLastError());
    return grid;
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

#define SA_INST(T) \
    template int launch_pgm_grad<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int, int, double *); \
    template void launch_axpy_c<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, T, int64_t); \
    template void launch_lincomb<T>(hipStream_t, cx<T> *, T, const cx<T> *, T, const cx<T> *, T, const cx<T> *, int64_t); \
    template int launch_pair_stats<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, int64_t, int64_t, int, double *, int64_t); \
    template int launch_dhs_absmax<T>(hipStream_t, const cx<T> *, const cx<T> *, int64_t, int, int, double *); \
    template int launch_ccmod_grad<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int, int, double *, int, int); \
    template int launch_mask_apply<T>(hipStream_t, T *, const Weight<T> &, bool, int, int, int, int, double *); \
    template void launch_md_pre<T>(hipStream_t, const T *, const T *, const T *, T *, T, int64_t); \
    template int launch_md_y0step<T>(hipStream_t, const MdY0Args<T> &, double *); \
    template void launch_conj_outer<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int); \
    template void launch_zf_adjoint<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int); \
    template void launch_mc_zf_adjoint<T>(hipStream_t, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int, int, int); \
    template void launch_cns_yu<T>(hipStream_t, const T *, const T *, T *, T, int64_t, int, int); \
    template void launch_cns_mean<T>(hipStream_t, const T *, const T *, const T *, T *, T, T, int64_t, int, int); \
    template int launch_cns_ustep<T>(hipStream_t, const T *, T *, const T *, const T *, T, T, int64_t, int, int, double *); \
    template int launch_cns_ystats<T>(hipStream_t, const T *, const T *, int64_t, double *); \
    template void launch_tiled_resid<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, cx<T> *, int64_t, int, int, int, int); \
    template int launch_md_dualres_tiled<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, int64_t, int, int, int, int, int, double *); \
    template int launch_resid_stats<T>(hipStream_t, const cx<T> *, const cx<T> *, const cx<T> *, const cx<T> *, const T *, int64_t, int, int, double *);
SA_INST(float)
SA_INST(double)

}  // namespace sporco_amd
