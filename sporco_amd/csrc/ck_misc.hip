// ck_misc.hip -- generic (any shape, float32 / float64) kernels of libsporco_amd.so, declared in
// csc_kernels.h: fixed-order final reductions, pre / post-processing on device arrays, conjugate gradients with device-side scalars, the device-resident ADMM control.
//
// All of them are HBM-bound streaming kernels over (pixel, C, N, K) arrays with the filter index
// K fastest: consecutive lanes -> consecutive K, 16 bytes per lane where the shape allows, wave64
// shuffles for the per-pixel K-length inner products, double-precision block partials summed in
// a fixed order by finalize_kernel (run-to-run deterministic).
#include "csc_kernels_dev.h"

namespace sporco_amd {

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
// placement probe (csc_kernels.h launch_place_probe)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) place_probe_kernel(float *a, float *b, int64_t n16, int64_t win16,
                                                               int64_t step_a, int64_t step_b, int fresh_a,
                                                               int fresh_b, int64_t rot16) {
    // even workgroups walk a, odd ones b: two independent streams, as the write streams of a real
    // kernel are (element i of both arrays written by the same lane would tie the two streams to
    // each other address bit by address bit, and the result would depend on the distance between
    // the arrays modulo the interleave)
    const bool second = blockIdx.x & 1;
    float *base = second ? b : a;
    const int64_t step = second ? step_b : step_a;
    const int fresh = second ? fresh_b : fresh_a;
    const int64_t stride = (int64_t)(gridDim.x >> 1) * blockDim.x;
    for (int64_t i = (int64_t)(blockIdx.x >> 1) * blockDim.x + threadIdx.x; i < n16; i += stride) {
        // (b is walked from word rot16 on, wrapping: a sweep of rot16 pairs every part of a with
        // every part of b)
        int64_t j = second ? i + rot16 : i;
        if (j >= n16) j -= n16;
        const int64_t w = j / win16, o = j - w * win16;
        float *p = base + 4 * (w * step + o);
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (!fresh) sa_stream_load4(p, x);     // (moved, never computed on: any bit pattern survives)
        sa_stream_store4(p, x);
    }
}
void launch_place_probe(hipStream_t st, void *a, void *b, int64_t n16, int64_t win16, int64_t step_a,
                        int64_t step_b, bool fresh_a, bool fresh_b, int64_t rot16) {
    hipLaunchKernelGGL(place_probe_kernel, dim3(grid_for(n16) & ~1), dim3(kThreads), 0, st,
                       static_cast<float *>(a), static_cast<float *>(b), n16, win16, step_a, step_b,
                       fresh_a ? 1 : 0, fresh_b ? 1 : 0, rot16 % (n16 > 0 ? n16 : 1));
    SA_HIP(hipGetLastError());
}

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

}  // namespace sporco_amd
