// csc_fused.hip -- register-resident column FFT + Sherman-Morrison + column IFFT.
//
// One workgroup of 16 waves owns one (wf, cn) tile: all H = 16*N1 points of all
// K <= 64 filters (N1 = 16 or 32, i.e. H = 256 or 512; 256 KiB of complex64 at
// 512 x 64, which is why the tile lives in the 512 KiB vector register file and
// not in the 160 KiB LDS).  Lane = filter k, so every global access of a wave
// is one contiguous K*8-byte row and the K-length inner product of
// linalg.solvedbi_sm (sporco/linalg.py:232-297) is a cross-lane reduction.
//
// The length-H transform is split H = N1 x 16 (Cooley-Tukey):
//   forward   wave w holds rows h = 16*h1 + w:   DIF FFT-N1 over h1 in registers,
//             twiddle W_H^(w*f1), exchange through LDS so that wave w' holds
//             f1 in {w', w'+16} x all 16 h2, DIF FFT-16 over h2  ->  X[f1 + N1*f2]
//   solve     per frequency f: q = sum_k Df*yuf (transposing wave reduction),
//             xf = yuf + conj(Df) * (Sf - q) / (sum_k |Df|^2 + rho)
//   inverse   the mirror image (DIT FFT-16, conj twiddle, LDS exchange, DIT FFT-N1),
//             landing on the rows the wave loaded, stored in place.
// LDS is used only for the two exchanges (128 KiB, real and imaginary halves in
// turn when N1 = 32).  Forward FFTs are decimation-in-frequency (natural in,
// bit-reversed out), inverse ones decimation-in-time (bit-reversed in, natural
// out), so no reordering pass exists anywhere.
#include "csc_fused.h"

#include "csc_fused_body.h"
#include "regfft.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace sporco_amd {

namespace {

using namespace regfft;

template <int N1, int NW, int LPARAM, int KC, bool GRAD, bool KRT = false, bool PER_TILE = false,
          int DBG = 0>
// (compile-time K: four waves per SIMD, i.e. at most 128 registers -- at 8 waves per workgroup that is
// the difference between two workgroups on a CU and one, and the GradReg form at H = 256 sat at 130)
__global__ void __launch_bounds__(NW * 64) SA_MIN_WAVES_PER_SIMD(KC == 64 ? 4 : 1)
fused_cols_kernel(const FusedColsArgs<float> a) {
    constexpr bool PERSIST = (NW == 16 || N1 == 64) && KC == 64;   // (run-time K: scalar registers are short)
    constexpr int AOFF = 0;
    SA_ARGS_PTR_T(FusedColsArgs<float>) afix = nullptr;
    (void)afix;
#include "csc_fused_body.inc"
}

// Dual residual of the mask-decoupled X-step (cbpdn.py:1814-1818): the forward half of the column
// pass on the row spectra of u1 -- load the tile, FFT-N1, twiddle, exchange, FFT-NW -- and then, per
// frequency f, sum_k |conj(Df[f][k]) u0f[f] + u1f[f][k]|^2 instead of a solve; nothing is written
// back (one read pass over the spectrum).  partials[tile] carries the Parseval weight of wf.
// N1: rows per thread -- 32, or a mixed-radix length (16 waves, LP = 1; the second exchange group partly
// filled, as in csc_fused_body.inc)
template <int NW, int LP, int KC, int N1 = 32>
__global__ void __launch_bounds__(NW * 64) cols_dualres_kernel(const FusedColsArgs<float> a) {
    constexpr bool MR = mr_length(N1);
    static_assert(!MR || (NW == 16 && LP == 1), "mixed-radix heights: 16 waves, one line per group");
    constexpr int H = N1 * NW, J = MR ? (N1 > NW ? 2 : 1) : N1 / NW;
    constexpr int LBW = ilog2(NW);
    constexpr int FP = LP * NW, Q = J / LP;
    static_assert(J % LP == 0, "lines per group must divide the lines per thread");
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int K = KC ? KC : (a.Ks ? a.Ks : a.K);
    const bool kv = KC == 64 ? true : k < a.K;
    f2 *LA = dyn_lds<f2>();
    double *scratch = reinterpret_cast<double *>(LA + FP * NW * 64);
    const cf zero = mk<float>(0.f, 0.f);
    const int ko = (w * K + k) * (int)sizeof(cf);
    const int Wf = a.W / 2 + 1;
    const int64_t ntiles = (int64_t)Wf * a.CN;
    int token = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int wf = (int)(tile / a.CN);
        const BufRsrc Tb = make_rsrc(a.t + tile * H * K, (uint32_t)(H * K * sizeof(cf)));
        const BufRsrc Db = make_rsrc(a.dft + (int64_t)wf * H * K, (uint32_t)(H * K * sizeof(cf)));
        const cf *S = a.sft + tile * H + w;
        const cf *twA = a.twA + w * N1;
        cf v[N1];
#pragma unroll
        for (int h1 = 0; h1 < N1; ++h1)
            v[h1] = kv ? buf_load_cf(Tb, ko, NW * h1 * K * (int)sizeof(cf)) : zero;
        dif1<N1, false>(v, 0);
        reg_fence<N1>(v, 0, token);
#pragma unroll
        for (int i = 1; i < N1; ++i) v[i] = cmul(v[i], twA[i]);
        reg_fence<N1>(v, 0, token);
        float acc = 0.f;
        static_for<Q>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const bool lv = !MR || q * FP + w < N1;      // (this wave's line of the group exists)
#pragma unroll
            for (int fl = 0; fl < FP; ++fl) {
                if (q * FP + fl >= N1) continue;
                const cf x = v[pos1<N1>(q * FP + fl)];
                f2 t;
                t.x = x.re;
                t.y = x.im;
                LA[(fl * NW + w) * 64 + k] = t;
            }
            __syncthreads();
            if (lv) {
            cf u[FP];
#pragma unroll
            for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
                for (int h2 = 0; h2 < NW; ++h2) {
                    const f2 t = LA[((w + NW * jl) * NW + h2) * 64 + k];
                    u[NW * jl + h2] = mk<float>(t.x, t.y);
                }
                dif<NW, false>(u, NW * jl);     // u[NW jl + i] = X[f1 + N1 brev(i)], f1 = w + NW j
            }
#pragma unroll
            for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const int fo = NW * (q * LP + jl) + N1 * brev(i, LBW);   // f - w
                    const cf d = kv ? buf_load_cf_cached(Db, ko, fo * K * (int)sizeof(cf)) : zero;
                    cf s0;
                    sa_uload2(reinterpret_cast<const float *>(S + fo), s0.re, s0.im);
                    const cf val = cmulc(d, s0) + u[NW * jl + i];
                    acc += kv ? cabs2(val) : 0.f;
                }
            }
            }   // lv
            __syncthreads();     // the exchange buffer is reused by the next group / tile
        });
        const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
        double ac[1] = {(double)acc * pw};
        block_sum_store<1>(ac, scratch, a.partials + tile);
        __syncthreads();
    }
}

// g1t[wf][h] = 1 + sum_k |Df|^2 / (mu wg_k (ghh[h] + ghw[wf]) + rho): the Sherman-Morrison
// denominator of linalg.solvedbd_sm_c (linalg.py:346-366), refreshed when rho changes.
// One wave per (wf, h) row of the tile-major Df, lane = filter.
__global__ void __launch_bounds__(256) grad_g1_kernel(const FusedColsArgs<float> a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nrows = (int64_t)(a.W / 2 + 1) * a.H;
    if (row >= nrows) return;
    const int wf = (int)(row / a.H), h = (int)(row % a.H);
    const float gh = a.ghh[h], gw = a.ghw[wf];
    float s = 0.f;
    const int Ks = a.Ks ? a.Ks : a.K;
    for (int k = lane; k < a.K; k += 64) {
        const float ak = a.mu * (a.wg ? a.wg[k] : 1.f);
        const float dd = ak * gh + (ak * gw + a.rho);
        s += cabs2(a.dft[row * Ks + k]) / dd;
    }
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) a.g1t_out[row] = 1.f + s;
}

// ---------------------------------------------------------------------------
// K = 64 * NH: the column pass in two kernels over 64-filter slabs (see csc_fused.h)
// ---------------------------------------------------------------------------
// KS: compile-time row stride in filters (128), or 0 for a run-time a.c.K.
template <int NW, int LP, int KS, bool GRAD>
__global__ void __launch_bounds__(NW * 64) cols_fwd_partial_kernel(const FusedSlabArgs<float> aa) {
    const FusedColsArgs<float> &a = aa.c;
    constexpr int N1 = 32, H = N1 * NW, J = N1 / NW;
    constexpr int LBW = ilog2(NW);
    constexpr int FP = LP * NW, Q = J / LP, CPL = NW / 4, NCH = LP * CPL;
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int K = KS ? KS : a.K;
    const int NH = (K + 63) / 64, slab = blockIdx.y;
    const bool kv = KS == 128 ? true : slab * 64 + k < K;   // the last slab may be partial
    const cf zero = mk<float>(0.f, 0.f);
    const int Wf = a.W / 2 + 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wf = (slot / a.CN) * 8 + xcd;
    if (wf >= Wf) return;
    if (a.ctl && a.ctl->stop) return;
    const int tile = wf * a.CN + slot % a.CN;
    const uint32_t tbytes = (uint32_t)(H * K * sizeof(cf));
    const BufRsrc Tb = make_rsrc(a.t + (int64_t)tile * H * K, tbytes);
    const BufRsrc Db = make_rsrc(a.dft + (int64_t)wf * H * K, tbytes);
    const int ko = (w * K + slab * 64 + k) * (int)sizeof(cf);
    const cf *twA = a.twA + w * N1;
    cf *qp = aa.qpart + ((int64_t)tile * NH + slab) * H + w;
    f2 *L = dyn_lds<f2>();
    int token = 0;
    // GRAD (ConvBPDNGradReg): the partial sums are of Df yuf / dd, dd = ak ghh[f] + bk
    float ak = 0.f, bk = 0.f;
    const float *GH = a.ghh + w;
    if constexpr (GRAD) {
        ak = a.mu * ((a.wg && kv) ? a.wg[slab * 64 + k] : 1.f);
        bk = ak * sa_uload(a.ghw + wf) + a.rho;
    }

    cf v[N1];
#pragma unroll
    for (int h1 = 0; h1 < N1; ++h1) v[h1] = kv ? buf_load_cf(Tb, ko, NW * h1 * K * (int)sizeof(cf)) : zero;
    dif<N1, false>(v, 0);
    reg_fence<N1>(v, 0, token);
#pragma unroll
    for (int i = 1; i < N1; ++i) {
        cf tw;
        sa_uload2(reinterpret_cast<const float *>(twA + i), tw.re, tw.im);
        v[i] = cmul(v[i], tw);
    }
    reg_fence<N1>(v, 0, token);

    static_for<Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
#pragma unroll
        for (int fl = 0; fl < FP; ++fl) {
            const cf x = v[brev(q * FP + fl, 5)];
            f2 t;
            t.x = x.re;
            t.y = x.im;
            L[(fl * NW + w) * 64 + k] = t;
        }
        cf dn[4];
        auto prefetch = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                dn[e] = kv ? buf_load_cf_cached(Db, ko, fo * K * (int)sizeof(cf)) : zero;
            }
        };
        prefetch(std::integral_constant<int, 0>{});
        __syncthreads();
        cf u[FP];
#pragma unroll
        for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
            for (int h2 = 0; h2 < NW; ++h2) {
                const f2 t = L[((w + NW * jl) * NW + h2) * 64 + k];
                u[NW * jl + h2] = mk<float>(t.x, t.y);
            }
        }
        if (q + 1 < Q) __syncthreads();
        static_for<NCH>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
            if constexpr (c == 0) dif<NW, false>(u, NW * jl);
            cf d[4];
            float red[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = dn[e];
            if constexpr (g + 1 < NCH) prefetch(std::integral_constant<int, g + 1>{});
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cf p = cmul(d[e], u[NW * jl + 4 * c + e]);
                if constexpr (GRAD) {
                    const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                    p = cscale(p, sa_rcp(ak * sa_uload(GH + fo) + bk));
                }
                red[2 * e] = p.re;
                red[2 * e + 1] = p.im;
            }
            const float tot = reduce8_across_lanes(red, k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                if (kv) buf_store_cf(Tb, ko, fo * K * (int)sizeof(cf), u[NW * jl + 4 * c + e]);
                // lane 16 e holds Re, lane 16 e + 8 holds Im of the slab's partial sum
                if (k == 16 * e) qp[fo].re = tot;
                if (k == 16 * e + 8) qp[fo].im = tot;
            }
        });
    });
}

template <int NW, int LP, int KS, bool GRAD>
// (K = 128: at most 128 registers, as fused_cols_kernel)
__global__ void __launch_bounds__(NW * 64) SA_MIN_WAVES_PER_SIMD(KS == 128 ? 4 : 1)
cols_sm_apply_inv_kernel(const FusedSlabArgs<float> aa) {
    const FusedColsArgs<float> &a = aa.c;
    constexpr int N1 = 32, H = N1 * NW, J = N1 / NW;
    constexpr int LBW = ilog2(NW);
    constexpr int FP = LP * NW, Q = J / LP, CPL = NW / 4, NCH = LP * CPL;
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int K = KS ? KS : a.K;
    const int NH = (K + 63) / 64, slab = blockIdx.y;
    const bool kv = KS == 128 ? true : slab * 64 + k < K;   // the last slab may be partial
    const cf zero = mk<float>(0.f, 0.f);
    const int Wf = a.W / 2 + 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wf = (slot / a.CN) * 8 + xcd;
    if (wf >= Wf) return;
    const int tile = wf * a.CN + slot % a.CN;
    const uint32_t tbytes = (uint32_t)(H * K * sizeof(cf));
    const BufRsrc Tb = make_rsrc(a.t + (int64_t)tile * H * K, tbytes);
    const BufRsrc Db = make_rsrc(a.dft + (int64_t)wf * H * K, tbytes);
    const int ko = (w * K + slab * 64 + k) * (int)sizeof(cf);
    const cf *S = a.sft + (int64_t)tile * H + w;
    const float *G = (GRAD ? a.g1t : a.gramt) + (int64_t)wf * H + w;
    const float *GH = a.ghh + w;
    const cf *twB = a.twB + w * N1;
    const cf *qp = aa.qpart + (int64_t)tile * NH * H + w;
    f2 *L = dyn_lds<f2>();
    double *scratch = reinterpret_cast<double *>(L + FP * NW * 64);
    // (device-driven solve: rho from the control block; nothing to do once it has stopped)
    if (a.ctl && a.ctl->stop) return;
    const float rho = a.ctl ? a.ctl->rho_f : a.rho;
    int token = 0;
    float obj = 0.f, rg = 0.f, ak = 0.f, bk = 0.f, gw = 0.f;
    if constexpr (GRAD) {
        gw = sa_uload(a.ghw + wf);
        ak = a.mu * ((a.wg && kv) ? a.wg[slab * 64 + k] : 1.f);
        bk = ak * gw + rho;
    }

    cf v[N1];
    static_for<Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        cf u[FP];
#pragma unroll
        for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int fo = NW * (q * LP + jl) + N1 * brev(i, LBW);
                u[NW * jl + i] = kv ? buf_load_cf(Tb, ko, fo * K * (int)sizeof(cf)) : zero;
            }
        }
        static_for<NCH>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                const cf d = kv ? buf_load_cf_cached(Db, ko, fo * K * (int)sizeof(cf)) : zero;
                cf qq = mk<float>(0.f, 0.f), sv;
                for (int sl = 0; sl < NH; ++sl) {
                    cf t;
                    sa_uload2(reinterpret_cast<const float *>(qp + (int64_t)sl * H + fo), t.re, t.im);
                    qq = qq + t;
                }
                sa_uload2(reinterpret_cast<const float *>(S + fo), sv.re, sv.im);
                if constexpr (GRAD) {
                    const float gh = sa_uload(GH + fo);
                    const cf coef = cscale(sv - cscale(qq, rho), sa_rcp(sa_uload(G + fo)));
                    obj += cabs2(coef);
                    const cf xn = cscale(cscale(u[NW * jl + 4 * c + e], rho) + cmulc(d, coef),
                                         sa_rcp(ak * gh + bk));
                    rg += (gh + gw) * cabs2(xn);
                    u[NW * jl + 4 * c + e] = xn;
                } else {
                    const float inv = sa_rcp(sa_uload(G + fo) + rho);
                    const cf coef = cscale(sv - qq, inv);
                    obj = cabs2_add(obj, coef);
                    u[NW * jl + 4 * c + e] = cmulc_add(u[NW * jl + 4 * c + e], d, coef);
                }
            }
            if constexpr (c == CPL - 1) {
                dit<NW, true>(u, NW * jl);
#pragma unroll
                for (int h2 = 1; h2 < NW; ++h2) {
                    cf tw;
                    sa_uload2(reinterpret_cast<const float *>(twB + NW * j + h2), tw.re, tw.im);
                    u[NW * jl + h2] = cmulc(tw, u[NW * jl + h2]);
                }
            }
        });
        {
            float &ob_ = obj, &rg_ = rg;
            int &tk_ = token;
            SA_VGPR_FENCE3(ob_, rg_, tk_);
        }
#pragma unroll
        for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
            for (int h2 = 0; h2 < NW; ++h2) {
                f2 t;
                t.x = u[NW * jl + h2].re;
                t.y = u[NW * jl + h2].im;
                L[((w + NW * jl) * NW + h2) * 64 + k] = t;
            }
        }
        __syncthreads();
#pragma unroll
        for (int fl = 0; fl < FP; ++fl) {
            const f2 t = L[(fl * NW + w) * 64 + k];
            v[brev(q * FP + fl, 5)] = mk<float>(t.x, t.y);
        }
        if (q + 1 < Q) __syncthreads();
    });
    reg_fence<N1>(v, 0, token);
    dit<N1, true>(v, 0);
#pragma unroll
    for (int h1 = 0; h1 < N1; ++h1)
        if (kv) buf_store_cf(Tb, ko, NW * h1 * K * (int)sizeof(cf), v[h1]);

    // every slab computes the same |coef|^2: slab 0 reports it
    const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
    if constexpr (GRAD) {
        // per (tile, slab): the data-fidelity sum (slab 0) and this slab's share of the
        // gradient term
        const float wk = (a.wg && kv) ? a.wg[slab * 64 + k] : 1.f;
        double acc[2] = {(k == 0 && slab == 0) ? (double)obj * pw : 0.0,
                         kv ? (double)(rg * wk) * pw : 0.0};
        block_sum_store<2>(acc, scratch, a.partials + 2 * ((int64_t)tile * NH + slab));
    } else {
        double acc[1] = {(k == 0 && slab == 0) ? (double)obj * pw * (double)rho * (double)rho : 0.0};
        if (slab == 0) block_sum_store<1>(acc, scratch, a.partials + tile);
    }
}

// ---------------------------------------------------------------------------
// The same column pass as ONE launch: the NH slab workgroups of a tile run side by side on
// different CUs, keep their 64-filter slab of the spectrum in registers, and exchange only
// the partial inner products through `qpart` (written through, flagged per (tile, slab) with
// the launch's sequence number).  Two X-sized passes instead of four.  The grid is persistent
// and never larger than the device holds at once (one 16-wave / two 8-wave workgroups per
// CU): partners are consecutive workgroup indices and walk the same tiles in the same order,
// so whoever waits, waits for a workgroup that is resident.  A poll that does not complete
// (2^22 rounds) raises `*coop_err` instead of hanging the device.
// ---------------------------------------------------------------------------
// PGM: the gradient step of the fused FISTA iteration for K > 64 instead (csc_pgm.h pgm_grad_ifft):
// the input rows are the spectrum Yf itself (no forward transform), the per-row coefficient is
// -(sum_k Df Yf - Sf) / L, the output goes to a.c.t, and partials[tile] = sum |sum_k Df Yf - Sf|^2.
// N1: rows per thread -- 32, or a mixed-radix length (16 waves, LP = 1: the second exchange group partly
// filled, as in csc_fused_body.inc; not with PGM)
template <int NW, int LP, int KS, bool GRAD, bool PGM = false, int N1 = 32>
__global__ void __launch_bounds__(NW * 64) cols_slab_coop_kernel(const FusedSlabArgs<float> aa) {
    static_assert(!(GRAD && PGM), "one or the other");
    constexpr bool MR = mr_length(N1);
    static_assert(!MR || (NW == 16 && LP == 1 && !PGM), "mixed-radix heights: 16 waves, one line per group, ADMM");
    constexpr int H = N1 * NW, J = MR ? (N1 > NW ? 2 : 1) : N1 / NW;
    constexpr int LBW = ilog2(NW);
    constexpr int FP = LP * NW, Q = J / LP, CPL = NW / 4, NCH = LP * CPL;
    static_assert(MR || Q * FP == N1, "a thread holds N1 spectrum rows");
    constexpr int NU = Q * FP;          // slots of the spectrum rows of a thread (>= N1)
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int K = KS ? KS : aa.c.K;
    const int NH = (K + 63) / 64;
    const int slab = blockIdx.x % NH, pair = blockIdx.x / NH, npairs = gridDim.x / NH;
    const bool kv = KS == 128 ? true : slab * 64 + k < K;   // the last slab may be partial
    const cf zero = mk<float>(0.f, 0.f);
    const int xcd = pair & 7;          // (virtual: the residue of the row frequencies it walks)
    const int ko = (w * K + slab * 64 + k) * (int)sizeof(cf);
    f2 *L = dyn_lds<f2>();
    double *scratch = reinterpret_cast<double *>(L + FP * NW * 64);
    int token = 0;
    if (aa.c.ctl && aa.c.ctl->stop) return;     // (every workgroup of the launch sees the same value)
    {   // groups start a fraction of a tile's time apart (the partners of a group together)
        const int ph = (pair >> 3) % aa.c.stagger_groups;
        for (int i = 0; i < ph * aa.c.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // the partial sums this wave needs in phase 2, one row per lane: lane l < 32 holds row
    // r = l of slab 0 (+ 2, ...), lane l >= 32 the same row of slab 1 (+ 3, ...)
    const int rl = k & 31;
    const int fo_lane = NW * (rl / NW) + N1 * brev(rl % NW, LBW);
    // (mixed-radix heights: the line w + NW (rl / NW) may not exist -- its lanes hold zeros)
    const bool row_ok = !MR || (rl < NU && w + NW * (rl / NW) < N1);
    bool gave_up = false;

    for (int slot = pair >> 3;; slot += npairs >> 3) {
    SA_ARGS_PTR_T(FusedSlabArgs<float>) ap = sa_args_reload<true>(aa);
    const int Wf = ap->c.W / 2 + 1, CN = ap->c.CN;
    if (slot >= ((Wf + 7) / 8) * CN) break;
    const int wf = (slot / CN) * 8 + xcd;
    if (wf >= Wf) break;
    const int tile = wf * CN + slot % CN;
    const AdmmCtl *ctl = ap->c.ctl;
    const float rho = ctl ? ctl->rho_f : ap->c.rho;
    const uint32_t tbytes = (uint32_t)(H * K * sizeof(cf));
    const BufRsrc Tb = make_rsrc(ap->c.t + (int64_t)tile * H * K, tbytes);
    const BufRsrc Db = make_rsrc(ap->c.dft + (int64_t)wf * H * K, tbytes);
    const cf *twA = ap->c.twA + w * N1;
    const cf *twB = ap->c.twB + w * (J * NW);
    const cf *S = ap->c.sft + (int64_t)tile * H + w;
    const float *G = PGM ? nullptr : (GRAD ? ap->c.g1t : ap->c.gramt) + (int64_t)wf * H + w;
    const float *GH = ap->c.ghh + w;
    cf *qp = ap->qpart + (int64_t)tile * NH * H + w;      // [slab][f]
    // where this lane publishes: lane 16 e (+ 8) -> Re (Im) of row N1 brev(e, 2) 2^(LBW - 2) + ...
    float *pub = reinterpret_cast<float *>(qp + (int64_t)slab * H) +
                 2 * (N1 * (brev(k >> 4, 2) << (LBW - 2))) + ((k >> 3) & 1);
    unsigned *flags = ap->coop_flags + (int64_t)tile * NH;
    const unsigned seq = ap->coop_seq;
    float rg = 0.f, ak = 0.f, bk = 0.f, gw = 0.f;
    if constexpr (GRAD) {
        gw = sa_uload(ap->c.ghw + wf);
        ak = ap->c.mu * ((ap->c.wg && kv) ? ap->c.wg[slab * 64 + k] : 1.f);
        bk = ak * gw + rho;
    }

    // ---- phase 1: FFT along H, this slab's share of sum_k Df yuf ------------------------
    cf uall[NU];                       // the slab's spectrum rows: group q in [q FP, (q + 1) FP)
    if constexpr (MR) {
#pragma unroll
        for (int i = 0; i < NU; ++i) uall[i] = zero;
    }
    if constexpr (PGM) {
        // the iterate is already a spectrum: rows f = w + NW j + N1 brev(i) of Yf, and the slab's
        // share of sum_k Df Yf
        const BufRsrc Yb = make_rsrc(ap->pgm_yf + (int64_t)tile * H * K, tbytes);
        static_for<Q>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
#pragma unroll
            for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
                for (int i = 0; i < NW; ++i) {
                    const int fo = NW * (q * LP + jl) + N1 * brev(i, LBW);
                    uall[q * FP + NW * jl + i] = kv ? buf_load_cf(Yb, ko, fo * K * (int)sizeof(cf)) : zero;
                }
            }
            static_for<NCH>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
                float red[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                    const cf d = kv ? buf_load_cf_cached(Db, ko, fo * K * (int)sizeof(cf)) : zero;
                    const cf p = cmul(d, uall[q * FP + NW * jl + 4 * c + e]);
                    red[2 * e] = p.re;
                    red[2 * e + 1] = p.im;
                }
                const float tot = reduce8_across_lanes(red, k);
                constexpr int fo_c = NW * j + N1 * brev(c, LBW - 2);
                if ((k & 7) == 0) sa_store_agent(pub + 2 * fo_c, tot);
            });
        });
    } else {
        cf v[N1];
#pragma unroll
        for (int h1 = 0; h1 < N1; ++h1)
            v[h1] = kv ? buf_load_cf(Tb, ko, NW * h1 * K * (int)sizeof(cf)) : zero;
        dif1<N1, false>(v, 0);
        reg_fence<N1>(v, 0, token);
#pragma unroll
        for (int i = 1; i < N1; ++i) {
            cf tw;
            sa_uload2(reinterpret_cast<const float *>(twA + i), tw.re, tw.im);
            v[i] = cmul(v[i], tw);
        }
        reg_fence<N1>(v, 0, token);
        static_for<Q>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const bool lv = !MR || q * FP + w < N1;      // (this wave's line of the group exists)
#pragma unroll
            for (int fl = 0; fl < FP; ++fl) {
                if (q * FP + fl >= N1) continue;
                const cf x = v[pos1<N1>(q * FP + fl)];
                f2 t;
                t.x = x.re;
                t.y = x.im;
                L[(fl * NW + w) * 64 + k] = t;
            }
            cf dn[4];
            auto prefetch = [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                    dn[e] = kv ? buf_load_cf_cached(Db, ko, fo * K * (int)sizeof(cf)) : zero;
                }
            };
            if (lv) prefetch(std::integral_constant<int, 0>{});
            __syncthreads();
            if (lv) {
#pragma unroll
            for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
                for (int h2 = 0; h2 < NW; ++h2) {
                    const f2 t = L[((w + NW * jl) * NW + h2) * 64 + k];
                    uall[q * FP + NW * jl + h2] = mk<float>(t.x, t.y);
                }
            }
            }
            if (q + 1 < Q) __syncthreads();
            if (lv) {
            static_for<NCH>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
                if constexpr (c == 0) dif<NW, false>(uall, q * FP + NW * jl);
                cf d[4];
                float red[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = dn[e];
                if constexpr (g + 1 < NCH) prefetch(std::integral_constant<int, g + 1>{});
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    cf p = cmul(d[e], uall[q * FP + NW * jl + 4 * c + e]);
                    if constexpr (GRAD) {
                        const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                        p = cscale(p, sa_rcp(ak * sa_uload(GH + fo) + bk));
                    }
                    red[2 * e] = p.re;
                    red[2 * e + 1] = p.im;
                }
                const float tot = reduce8_across_lanes(red, k);
                // lane 16 e holds Re, lane 16 e + 8 holds Im of the slab's partial sum for row
                // fo(e) = NW j + N1 brev(4 c + e): one store by those eight lanes
                constexpr int fo_c = NW * j + N1 * brev(c, LBW - 2);
                if ((k & 7) == 0) sa_store_agent(pub + 2 * fo_c, tot);
            });
            }   // lv
        });
    }
    // ---- publish, and wait for the other slabs of this tile ------------------------------
    sa_wait_stores();
    __syncthreads();
    if (tid == 0) sa_store_agent(flags + slab, seq);
    // what phase 2 needs besides the sums, requested before the wait: per row (one row per
    // lane, as the sums below) Sf and the Sherman-Morrison denominator; the first rows of Df
    float s_re, s_im, g_l, gh_l = 0.f;
    if (row_ok) {
        const f2 t = *reinterpret_cast<const f2 *>(S + fo_lane);
        s_re = t.x;
        s_im = t.y;
        g_l = PGM ? 0.f : G[fo_lane];
        if constexpr (GRAD) gh_l = GH[fo_lane];
    } else {
        s_re = s_im = 0.f;
        g_l = 1.f;
    }
    cf dn[4];
    auto prefetch_d = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int q = n / NCH, g = n % NCH, jl = g / CPL, c = g % CPL, j = q * LP + jl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int fo = NW * j + N1 * brev(4 * c + e, LBW);
            dn[e] = kv ? buf_load_cf_cached(Db, ko, fo * K * (int)sizeof(cf)) : zero;
        }
    };
    if (!MR || w < N1) prefetch_d(std::integral_constant<int, 0>{});
    if (tid < NH && tid != slab && !gave_up) {
        int polls = 0;
        while (sa_load_agent(flags + tid) != seq) {
            sa_spin_pause();
            if (++polls > (1 << 22)) {
                *ap->coop_err = 1;
                gave_up = true;      // (no further waiting in this launch: the result is void anyway)
                break;
            }
        }
    }
    __syncthreads();
    float qre = 0.f, qim = 0.f;
    for (int sl = 0; sl < NH; sl += 2) {
        const int mine = sl + (k >> 5);
        if (mine < NH && row_ok) {
            float a0, b0;
            sa_load_agent2(reinterpret_cast<const float *>(qp + (int64_t)mine * H + fo_lane), a0, b0);
            qre += a0;
            qim += b0;
        }
    }
    qre += __shfl_xor(qre, 32, 64);
    qim += __shfl_xor(qim, 32, 64);
    // the Sherman-Morrison coefficient of this lane's row (both halves of the wave hold it)
    cf coef_l;
    float obj_l;
    if constexpr (PGM) {
        const cf r = mk<float>(qre - s_re, qim - s_im);        // e_y = sum_k Df Yf - Sf
        coef_l = cscale(r, -ap->pgm_inv_L);
        obj_l = k < 32 ? cabs2(r) : 0.f;
        if (ap->pgm_ey && slab == 0 && k < 32) ap->pgm_ey[(int64_t)tile * H + w + fo_lane] = r;
    } else {
        if constexpr (GRAD)
            coef_l = cscale(mk<float>(s_re - rho * qre, s_im - rho * qim), sa_rcp(g_l));
        else
            coef_l = cscale(mk<float>(s_re - qre, s_im - qim), sa_rcp(g_l + rho));
        obj_l = k < 32 ? cabs2(coef_l) : 0.f;
    }

    // ---- phase 2: Sherman-Morrison with the complete sums, IFFT along H --------------------
    static_for<Q * NCH>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        constexpr int q = n / NCH, g = n % NCH, jl = g / CPL, c = g % CPL, j = q * LP + jl;
        const bool lv = !MR || q * FP + w < N1;          // (this wave's line of the group exists)
        // (the operand prefetch runs one chunk ahead: chunk n + 1 is requested when ITS line exists)
        constexpr int qn = (n + 1) / NCH;
        const bool lvn = !MR || qn * FP + w < N1;
        cf d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = dn[e];
        if constexpr (n + 1 < Q * NCH) {
            if (lvn) prefetch_d(std::integral_constant<int, n + 1>{});
        }
        if (lv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = NW * j + 4 * c + e;        // the lane that holds this row's values
            const cf coef = mk<float>(sa_readlane(coef_l.re, r), sa_readlane(coef_l.im, r));
            cf &ue = uall[q * FP + NW * jl + 4 * c + e];
            if constexpr (GRAD) {
                const float gh = sa_readlane(gh_l, r);
                const cf xn = cscale(cscale(ue, rho) + cmulc(d[e], coef), sa_rcp(ak * gh + bk));
                rg += (gh + gw) * cabs2(xn);
                ue = xn;
            } else {
                // (product first, then the sum: the four-instruction cmulc_add measured 2 % slower
                // here -- 7.22 against 7.08 ms at 1024 x 1024, profiles/r06s_config3_ab.txt)
                ue = ue + cmulc(d[e], coef);
            }
        }
        if constexpr (c == CPL - 1) {
            dit<NW, true>(uall, q * FP + NW * jl);
#pragma unroll
            for (int h2 = 1; h2 < NW; ++h2) {
                cf tw;
                sa_uload2(reinterpret_cast<const float *>(twB + NW * j + h2), tw.re, tw.im);
                uall[q * FP + NW * jl + h2] = cmulc(tw, uall[q * FP + NW * jl + h2]);
            }
        }
        }   // lv
        if constexpr (g == NCH - 1) {
            {
                float &rg_ = rg;
                int &tk_ = token;
                SA_VGPR_FENCE3(rg_, tk_, tk_);
            }
            if (lv) {
#pragma unroll
            for (int jl2 = 0; jl2 < LP; ++jl2) {
#pragma unroll
                for (int h2 = 0; h2 < NW; ++h2) {
                    f2 t;
                    t.x = uall[q * FP + NW * jl2 + h2].re;
                    t.y = uall[q * FP + NW * jl2 + h2].im;
                    L[((w + NW * jl2) * NW + h2) * 64 + k] = t;
                }
            }
            }
            __syncthreads();
            // (back into the group's own registers: rows h1 = pos(q FP + fl) of the last stage)
#pragma unroll
            for (int fl = 0; fl < FP; ++fl) {
                if (q * FP + fl >= N1) continue;
                const f2 t = L[(fl * NW + w) * 64 + k];
                uall[q * FP + fl] = mk<float>(t.x, t.y);
            }
            if (q + 1 < Q) __syncthreads();
        }
    });
    cf v[N1];
#pragma unroll
    for (int i = 0; i < N1; ++i) v[pos1<N1>(i)] = uall[i];
    reg_fence<N1>(v, 0, token);
    dit1<N1, true>(v, 0);
#pragma unroll
    for (int h1 = 0; h1 < N1; ++h1)
        if (kv) buf_store_cf(Tb, ko, NW * h1 * K * (int)sizeof(cf), v[h1]);

    // every slab computes the same |coef|^2: slab 0 reports it
    const double pw = (wf == 0 || ((ap->c.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
    if constexpr (GRAD) {
        const float wk = (ap->c.wg && kv) ? ap->c.wg[slab * 64 + k] : 1.f;
        double acc[2] = {slab == 0 ? (double)obj_l * pw : 0.0, kv ? (double)(rg * wk) * pw : 0.0};
        block_sum_store<2>(acc, scratch, ap->c.partials + 2 * ((int64_t)tile * NH + slab));
    } else if constexpr (PGM) {
        double acc[1] = {(double)obj_l};
        if (slab == 0) block_sum_store<1>(acc, scratch, ap->c.partials + tile);
    } else {
        double acc[1] = {(double)obj_l * pw * (double)rho * (double)rho};
        if (slab == 0) block_sum_store<1>(acc, scratch, ap->c.partials + tile);
    }
    __syncthreads();      // (scratch and the exchange buffer are reused by the next tile)
    }
}

template <typename E>
__global__ void __launch_bounds__(256) permute_ab_kernel(const E *__restrict__ in,
                                                         E *__restrict__ out, int64_t A, int64_t B,
                                                         int64_t C, int64_t Cin, int64_t Cout) {
    const int64_t n = A * B * C;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n;
         o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = o % C, ba = o / C;
        const int64_t aa = ba % A, bb = ba / A;
        out[ba * Cout + c] = in[(aa * B + bb) * Cin + c];
    }
}

}  // namespace

template <typename E>
void launch_permute_ab(hipStream_t st, const E *in, E *out, int64_t A, int64_t B, int64_t C,
                       int64_t in_stride, int64_t out_stride) {
    const int64_t n = A * B * C;
    if (n <= 0) return;
    int64_t g = ceil_div(n, 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL((permute_ab_kernel<E>), dim3((unsigned)g), dim3(256), 0, st, in, out, A, B, C,
                       in_stride ? in_stride : C, out_stride ? out_stride : C);
    SA_HIP(hipGetLastError());
}

// Split of a supported shape into N1 = 32 (in-register FFT length) x NW = H/32 waves,
// and the number LP of stage-2 lines per exchange group.  (An 8-wave x 64-point
// layout of H = 512 was measured too: ~200 VGPRs, 2 waves/SIMD, 1.4x slower.)
struct FusedSplit {
    int N1, NW, LP;
};
static FusedSplit fused_split(int H, int K) {
    (void)K;
    if (H == 128) return {32, 4, 4};
    if (H == 256) return {32, 8, 2};
    if (fused_mr_height(H)) return {H / 16, 16, 1};     // 160 ... 480: 10 ... 30 rows per thread
    return {32, 16, 1};
}
static int fused_rev(int N1, int i) {
    switch (N1) {
#define SA_MR_CASE(n) case n: return mr_rev<n>(i);
    SA_MR_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: return brev(i, ilog2(N1));
    }
}
bool fused_mr_height(int H) { return H % 16 == 0 && mr_length(H / 16); }
// (the second table has (lines per wave) x 16 entries for each of the 16 waves)
int fused_twiddle_count(int H) { return fused_mr_height(H) ? (H > 256 ? 512 : 256) : H; }

template <typename T> void fused_twiddles(int H, int K, cx<T> *twA, cx<T> *twB) {
    const FusedSplit sp = fused_split(H, K);
    const int N1 = sp.N1, NW = sp.NW;
    // (mixed-radix heights: two stage-2 lines per wave, the second one only while w + 16 < N1; the
    // second table has J NW = 32 entries per wave -- fused_twiddle_count(H) in all)
    const int J = fused_mr_height(H) ? (N1 > NW ? 2 : 1) : N1 / NW;
    const double two_pi = 6.283185307179586476925286766559;
    for (int w = 0; w < NW; ++w) {
        for (int i = 0; i < N1; ++i) {
            const double ang = -two_pi * (double)(w * fused_rev(N1, i)) / (double)H;
            twA[w * N1 + i] = mk<T>((T)std::cos(ang), (T)std::sin(ang));
        }
        for (int j = 0; j < J; ++j)
            for (int h2 = 0; h2 < NW; ++h2) {
                const double ang = -two_pi * (double)((w + NW * j) * h2) / (double)H;
                twB[w * (J * NW) + NW * j + h2] = mk<T>((T)std::cos(ang), (T)std::sin(ang));
            }
    }
}
template void fused_twiddles<float>(int, int, cx<float> *, cx<float> *);
template void fused_twiddles<double>(int, int, cx<double> *, cx<double> *);

template <> bool fused_cols_supported<float>(int H, int K) {
    return (H == 128 || H == 256 || H == 512 || fused_mr_height(H)) && K >= 1 && K <= 64;
}
template <> bool fused_cols_supported<double>(int, int) { return false; }

// Workgroups of a persistent launch: as many as the device holds at once (16-wave
// workgroups: one per CU; 8-wave ones: two), a multiple of 8 so that the XCD of a workgroup
// is blockIdx % 8 for every slot it walks.
static int64_t persistent_grid(int NW) {
    const int cus = current_device_cus();
    if (NW != 16) return INT64_MAX;      // (the 8-wave kernel takes one tile per workgroup)
    return std::max<int64_t>(8, cus / 8 * 8);
}

template <int N1, int NW, int LP, int KC, bool GRAD, bool KRT = false, bool PER_TILE = false,
          int DBG = 0>
static void launch_fused_inst(hipStream_t st, const FusedColsArgs<float> &a, int64_t ntiles) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        SA_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void *>(
                &fused_cols_kernel<N1, NW, LP, KC, GRAD, KRT, PER_TILE, DBG>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds_bytes(NW, LP)));
    }
    const int64_t wf_groups = ceil_div(a.W / 2 + 1, 8);   // see the tile mapping in the kernel
    const int64_t all = wf_groups * 8 * a.CN;
    hipLaunchKernelGGL((fused_cols_kernel<N1, NW, LP, KC, GRAD, KRT, PER_TILE, DBG>),
                       dim3((unsigned)std::min<int64_t>(all, persistent_grid(KC == 64 ? (N1 == 64 ? 16 : NW) : 0))),
                       dim3(NW * 64),
                       fused_lds_bytes(NW, LP), st, a);
}

template <int N1, int NW, int LP>
static void launch_fused_k(hipStream_t st, const FusedColsArgs<float> &a, int64_t ntiles) {
    const bool grad = a.g1t != nullptr;
    if (a.per_tile) {
        SA_REQUIRE(!grad, "per-tile operands do not combine with the gradient term");
        if (a.K == 64) launch_fused_inst<N1, NW, LP, 64, false, false, true>(st, a, ntiles);
        else launch_fused_inst<N1, NW, LP, 0, false, false, true>(st, a, ntiles);
    } else if (a.Kv == 64 && a.K > 64) {
        if (grad) launch_fused_inst<N1, NW, LP, 64, true, true>(st, a, ntiles);
        else launch_fused_inst<N1, NW, LP, 64, false, true>(st, a, ntiles);
    } else if (a.K == 64) {
        // (coef_out on a K <= 64 system: the instantiation that stores the multipliers -- the
        // mask-decoupled X-step reads D x = Sf - rho coef off them, api_maskdcpl.inc)
        if (grad) launch_fused_inst<N1, NW, LP, 64, true>(st, a, ntiles);
        else if (a.coef_out) launch_fused_inst<N1, NW, LP, 64, false, true>(st, a, ntiles);
        else launch_fused_inst<N1, NW, LP, 64, false>(st, a, ntiles);
    } else {
        if (grad) launch_fused_inst<N1, NW, LP, 0, true>(st, a, ntiles);
        else if (a.coef_out) launch_fused_inst<N1, NW, LP, 0, false, true>(st, a, ntiles);
        else launch_fused_inst<N1, NW, LP, 0, false>(st, a, ntiles);
    }
}

template <> int64_t launch_fused_cols<float>(hipStream_t st, const FusedColsArgs<float> &a_in) {
    SA_REQUIRE(fused_cols_supported<float>(a_in.H, a_in.Kv ? a_in.Kv : a_in.K),
               "shape not handled by the fused column kernel");
    const int64_t ntiles = (int64_t)(a_in.W / 2 + 1) * a_in.CN;
    const FusedSplit sp = fused_split(a_in.H, a_in.Kv ? a_in.Kv : a_in.K);
    FusedColsArgs<float> a = a_in;
    // start-up stagger of the persistent workgroups: 4 phase groups 2 x 8128 cycles apart (about a
    // fifth of a tile's time each): measured 1.13 -> 1.06 ms at 512 x 512, K = 64, N = 32
    // (profiles/r02_fused_cols_notes.md)
    a.stagger_groups = kColsStaggerGroups;
    a.stagger_sleeps = kColsStaggerSleeps;
    if (sp.N1 != 32) {
        // mixed-radix heights: the plain system, the gradient term, or the multipliers stored (mask
        // decoupling); no per-tile operands -- the API layer keeps everything else on the generic chain
        SA_REQUIRE(!a.per_tile && !(a.Kv == 64 && a.K > 64) && !(a.coef_out && a.g1t),
                   "mixed-radix heights: the plain and the gradient-regularised column pass only");
        const bool k64 = a.K == 64, grad = a.g1t != nullptr, krt = a.coef_out != nullptr;
        switch (sp.N1) {
#define SA_MR_CASE(n)                                                                     \
    case n:                                                                               \
        if (grad) k64 ? launch_fused_inst<n, 16, 1, 64, true>(st, a, ntiles)              \
                      : launch_fused_inst<n, 16, 1, 0, true>(st, a, ntiles);              \
        else if (krt) k64 ? launch_fused_inst<n, 16, 1, 64, false, true>(st, a, ntiles)   \
                          : launch_fused_inst<n, 16, 1, 0, false, true>(st, a, ntiles);   \
        else k64 ? launch_fused_inst<n, 16, 1, 64, false>(st, a, ntiles)                  \
                 : launch_fused_inst<n, 16, 1, 0, false>(st, a, ntiles);                  \
        break;
        SA_MR_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
        default: SA_REQUIRE(false, "height not handled by the mixed-radix column kernel");
        }
    } else if (sp.NW == 4)
        launch_fused_k<32, 4, 4>(st, a, ntiles);
    else if (sp.NW == 8)
        launch_fused_k<32, 8, 2>(st, a, ntiles);
    else
        launch_fused_k<32, 16, 1>(st, a, ntiles);
    SA_HIP(hipGetLastError());
    return ntiles;
}
template <int NW, int LP, int KC, int N1 = 32>
static void launch_dualres_inst(hipStream_t st, const FusedColsArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cols_dualres_kernel<NW, LP, KC, N1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds_bytes(NW, LP)));
    }
    const int64_t ntiles = (int64_t)(a.W / 2 + 1) * a.CN;
    const int64_t cus = current_device_cus();
    // (one 16-wave workgroup fills a CU; two 8-wave, four 4-wave ones share it)
    const int64_t grid = std::min<int64_t>(ntiles, cus * (16 / NW));
    hipLaunchKernelGGL((cols_dualres_kernel<NW, LP, KC, N1>), dim3((unsigned)grid), dim3(NW * 64),
                       fused_lds_bytes(NW, LP), st, a);
}
template <> int64_t launch_cols_dualres<float>(hipStream_t st, const FusedColsArgs<float> &a) {
    SA_REQUIRE(fused_cols_supported<float>(a.H, a.K), "shape not handled by the fused column kernel");
    const FusedSplit sp = fused_split(a.H, a.K);
    const bool k64 = a.K == 64 && (a.Ks == 0 || a.Ks == 64);
    if (sp.N1 != 32) {
        switch (sp.N1) {
#define SA_MR_CASE(n) \
    case n: k64 ? launch_dualres_inst<16, 1, 64, n>(st, a) : launch_dualres_inst<16, 1, 0, n>(st, a); break;
        SA_MR_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
        default: SA_REQUIRE(false, "height not handled by the mixed-radix column kernel");
        }
    } else if (sp.NW == 4) k64 ? launch_dualres_inst<4, 4, 64>(st, a) : launch_dualres_inst<4, 4, 0>(st, a);
    else if (sp.NW == 8) k64 ? launch_dualres_inst<8, 2, 64>(st, a) : launch_dualres_inst<8, 2, 0>(st, a);
    else k64 ? launch_dualres_inst<16, 1, 64>(st, a) : launch_dualres_inst<16, 1, 0>(st, a);
    SA_HIP(hipGetLastError());
    return (int64_t)(a.W / 2 + 1) * a.CN;
}
template <> int64_t launch_cols_dualres<double>(hipStream_t, const FusedColsArgs<double> &) {
    throw Error(-1, "the fused column kernel is float32 only");
}
__global__ void __launch_bounds__(256) gram_rows_kernel(const cf *__restrict__ z,
                                                        float *__restrict__ out, int64_t nrows,
                                                        int K) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= nrows) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += cabs2(z[row * K + k]);
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) out[row] = s;
}
template <> void launch_gram_rows<float>(hipStream_t st, const cx<float> *z, float *out,
                                         int64_t nrows, int K) {
    hipLaunchKernelGGL(gram_rows_kernel, dim3((unsigned)ceil_div(nrows, 4)), dim3(256), 0, st, z, out,
                       nrows, K);
    SA_HIP(hipGetLastError());
}
template <> void launch_gram_rows<double>(hipStream_t, const cx<double> *, double *, int64_t, int) {
    throw Error(-1, "the fused column kernel is float32 only");
}
template <> void launch_grad_g1<float>(hipStream_t st, const FusedColsArgs<float> &a) {
    const int64_t nrows = (int64_t)(a.W / 2 + 1) * a.H;
    hipLaunchKernelGGL(grad_g1_kernel, dim3((unsigned)ceil_div(nrows, 4)), dim3(256), 0, st, a);
    SA_HIP(hipGetLastError());
}
template <> void launch_grad_g1<double>(hipStream_t, const FusedColsArgs<double> &) {
    throw Error(-1, "the fused column kernel is float32 only");
}
// ---------------------------------------------------------------------------
// a few more than 64 filters: the column pass of the filters >= Kv (csc_fused.h)
// ---------------------------------------------------------------------------
// sft_eff[tile][f] = sft[tile][f] - sum_{k >= Kv} dft[wf][f][k] t[tile][f][k]
// (gradient-regularised system, a.g1t set: - rho sum_{k >= Kv} dft t / dd_k, with
// dd_k = mu wg_k (ghh[f] + ghw[wf]) + rho as in the column kernel)
__global__ void __launch_bounds__(256) tail_inner_kernel(const FusedColsArgs<float> a,
                                                         const cf *__restrict__ sft,
                                                         cf *__restrict__ sft_eff, int64_t ntiles) {
    const int H = a.H, K = a.K, Kv = a.Kv, Ks = a.Ks ? a.Ks : a.K;
    const bool grad = a.g1t != nullptr;
    const int64_t total = ntiles * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tile = i / H;
        const int f = (int)(i - tile * H);
        const int64_t wf = tile / a.CN;
        const cf *d = a.dft + (wf * H + f) * Ks, *x = a.t + i * Ks;
        const float gh = grad ? a.ghh[f] + a.ghw[wf] : 0.f;
        cf q = mk<float>(0.f, 0.f);
        for (int k = Kv; k < K; ++k) {
            cf p = cmul(d[k], x[k]);
            if (grad) p = cscale(p, a.rho / (a.mu * (a.wg ? a.wg[k] : 1.f) * gh + a.rho));
            q = q + p;
        }
        sft_eff[i] = sft[i] - q;
    }
}

// t[tile][f][k] += conj(dft[wf][f][k]) coef[tile][f] for k >= Kv; one workgroup per tile,
// thread = column frequency.  Gradient-regularised system: t = (rho t + conj(dft) coef) / dd_k,
// and the tail's share of the gradient term is added to the tile's second partial.
__global__ void __launch_bounds__(512) tail_update_kernel(const FusedColsArgs<float> a) {
    const int H = a.H, K = a.K, Kv = a.Kv, Ks = a.Ks ? a.Ks : a.K;
    const bool grad = a.g1t != nullptr;
    const int64_t tile = blockIdx.x;
    const int64_t wf = tile / a.CN;
    const int Wf = a.W / 2 + 1;
    double acc[1] = {0.0};
    for (int f = threadIdx.x; f < H; f += blockDim.x) {
        const int64_t row = tile * H + f;
        const cf cf_ = a.coef_out[row];
        const cf *d = a.dft + (wf * H + f) * Ks;
        cf *x = a.t + row * Ks;
        const float gh = grad ? a.ghh[f] + a.ghw[wf] : 0.f;
        for (int k = Kv; k < K; ++k) {
            if (grad) {
                const float wk = a.wg ? a.wg[k] : 1.f;
                const cf xn = cscale(cscale(x[k], a.rho) + cmulc(d[k], cf_),
                                     1.f / (a.mu * wk * gh + a.rho));
                x[k] = xn;
                acc[0] += (double)(gh * wk * cabs2(xn));
            } else {
                x[k] = x[k] + cmulc(d[k], cf_);
            }
        }
    }
    if (grad) {
        double *scratch = dyn_lds<double>();
        block_sum_store<1>(acc, scratch, scratch + 12);   // (thread 0 writes, thread 0 reads)
        if (threadIdx.x == 0) {
            const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
            a.partials[2 * tile + 1] += scratch[12] * pw;
        }
    }
}

template <> void launch_tail_inner<float>(hipStream_t st, const FusedColsArgs<float> &a,
                                          const cx<float> *sft, cx<float> *sft_eff) {
    const int64_t ntiles = (int64_t)(a.W / 2 + 1) * a.CN;
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(ntiles * a.H, 256), 65535);
    hipLaunchKernelGGL(tail_inner_kernel, dim3(grid), dim3(256), 0, st, a, sft, sft_eff, ntiles);
    SA_HIP(hipGetLastError());
}
template <> void launch_tail_update<float>(hipStream_t st, const FusedColsArgs<float> &a) {
    const int64_t ntiles = (int64_t)(a.W / 2 + 1) * a.CN;
    hipLaunchKernelGGL(tail_update_kernel, dim3((unsigned)ntiles), dim3(a.H), sizeof(double) * 16, st, a);
    SA_HIP(hipGetLastError());
}
template <> void launch_tail_inner<double>(hipStream_t, const FusedColsArgs<double> &,
                                           const cx<double> *, cx<double> *) {
    throw Error(-1, "the fused column kernel is float32 only");
}
template <> void launch_tail_update<double>(hipStream_t, const FusedColsArgs<double> &) {
    throw Error(-1, "the fused column kernel is float32 only");
}

template <> bool fused_slabs_supported<float>(int H, int K) {
    return (H == 128 || H == 256 || H == 512 || fused_mr_height(H)) && K > 64 && K <= 256 && K % 2 == 0;
}
template <> bool fused_slabs_supported<double>(int, int) { return false; }

template <int NW, int LP, int KS, bool GRAD>
static void launch_slabs(hipStream_t st, const FusedSlabArgs<float> &a, bool second) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        for (const void *f :
             {reinterpret_cast<const void *>(&cols_fwd_partial_kernel<NW, LP, KS, GRAD>),
              reinterpret_cast<const void *>(&cols_sm_apply_inv_kernel<NW, LP, KS, GRAD>)})
            SA_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)fused_lds_bytes(NW, LP)));
    }
    const dim3 grid((unsigned)(ceil_div(a.c.W / 2 + 1, 8) * 8 * a.c.CN), (unsigned)ceil_div(a.c.K, 64));
    if (!second)
        hipLaunchKernelGGL((cols_fwd_partial_kernel<NW, LP, KS, GRAD>), grid, dim3(NW * 64),
                           fused_lds_bytes(NW, LP), st, a);
    else
        hipLaunchKernelGGL((cols_sm_apply_inv_kernel<NW, LP, KS, GRAD>), grid, dim3(NW * 64),
                           fused_lds_bytes(NW, LP), st, a);
    SA_HIP(hipGetLastError());
}

// Workgroups of the one-launch form: NH per tile side by side, as many groups as the device
// holds at once with one workgroup per CU (a multiple of 8 groups: the residue of the row
// frequencies a group walks stays fixed).
template <int NW, int LP, int KS, bool GRAD, int N1 = 32>
static void launch_slab_coop(hipStream_t st, const FusedSlabArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cols_slab_coop_kernel<NW, LP, KS, GRAD, false, N1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)fused_lds_bytes(NW, LP)));
    }
    const int cus = current_device_cus();
    const int NH = (int)ceil_div(a.c.K, 64);
    int groups = (cus / NH) & ~7;
#ifdef SPORCO_AMD_HOSTSIM
    groups = 8;
    hostsim::set_coop(NH);     // (the CPU test simulator runs the NH partners side by side)
#endif
    SA_REQUIRE(groups >= 8, "too few compute units for cooperating slab workgroups");
    const int64_t slots = ceil_div(a.c.W / 2 + 1, 8) * a.c.CN;
    if ((int64_t)(groups >> 3) > slots) groups = (int)slots * 8;
    hipLaunchKernelGGL((cols_slab_coop_kernel<NW, LP, KS, GRAD, false, N1>), dim3((unsigned)(groups * NH)),
                       dim3(NW * 64), fused_lds_bytes(NW, LP), st, a);
    SA_HIP(hipGetLastError());
}
template <> int64_t launch_cols_slab_coop<float>(hipStream_t st, const FusedSlabArgs<float> &a_in) {
    SA_REQUIRE(fused_slabs_supported<float>(a_in.c.H, a_in.c.K), "shape not handled by the slab column kernels");
    SA_REQUIRE(a_in.coop_flags && a_in.coop_err, "the cooperating slab kernel needs its flag buffers");
    FusedSlabArgs<float> a = a_in;
    a.c.stagger_groups = kColsStaggerGroups;
    a.c.stagger_sleeps = kColsStaggerSleeps;
    const bool g = a.c.g1t != nullptr;
    if (fused_mr_height(a.c.H)) {
        // mixed-radix heights: run-time K, plain and gradient-regularised systems
        switch (a.c.H / 16) {
#define SA_MR_CASE(n)                                                                          \
    case n:                                                                                    \
        if (g) launch_slab_coop<16, 1, 0, true, n>(st, a);                                     \
        else launch_slab_coop<16, 1, 0, false, n>(st, a);                                      \
        break;
        SA_MR_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
        default: SA_REQUIRE(false, "height not handled by the mixed-radix slab kernel");
        }
        return (int64_t)(a.c.W / 2 + 1) * a.c.CN;
    }
    if (a.c.H == 128) {
        if (a.c.K == 128) { if (g) launch_slab_coop<4, 4, 128, true>(st, a); else launch_slab_coop<4, 4, 128, false>(st, a); }
        else { if (g) launch_slab_coop<4, 4, 0, true>(st, a); else launch_slab_coop<4, 4, 0, false>(st, a); }
    } else if (a.c.H == 256) {
        if (a.c.K == 128) { if (g) launch_slab_coop<8, 2, 128, true>(st, a); else launch_slab_coop<8, 2, 128, false>(st, a); }
        else { if (g) launch_slab_coop<8, 2, 0, true>(st, a); else launch_slab_coop<8, 2, 0, false>(st, a); }
    } else {
        if (a.c.K == 128) { if (g) launch_slab_coop<16, 1, 128, true>(st, a); else launch_slab_coop<16, 1, 128, false>(st, a); }
        else { if (g) launch_slab_coop<16, 1, 0, true>(st, a); else launch_slab_coop<16, 1, 0, false>(st, a); }
    }
    return (int64_t)(a.c.W / 2 + 1) * a.c.CN;
}
template <int NW, int LP, int KS>
static void launch_pgm_grad_coop(hipStream_t st, const FusedSlabArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        SA_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void *>(&cols_slab_coop_kernel<NW, LP, KS, false, true>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds_bytes(NW, LP)));
    }
    const int cus = current_device_cus();
    const int NH = (int)ceil_div(a.c.K, 64);
    int groups = (cus / NH) & ~7;
#ifdef SPORCO_AMD_HOSTSIM
    groups = 8;
    hostsim::set_coop(NH);
#endif
    SA_REQUIRE(groups >= 8, "too few compute units for cooperating slab workgroups");
    const int64_t slots = ceil_div(a.c.W / 2 + 1, 8) * a.c.CN;
    if ((int64_t)(groups >> 3) > slots) groups = (int)slots * 8;
    hipLaunchKernelGGL((cols_slab_coop_kernel<NW, LP, KS, false, true>), dim3((unsigned)(groups * NH)),
                       dim3(NW * 64), fused_lds_bytes(NW, LP), st, a);
    SA_HIP(hipGetLastError());
}
template <> int64_t launch_pgm_grad_slabs<float>(hipStream_t st, const FusedSlabArgs<float> &a_in) {
    SA_REQUIRE(fused_slabs_supported<float>(a_in.c.H, a_in.c.K), "shape not handled by the slab column kernels");
    SA_REQUIRE(a_in.coop_flags && a_in.coop_err && a_in.pgm_yf, "the cooperating slab kernel needs its buffers");
    FusedSlabArgs<float> a = a_in;
    a.c.stagger_groups = 1;
    a.c.stagger_sleeps = 0;
    if (a.c.H == 128) {
        if (a.c.K == 128) launch_pgm_grad_coop<4, 4, 128>(st, a);
        else launch_pgm_grad_coop<4, 4, 0>(st, a);
    } else if (a.c.H == 256) {
        if (a.c.K == 128) launch_pgm_grad_coop<8, 2, 128>(st, a);
        else launch_pgm_grad_coop<8, 2, 0>(st, a);
    } else {
        if (a.c.K == 128) launch_pgm_grad_coop<16, 1, 128>(st, a);
        else launch_pgm_grad_coop<16, 1, 0>(st, a);
    }
    return (int64_t)(a.c.W / 2 + 1) * a.c.CN;
}
template <> int64_t launch_pgm_grad_slabs<double>(hipStream_t, const FusedSlabArgs<double> &) {
    throw Error(-1, "the fused column kernels are float32 only");
}
template <> int64_t launch_cols_slab_coop<double>(hipStream_t, const FusedSlabArgs<double> &) {
    throw Error(-1, "the fused column kernels are float32 only");
}

template <int NW, int LP, int KS>
static void launch_slabs_g(hipStream_t st, const FusedSlabArgs<float> &a, bool second) {
    if (a.c.g1t) launch_slabs<NW, LP, KS, true>(st, a, second);
    else launch_slabs<NW, LP, KS, false>(st, a, second);
}

static void launch_slabs_any(hipStream_t st, const FusedSlabArgs<float> &a, bool second) {
    SA_REQUIRE(fused_slabs_supported<float>(a.c.H, a.c.K), "shape not handled by the slab column kernels");
    if (a.c.H == 128) {
        if (a.c.K == 128) launch_slabs_g<4, 4, 128>(st, a, second);
        else launch_slabs_g<4, 4, 0>(st, a, second);
    } else if (a.c.H == 256) {
        if (a.c.K == 128) launch_slabs_g<8, 2, 128>(st, a, second);
        else launch_slabs_g<8, 2, 0>(st, a, second);
    } else {
        if (a.c.K == 128) launch_slabs_g<16, 1, 128>(st, a, second);
        else launch_slabs_g<16, 1, 0>(st, a, second);
    }
}
template <> void launch_cols_fwd_partial<float>(hipStream_t st, const FusedSlabArgs<float> &a) {
    launch_slabs_any(st, a, false);
}
template <> int64_t launch_cols_sm_apply_inv<float>(hipStream_t st, const FusedSlabArgs<float> &a) {
    launch_slabs_any(st, a, true);
    return (int64_t)(a.c.W / 2 + 1) * a.c.CN;
}
template <> void launch_cols_fwd_partial<double>(hipStream_t, const FusedSlabArgs<double> &) {
    throw Error(-1, "the fused column kernels are float32 only");
}
template <> int64_t launch_cols_sm_apply_inv<double>(hipStream_t, const FusedSlabArgs<double> &) {
    throw Error(-1, "the fused column kernels are float32 only");
}

template <> int64_t launch_fused_cols<double>(hipStream_t, const FusedColsArgs<double> &) {
    throw Error(-1, "the fused column kernel is float32 only");
}

template void launch_permute_ab<float>(hipStream_t, const float *, float *, int64_t, int64_t, int64_t,
                                       int64_t, int64_t);
template void launch_permute_ab<double>(hipStream_t, const double *, double *, int64_t, int64_t,
                                        int64_t, int64_t, int64_t);
template void launch_permute_ab<cx<float>>(hipStream_t, const cx<float> *, cx<float> *, int64_t,
                                           int64_t, int64_t, int64_t, int64_t);
template void launch_permute_ab<cx<double>>(hipStream_t, const cx<double> *, cx<double> *, int64_t,
                                            int64_t, int64_t, int64_t, int64_t);

}  // namespace sporco_amd
