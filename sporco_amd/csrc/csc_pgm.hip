// csc_pgm.hip -- column passes of the fused PGM iteration (see csc_pgm.h).
//
// Same decomposition as csc_fused.hip (H = 32 x NW, lane = filter, one (wf, cn)
// tile per workgroup, XCD-aware tile order), but each kernel needs only one of
// the two LDS exchanges: the iterates live in the frequency domain, so
//   pgm_grad_ifft     starts on the spectral side (wave w owns the frequencies
//                     f = w + NW j + 32 brev(i), exactly the order the inverse
//                     DIT transform wants), and ends with rows of T;
//   pgm_fft_momentum  starts with rows of T' and ends on the spectral side, where
//                     the momentum update and the sums are element-wise.
#include "csc_pgm.h"

#include "csc_fused.h"
#include "regfft.h"

namespace sporco_amd {

namespace {

using namespace regfft;

constexpr size_t pgm_lds_bytes(int NW, int LP) {
    return sizeof(f2) * LP * NW * NW * 64 + sizeof(double) * kPgmPartialStride * 16;
}

// sum over the K filters of d[e] * x[e] for 4 frequencies e at once (wave reduction);
// returns the 4 complex totals as wave-uniform values
__device__ __forceinline__ void inner4(const cf (&d)[4], const cf *x, int lane, cf (&q)[4]) {
    float red[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const cf p = cmul(d[e], x[e]);
        red[2 * e] = p.re;
        red[2 * e + 1] = p.im;
    }
    const float tot = reduce8_across_lanes(red, lane);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        q[e] = mk<float>(sa_readlane(tot, 16 * e), sa_readlane(tot, 16 * e + 8));
}

// ---------------------------------------------------------------------------
// Vf = Yf - conj(Df) (sum_k Df Yf - Sf) / L;  T = IFFT_H(Vf)      (grad_f + the step,
// sporco/pgm/cbpdn.py:263-279, sporco/pgm/pgm.py:800)
// ---------------------------------------------------------------------------
// BT: a held (backtracking) trial -- e_y = sum_k Df Yf - Sf is stored per frequency and f(Yf)
// summed; the default iteration needs neither and compiles them out.
// With 16 waves and K = 64 the launch is persistent (one workgroup per CU walking its XCD's
// tile list, staggered start): see fused_cols_kernel, whose measurements carried over.
// EYIN: the residual per frequency comes from memory (a.ey_in) instead of being formed from
// Yf -- the masked classes, whose residual passes through the spatial domain for the mask
// between the inner product and the gradient (pgm/cbpdn.py:454-477); no wave reduction then.
// N1: rows per thread -- 32, or one of the mixed-radix lengths (regfft.h SA_MR_LENGTHS; 16 waves, LP = 1:
// the exchange groups of csc_fused_body.inc, the last one partly filled)
template <int NW, int LP, int KC, bool BT, bool PERS = false, bool EYIN = false, int N1 = 32>
__global__ void __launch_bounds__(NW * 64) pgm_grad_ifft_kernel(const PgmColsArgs<float> a) {
    static_assert(!(BT && EYIN), "a held trial forms its own residual");
    constexpr bool MR = mr_length(N1);
    constexpr int H = N1 * NW, J = MR ? (N1 > NW ? 2 : 1) : N1 / NW;
    static_assert(!MR || (NW == 16 && LP == 1), "mixed-radix heights: 16 waves, one line per group");
    constexpr int LBW = ilog2(NW);
    constexpr int FP = LP * NW, Q = J / LP, CPL = NW / 4, NCH = LP * CPL;
    constexpr bool PERSIST = PERS;
    static_assert(!PERS || (NW == 16 && KC == 64), "persistent form: 16 waves, K = 64");
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int K = KC ? KC : a.K;
    const bool kv = KC == 64 ? true : k < K;
    const int xcd = blockIdx.x & 7;
    const int ko = (w * K + k) * (int)sizeof(cf);   // row w, filter k
    f2 *L = dyn_lds<f2>();
    double *scratch = reinterpret_cast<double *>(L + FP * NW * 64);
    const cf zero = mk<float>(0.f, 0.f);
    int token = 0;
    if constexpr (PERSIST) {
        const int ph = (int)(blockIdx.x >> 3) % a.stagger_groups;
        for (int i = 0; i < ph * a.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (int slot = blockIdx.x >> 3;; slot += gridDim.x >> 3) {
    SA_ARGS_PTR_T(PgmColsArgs<float>) ap = sa_args_reload<PERSIST>(a);
    const int Wf = ap->W / 2 + 1, CN = ap->CN;
    if (slot >= ((Wf + 7) / 8) * CN) break;
    const int wf = (slot / CN) * 8 + xcd;   // see fused_cols_kernel for the tile order
    if (wf >= Wf) break;
    const int tile = wf * CN + slot % CN;
    const uint32_t tbytes = (uint32_t)(H * K * sizeof(cf));
    const BufRsrc Yb = make_rsrc(ap->yf + (int64_t)tile * H * K, tbytes);
    const BufRsrc Ob = make_rsrc(ap->t + (int64_t)tile * H * K, tbytes);
    const BufRsrc Db = make_rsrc(ap->dft + (int64_t)wf * H * K, tbytes);
    const cf *S = (EYIN ? ap->ey_in : ap->sft) + (int64_t)tile * H + w;
    cf *EY = BT ? ap->ey + (int64_t)tile * H + w : nullptr;
    const cf *twB = ap->twB + w * (J * NW);
    const float inv_L = ap->inv_L;
    float fsum = 0.f;

    cf v[N1];
    static_for<Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const bool lv = !MR || q * FP + w < N1;      // (this wave's line of the group exists)
        if (lv) {
        // this group's lines of Yf and the matching rows of Df
        cf u[FP];
#pragma unroll
        for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int fo = NW * (q * LP + jl) + N1 * brev(i, LBW);   // f - w
                u[NW * jl + i] = kv ? buf_load_cf(Yb, ko, fo * K * (int)sizeof(cf)) : zero;
            }
        }
        static_for<NCH>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
            cf dd[4], sv[4], qq[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                dd[e] = kv ? buf_load_cf_cached(Db, ko, fo * K * (int)sizeof(cf)) : zero;
                sa_uload2(reinterpret_cast<const float *>(S + fo), sv[e].re, sv[e].im);
            }
            if constexpr (!EYIN) inner4(dd, &u[NW * jl + 4 * c], k, qq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                cf r;
                if constexpr (EYIN) r = sv[e];          // the (masked) residual, from memory
                else r = qq[e] - sv[e];                 // sum_k Df Yf - Sf
                if constexpr (BT) {
                    fsum = cabs2_add(fsum, r);
                    if (k == 0) EY[NW * j + N1 * brev(4 * c + e, LBW)] = r;
                }
                u[NW * jl + 4 * c + e] = u[NW * jl + 4 * c + e] - cscale(cmulc(dd[e], r), inv_L);
            }
            if constexpr (c == CPL - 1) {
                // inverse FFT over f2, conj twiddle
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
            float &fs_ = fsum;
            int &tk_ = token;
            SA_VGPR_FENCE3(fs_, tk_, tk_);
        }
        // exchange: (wave = f1 mod NW; h2 in registers) -> (wave = h2; f1 in registers)
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
        }   // lv
        __syncthreads();
#pragma unroll
        for (int fl = 0; fl < FP; ++fl) {
            if (q * FP + fl >= N1) continue;
            const f2 t = L[(fl * NW + w) * 64 + k];
            v[pos1<N1>(q * FP + fl)] = mk<float>(t.x, t.y);
        }
        if (q + 1 < Q) __syncthreads();
    });
    reg_fence<N1>(v, 0, token);
    // (the tile's sum before the last transform: the stores are then the last thing a wave does
    // for this tile, and the next tile's loads follow them directly)
    if constexpr (BT) {
        double acc[1] = {k == 0 ? (double)fsum : 0.0};
        block_sum_store<1>(acc, scratch, ap->partials + tile);
    } else if constexpr (PERSIST) {
        __syncthreads();     // the exchange buffer is reused by the next tile
    }
    dit1<N1, true>(v, 0);
    if (kv) {
#pragma unroll
        for (int h1 = 0; h1 < N1; ++h1) buf_store_cf(Ob, ko, NW * h1 * K * (int)sizeof(cf), v[h1]);
    }
    if constexpr (!PERSIST) break;
    }   // persistent loop over this workgroup's tiles
}

// ---------------------------------------------------------------------------
// Xf' = FFT_H(T');  Yf' = Xf' + beta (Xf' - Xf);  sums of |Xf' - Yf|^2 and f(Xf')
// (sporco/pgm/pgm.py:803, :815-831; sporco/pgm/cbpdn.py:314-345)
// ---------------------------------------------------------------------------
// PLAIN: forward transform only (no momentum, no sums): t <- FFT_H(t)
// BT: with STATS, also the linear term of the backtracking model from a.ey
template <int NW, int LP, int KC, bool STATS, bool PLAIN = false, bool BT = false, bool PERS = false, int N1 = 32>
__global__ void __launch_bounds__(NW * 64) pgm_fft_momentum_kernel(const PgmColsArgs<float> a) {
    constexpr bool MR = mr_length(N1);
    constexpr int H = N1 * NW, J = MR ? (N1 > NW ? 2 : 1) : N1 / NW;
    static_assert(!MR || (NW == 16 && LP == 1), "mixed-radix heights: 16 waves, one line per group");
    constexpr int LBW = ilog2(NW);
    constexpr int FP = LP * NW, Q = J / LP, CPL = NW / 4, NCH = LP * CPL;
    // (persistent as pgm_grad_ifft: 16 waves, K = 64 -- there is no slab axis then)
    constexpr bool PERSIST = PERS;
    static_assert(!PERS || (NW == 16 && KC == 64), "persistent form: 16 waves, K = 64");
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int K = KC ? KC : a.K;
    // (K > 64, run-time K only: one workgroup per (tile, 64-filter slab), slab = blockIdx.y)
    const int slab = KC ? 0 : (int)blockIdx.y, NHs = KC ? 1 : (int)gridDim.y;
    const bool kv = KC == 64 ? true : slab * 64 + k < K;
    const int xcd = blockIdx.x & 7;
    const int ko = (w * K + slab * 64 + k) * (int)sizeof(cf);
    f2 *L = dyn_lds<f2>();
    double *scratch = reinterpret_cast<double *>(L + FP * NW * 64);
    const cf zero = mk<float>(0.f, 0.f);
    int token = 0;
    if constexpr (PERSIST) {
        const int ph = (int)(blockIdx.x >> 3) % a.stagger_groups;
        for (int i = 0; i < ph * a.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (int slot = blockIdx.x >> 3;; slot += gridDim.x >> 3) {
    SA_ARGS_PTR_T(PgmColsArgs<float>) ap = sa_args_reload<PERSIST>(a);
    const int Wf = ap->W / 2 + 1, CN = ap->CN;
    if (slot >= ((Wf + 7) / 8) * CN) break;
    const int wf = (slot / CN) * 8 + xcd;
    if (wf >= Wf) break;
    const int tile = wf * CN + slot % CN;
    const uint32_t tbytes = (uint32_t)(H * K * sizeof(cf));
    const BufRsrc Tb = make_rsrc(ap->t + (int64_t)tile * H * K, tbytes);
    const BufRsrc Xo = PLAIN ? Tb : make_rsrc(ap->xf_old + (int64_t)tile * H * K, tbytes);
    const BufRsrc Yo = PLAIN ? Tb : make_rsrc(ap->yf + (int64_t)tile * H * K, tbytes);
    // (a null yf_new -- a rule that forms Yf itself, pgm_iter hold == 2 -- makes the descriptor empty:
    // the stores are out of range and cost no traffic)
    const BufRsrc Yn = PLAIN ? Tb
                             : make_rsrc(ap->yf_new ? ap->yf_new + (int64_t)tile * H * K : nullptr,
                                         ap->yf_new ? tbytes : 0u);
    const BufRsrc Db = make_rsrc(ap->dft + (int64_t)wf * H * K, tbytes);
    const cf *S = ap->sft + (int64_t)tile * H + w;
    const cf *EY = BT ? ap->ey + (int64_t)tile * H + w : nullptr;
    cf *QP = (STATS && !KC && ap->qpart) ? ap->qpart + ((int64_t)tile * NHs + slab) * H + w : nullptr;
    const cf *twA = ap->twA + w * N1;
    const float beta = ap->beta;
    float rs = 0.f, fsum = 0.f, lin = 0.f;

    // rows h = NW h1 + w of T', forward FFT over h1, twiddle
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

    // group q of the exchange leaves the register tile: (wave = h2; f1 in registers) -> LDS
    auto write_group = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
#pragma unroll
        for (int fl = 0; fl < FP; ++fl) {
            if (q * FP + fl >= N1) continue;
            const cf x = v[pos1<N1>(q * FP + fl)];
            f2 t;
            t.x = x.re;
            t.y = x.im;
            L[(fl * NW + w) * 64 + k] = t;
        }
    };
    write_group(std::integral_constant<int, 0>{});
    static_for<Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        // previous iterates (and Df / Sf for the objective) of chunk g+1 are requested while
        // chunk g is processed; chunk 0's before the barrier
        const bool lv = !MR || q * FP + w < N1;      // (this wave's line of the group exists)
        cf xn4[4], yn4[4];
        auto prefetch = [&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                if constexpr (!PLAIN) {
                    xn4[e] = kv ? buf_load_cf(Xo, ko, fo * K * (int)sizeof(cf)) : zero;
                    yn4[e] = kv ? buf_load_cf(Yo, ko, fo * K * (int)sizeof(cf)) : zero;
                } else {
                    xn4[e] = zero;
                    yn4[e] = zero;
                }
            }
        };
        if (lv) prefetch(std::integral_constant<int, 0>{});
        __syncthreads();
        cf u[FP];
        if (lv) {
#pragma unroll
        for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
            for (int h2 = 0; h2 < NW; ++h2) {
                const f2 t = L[((w + NW * jl) * NW + h2) * 64 + k];
                u[NW * jl + h2] = mk<float>(t.x, t.y);
            }
        }
        }
        if constexpr (q + 1 < Q) {
            // the next group goes to the exchange buffer as soon as this one has been read:
            // its half of the register tile is then free while this group's chunks -- the
            // register-hungry part: momentum operands, Df rows, the wave reduction -- run
            __syncthreads();
            write_group(std::integral_constant<int, q + 1>{});
        }
        if (lv) {
        static_for<NCH>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int jl = g / CPL, c = g % CPL, j = q * LP + jl;
            if constexpr (c == 0) dif<NW, false>(u, NW * jl);   // u[NW jl + i] = Xf'[f]
            cf xo[4], yo[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xo[e] = xn4[e];
                yo[e] = yn4[e];
            }
            if constexpr (g + 1 < NCH) prefetch(std::integral_constant<int, g + 1>{});
            if constexpr (STATS) {
                // f(Xf') = 0.5 sum |sum_k Df Xf' - Sf|^2: Df (L2-resident) loaded at use
                cf dd[4], sv[4], qq[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                    dd[e] = kv ? buf_load_cf_cached(Db, ko, fo * K * (int)sizeof(cf)) : zero;
                    sa_uload2(reinterpret_cast<const float *>(S + fo), sv[e].re, sv[e].im);
                }
                inner4(dd, &u[NW * jl + 4 * c], k, qq);
                if (QP) {     // the slab's share only: the sums are formed by pgm_stats_slabs
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k == 0) QP[NW * j + N1 * brev(4 * c + e, LBW)] = qq[e];
                } else
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const cf ex = qq[e] - sv[e];
                    fsum = cabs2_add(fsum, ex);
                    if constexpr (BT) {
                        cf ey;
                        sa_uload2(reinterpret_cast<const float *>(EY + NW * j + N1 * brev(4 * c + e, LBW)),
                                  ey.re, ey.im);
                        lin = fma1(ex.re - ey.re, ey.re, fma1(ex.im - ey.im, ey.im, lin));
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fo = NW * j + N1 * brev(4 * c + e, LBW);
                const cf xn = u[NW * jl + 4 * c + e];
                const cf yn = mk<float>(fma1(xn.re - xo[e].re, beta, xn.re), fma1(xn.im - xo[e].im, beta, xn.im));
                if (kv) {
                    buf_store_cf(Tb, ko, fo * K * (int)sizeof(cf), xn);
                    if constexpr (!PLAIN) {
                        buf_store_cf(Yn, ko, fo * K * (int)sizeof(cf), yn);
                        rs = cabs2_add(rs, xn - yo[e]);
                    }
                }
            }
        });
        }   // lv
        {
            float &rs_ = rs, &fs_ = fsum;   // (named references: asm operands alone do not capture)
            int &tk_ = token;
            SA_VGPR_FENCE3(rs_, fs_, tk_);
        }
    });

    if constexpr (PLAIN) {
        if constexpr (PERSIST) __syncthreads();     // the exchange buffer is reused by the next tile
    } else {
        const double pw = (wf == 0 || ((ap->W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
        const double rsw = wave_sum((double)rs);   // rs is per lane (all filters); fsum is wave-uniform
        double acc[kPgmPartialStride] = {(k == 0 ? rsw : 0.0) * pw, k == 0 ? (double)fsum * pw : 0.0,
                                         k == 0 ? (double)fsum : 0.0, k == 0 ? (double)lin : 0.0,
                                         k == 0 ? rsw : 0.0, 0.0};
        block_sum_store<kPgmPartialStride>(acc, scratch,
                                           ap->partials + ((int64_t)tile * NHs + slab) * kPgmPartialStride);
    }
    if constexpr (!PERSIST) break;
    }   // persistent loop over this workgroup's tiles
}

// K > 64: the objective sums from the slabs' shares of sum_k Df Xf' (see csc_pgm.h)
__global__ void __launch_bounds__(256) pgm_stats_slabs_kernel(const PgmColsArgs<float> a, int NH,
                                                              double *partials2) {
    const int tile = blockIdx.x, Wf = a.W / 2 + 1, wf = tile / a.CN;
    const cf *qp = a.qpart + (int64_t)tile * NH * a.H;
    const cf *S = a.sft + (int64_t)tile * a.H;
    const cf *EY = a.ey ? a.ey + (int64_t)tile * a.H : nullptr;
    double fs = 0.0, lin = 0.0;
    for (int f = threadIdx.x; f < a.H; f += blockDim.x) {
        cf ex = mk<float>(0.f, 0.f);
        for (int sl = 0; sl < NH; ++sl) ex = ex + qp[(int64_t)sl * a.H + f];
        ex = ex - S[f];
        fs += (double)cabs2(ex);
        if (EY) {
            const cf ey = EY[f];
            lin += (double)((ex.re - ey.re) * ey.re + (ex.im - ey.im) * ey.im);
        }
    }
    const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
    double acc[3] = {fs * pw, fs, lin};
    block_sum_store<3>(acc, dyn_lds<double>(), partials2 + (int64_t)tile * 3);
}

// ---------------------------------------------------------------------------
// dictionary-update gradient on tile-major coefficient spectra (see csc_pgm.h)
// ---------------------------------------------------------------------------
// MODE 0: everything in one pass (K <= 64).  K > 64, one workgroup per 64-filter slab
// (blockIdx.y): MODE 1 writes the slab's share of sum_k zf d to a.qpart; MODE 2 forms the
// gradient from the residual in a.rbuf (ccmod_resid_sum_kernel in between).
// N1: rows per thread (H = N1 NW); a length that is not a multiple of four runs a last chunk whose
// surplus rows read zeros (buffer range checks) and are not stored (MODE 0 only).
template <int NW, int KC, int MODE = 0, int N1 = 32>
__global__ void __launch_bounds__(NW * 64) ccmod_grad_tiled_kernel(const CcmodTiledArgs<float> a) {
    constexpr int H = N1 * NW, NC = (N1 + 3) / 4, NR = 4 * NC;
    static_assert(N1 % 4 == 0 || MODE == 0, "ragged heights: the one-pass form only");
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int K = KC ? KC : a.K;
    const int slab = MODE ? (int)blockIdx.y : 0, NHs = MODE ? (int)gridDim.y : 1;
    const bool kv = KC == 64 ? true : slab * 64 + k < K;
    const int Wf = a.W / 2 + 1;
    const int g = blockIdx.x % a.G, wf = blockIdx.x / a.G;
    const int cng = (a.CN + a.G - 1) / a.G;
    const int cn0 = g * cng, cn1 = (cn0 + cng < a.CN) ? cn0 + cng : a.CN;
    const cf zero = mk<float>(0.f, 0.f);
    double *scratch = dyn_lds<double>();
    // rows f = NW i + w of this thread; d(f, wf, k) is re-read per tile (L2-resident)
    const uint32_t dbytes = (uint32_t)((int64_t)H * Wf * K * sizeof(cf));
    const BufRsrc Db = make_rsrc(a.d, dbytes);
    const int dko = ((w * Wf + wf) * K + slab * 64 + k) * (int)sizeof(cf);
    const int drow = NW * Wf * K * (int)sizeof(cf);   // from row f to row f + NW
    const int ko = (w * K + slab * 64 + k) * (int)sizeof(cf);
    cf acc[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) acc[i] = zero;
    float s_r2 = 0.f, s_q2 = 0.f;
    int kov = ko, dkov = dko, token = 0;   // offsets routed through the register fences below
    for (int cn = cn0; cn < cn1; ++cn) {
        const int tile = wf * a.CN + cn;
        const BufRsrc Zb = make_rsrc(a.zf + (int64_t)tile * H * K, (uint32_t)(H * K * sizeof(cf)));
        const cf *S = a.sft + (int64_t)tile * H + w;
        // 4 rows at a time; the next 4 rows of Zf and d are in flight meanwhile.  The
        // offsets of prefetch c+1 are tied (empty asm) to a result of chunk c-1, which
        // keeps the compiler from hoisting all 64 loads above the arithmetic.
        cf zn[4], dn[4];
        auto prefetch = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * c + e;
                zn[e] = kv ? buf_load_cf(Zb, kov, NW * i * K * (int)sizeof(cf)) : zero;
                dn[e] = (MODE != 2 && kv) ? buf_load_cf_cached(Db, dkov, i * drow) : zero;
            }
        };
        prefetch(std::integral_constant<int, 0>{});
        static_for<NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            cf z[4], d[4], q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                z[e] = zn[e];
                d[e] = dn[e];
            }
            if constexpr (c + 1 < NC) {
                if constexpr (c > 0) {
                    float &dep = acc[4 * c - 1].re;
                    int &ko_ = kov, &dko_ = dkov;
                    SA_VGPR_FENCE3(dep, ko_, dko_);
                }
                prefetch(std::integral_constant<int, c + 1>{});
            }
            if constexpr (MODE == 1) {
                inner4(d, z, k, q);
                cf *qp = a.qpart + ((int64_t)tile * NHs + slab) * H + w;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k == 0) qp[NW * (4 * c + e)] = q[e];
            } else if constexpr (MODE == 2) {
                const cf *R = a.rbuf + (int64_t)tile * H + w;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * c + e;
                    cf r;
                    sa_uload2(reinterpret_cast<const float *>(R + NW * i), r.re, r.im);
                    acc[i] = cmulc_add(acc[i], z[e], r);
                }
            } else {
            inner4(d, z, k, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * c + e;
                if (i >= N1) continue;       // (the surplus rows of a ragged last chunk)
                cf sv;
                sa_uload2(reinterpret_cast<const float *>(S + NW * i), sv.re, sv.im);
                const cf r = q[e] - sv;
                s_r2 = cabs2_add(s_r2, r);
                s_q2 = cabs2_add(s_q2, q[e]);
                acc[i] = cmulc_add(acc[i], z[e], r);
            }
            }
        });
        {
            float &dep = acc[N1 - 1].re;
            SA_VGPR_FENCE3(dep, kov, token);
        }
    }
    if (MODE != 1 && a.gpart && kv) {
        cf *gp = a.gpart + (int64_t)g * H * Wf * K;
#pragma unroll
        for (int i = 0; i < N1; ++i) gp[((int64_t)(NW * i + w) * Wf + wf) * K + slab * 64 + k] = acc[i];
    }
    if constexpr (MODE == 0) {
        const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
        double accd[4] = {k == 0 ? (double)s_r2 : 0.0, k == 0 ? (double)s_r2 * pw : 0.0,
                          k == 0 ? (double)s_q2 : 0.0, 0.0};
        block_sum_store<4>(accd, scratch, a.partials + (int64_t)blockIdx.x * 4);
    }
}

// The same for K = 64 and H = 16 NW <= 256, rebuilt around what bounds the kernel above at those
// sizes -- bytes in flight: there, a wave has 4 rows of Zf requested while it works on 4 others,
// 16 waves per CU, 32 KB; a CU needs about twice that to keep its share of the HBM pipe full.
// Here one workgroup of 16 waves per CU (16 rows per thread instead of 32: half the accumulator
// registers) keeps a whole tile -- 16 rows per thread, 128 KiB per workgroup -- requested ahead: the
// four 4-row chunks of image n + 1 are asked for as the chunks of image n are consumed, the signal
// coefficients with them.  The dictionary column d(., wf, .), which does not change over the images
// of a workgroup, waits in LDS instead of being re-read from L2 for every image (each thread reads
// back what it stored: no barrier).  Gradient bits as above (same association per element); the
// two sums are added in another order.
template <int NW>
__global__ void __launch_bounds__(NW * 64) ccmod_grad_tiled_ahead_kernel(const CcmodTiledArgs<float> a) {
    constexpr int N1 = 16, H = N1 * NW, NC = N1 / 4, K = 64;
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int Wf = a.W / 2 + 1;
    const int g = blockIdx.x % a.G, wf = blockIdx.x / a.G;
    const int cng = (a.CN + a.G - 1) / a.G;
    const int cn0 = g * cng, cn1 = (cn0 + cng < a.CN) ? cn0 + cng : a.CN;
    const cf zero = mk<float>(0.f, 0.f);
    cf *dl = dyn_lds<cf>();                                       // [H][K]
    double *scratch = reinterpret_cast<double *>(dl + H * K);
    {
        const BufRsrc Db = make_rsrc(a.d, (uint32_t)((int64_t)H * Wf * K * sizeof(cf)));
        const int dko = ((w * Wf + wf) * K + k) * (int)sizeof(cf);
#pragma unroll
        for (int i = 0; i < N1; ++i)
            dl[(NW * i + w) * K + k] = buf_load_cf_cached(Db, dko, i * NW * Wf * K * (int)sizeof(cf));
    }
    cf acc[N1];
#pragma unroll
    for (int i = 0; i < N1; ++i) acc[i] = zero;
    float s_r2 = 0.f, s_q2 = 0.f;
    cf zb[NC][4], sb[NC];      // sb[c]: lane l holds the signal coefficient of row 4 c + (l & 3)
    // (the offsets pass through the register fences below: a request may not be scheduled before
    // the chunk whose registers it reuses has been consumed, nor the LDS reads of all 16 rows of d
    // ahead of the first chunk -- either would double the registers of the tile)
    int ko = (w * K + k) * (int)sizeof(cf);
    int so = (NW * (k & 3) + w) * (int)sizeof(cf);
    int dlo = w * K + k;
    // (live = false: a buffer of no bytes -- the loads return zero without touching memory; the
    // request after the last image, which keeps the loop free of a branch around the loads)
    auto request = [&](auto cc, int cn, bool live) {
        constexpr int c = decltype(cc)::value;
        const int tile = wf * a.CN + cn;
        const BufRsrc Zb = make_rsrc(a.zf + (int64_t)tile * H * K, live ? (uint32_t)(H * K * sizeof(cf)) : 0u);
        const BufRsrc Sb = make_rsrc(a.sft + (int64_t)tile * H, live ? (uint32_t)(H * sizeof(cf)) : 0u);
#pragma unroll
        for (int e = 0; e < 4; ++e) zb[c][e] = buf_load_cf(Zb, ko, NW * (4 * c + e) * K * (int)sizeof(cf));
        sb[c] = buf_load_cf_cached(Sb, so, NW * 4 * c * (int)sizeof(cf));
    };
    static_for<NC>([&](auto cc) { request(cc, cn0 < cn1 ? cn0 : 0, cn0 < cn1); });
    for (int cn = cn0; cn < cn1; ++cn) {
        const bool more = cn + 1 < cn1;
        static_for<NC>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            cf d[4], q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = dl[dlo + NW * (4 * c + e) * K];
            inner4(d, zb[c], k, q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * c + e;
                const cf r = q[e] - mk<float>(sa_readlane(sb[c].re, e), sa_readlane(sb[c].im, e));
                s_r2 = cabs2_add(s_r2, r);
                s_q2 = cabs2_add(s_q2, q[e]);
                acc[i] = cmulc_add(acc[i], zb[c][e], r);
            }
            {
                float &dep = acc[4 * c + 3].re;
                SA_VGPR_FENCE3(dep, ko, so);
                SA_VGPR_FENCE3(dep, dlo, dlo);
            }
            request(cc, more ? cn + 1 : cn, more);
        });
    }
    if (a.gpart) {
        cf *gp = a.gpart + (int64_t)g * H * Wf * K;
#pragma unroll
        for (int i = 0; i < N1; ++i) gp[((int64_t)(NW * i + w) * Wf + wf) * K + k] = acc[i];
    }
    const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
    double accd[4] = {k == 0 ? (double)s_r2 : 0.0, k == 0 ? (double)s_r2 * pw : 0.0,
                      k == 0 ? (double)s_q2 : 0.0, 0.0};
    block_sum_store<4>(accd, scratch, a.partials + (int64_t)blockIdx.x * 4);
}
template <int NW> static void launch_ccmod_ahead(hipStream_t st, const CcmodTiledArgs<float> &a, unsigned grid) {
    const size_t lds = sizeof(cf) * 16 * NW * 64 + sizeof(double) * 4 * 16;
    static PerDeviceOnce attr_set;
    if (attr_set.first())
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ccmod_grad_tiled_ahead_kernel<NW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((ccmod_grad_tiled_ahead_kernel<NW>), dim3(grid), dim3(NW * 64), lds, st, a);
}

// K > 64, between the two passes: r = sum_slab qpart - sf per frequency of a tile, and the sums
// of the one-pass kernel per tile
__global__ void __launch_bounds__(256) ccmod_resid_sum_kernel(const CcmodTiledArgs<float> a, int NH) {
    const int tile = blockIdx.x, Wf = a.W / 2 + 1, wf = tile / a.CN;
    const cf *qp = a.qpart + (int64_t)tile * NH * a.H;
    const cf *S = a.sft + (int64_t)tile * a.H;
    cf *R = a.rbuf + (int64_t)tile * a.H;
    double r2 = 0.0, q2 = 0.0;
    for (int f = threadIdx.x; f < a.H; f += blockDim.x) {
        cf q = mk<float>(0.f, 0.f);
        for (int sl = 0; sl < NH; ++sl) q = q + qp[(int64_t)sl * a.H + f];
        const cf r = q - S[f];
        R[f] = r;
        r2 += (double)cabs2(r);
        q2 += (double)cabs2(q);
    }
    const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
    double accd[4] = {r2, r2 * pw, q2, 0.0};
    block_sum_store<4>(accd, dyn_lds<double>(), a.partials + (int64_t)tile * 4);
}

__global__ void __launch_bounds__(256) sum_groups_kernel(const cf *__restrict__ part,
                                                         cf *__restrict__ out, int64_t n, int G) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        cf s = part[i];
        for (int g = 1; g < G; ++g) s = s + part[(int64_t)g * n + i];
        out[i] = s;
    }
}

template <typename F> void set_lds(F kernel, size_t bytes) {
    SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// Workgroups of a persistent launch (16 waves, K = 64): one per CU, a multiple of 8 (the XCD
// of a workgroup is blockIdx % 8 for every slot it walks); 0 = one workgroup per tile.
// SPORCO_AMD_PGM_PERSIST: bit 0 the gradient kernel, bit 1 the momentum kernel.  Default 1:
// measured at 512 x 512, K = 64, N = 32 (profiles/r03a_config4.jsonl) 253 it/s with the
// gradient kernel persistent, 247 with both, 246 with neither, 239 with the momentum kernel
// alone -- its tile loop costs it 20 more registers (scalar registers run out and spill into
// vector ones; the statistics variant then spills 60 bytes), which outweighs what the loop buys.
// Stagger as in csc_fused.hip.
static unsigned pgm_persist_grid(PgmColsArgs<float> &a, int NW, int KC, int which) {
    constexpr int sg = 4, ss = 2, mask = 1;   // (stagger as csc_fused.h kColsStagger*; mask: the gradient kernel only)
    const int cus = current_device_cus();
    const int64_t all = ceil_div(a.W / 2 + 1, 8) * 8 * a.CN;
    a.stagger_groups = sg;
    a.stagger_sleeps = ss;
    const int64_t g = std::max(8, cus / 8 * 8);
    if (!(mask & which) || NW != 16 || KC != 64 || all <= g) return 0u;
    return (unsigned)g;
}
static unsigned pgm_all_tiles(const PgmColsArgs<float> &a) {
    return (unsigned)(ceil_div(a.W / 2 + 1, 8) * 8 * a.CN);
}

template <int NW, int LP, int KC, bool BT, bool PERS, bool EYIN = false>
void launch_grad_inst(hipStream_t st, const PgmColsArgs<float> &a, unsigned grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds(&pgm_grad_ifft_kernel<NW, LP, KC, BT, PERS, EYIN>, pgm_lds_bytes(NW, LP));
    }
    hipLaunchKernelGGL((pgm_grad_ifft_kernel<NW, LP, KC, BT, PERS, EYIN>), dim3(grid), dim3(NW * 64),
                       pgm_lds_bytes(NW, LP), st, a);
}
template <int NW, int LP, int KC> void launch_grad(hipStream_t st, const PgmColsArgs<float> &a_in) {
    PgmColsArgs<float> a = a_in;
    const unsigned pg = pgm_persist_grid(a, NW, KC, 1);
    if (a.ey_in) {
        SA_REQUIRE(!a.ey, "a residual from memory does not combine with a held trial");
        if constexpr (NW == 16 && KC == 64) {
            if (pg) return launch_grad_inst<NW, LP, KC, false, true, true>(st, a, pg);
        }
        return launch_grad_inst<NW, LP, KC, false, false, true>(st, a, pgm_all_tiles(a));
    }
    if constexpr (NW == 16 && KC == 64) {
        if (pg) {
            if (a.ey) launch_grad_inst<NW, LP, KC, true, true>(st, a, pg);
            else launch_grad_inst<NW, LP, KC, false, true>(st, a, pg);
            return;
        }
    }
    if (a.ey) launch_grad_inst<NW, LP, KC, true, false>(st, a, pgm_all_tiles(a));
    else launch_grad_inst<NW, LP, KC, false, false>(st, a, pgm_all_tiles(a));
}

template <int NW, int LP, int KC, bool STATS, bool PLAIN, bool BT, bool PERS>
void launch_mom_inst(hipStream_t st, const PgmColsArgs<float> &a, unsigned grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds(&pgm_fft_momentum_kernel<NW, LP, KC, STATS, PLAIN, BT, PERS>, pgm_lds_bytes(NW, LP));
    }
    const unsigned slabs = KC ? 1u : (unsigned)ceil_div(a.K, 64);
    hipLaunchKernelGGL((pgm_fft_momentum_kernel<NW, LP, KC, STATS, PLAIN, BT, PERS>), dim3(grid, slabs),
                       dim3(NW * 64), pgm_lds_bytes(NW, LP), st, a);
}
template <int NW, int LP, int KC, bool STATS, bool BT = false, bool PLAIN = false>
void launch_mom(hipStream_t st, const PgmColsArgs<float> &a_in) {
    PgmColsArgs<float> a = a_in;
    const unsigned pg = pgm_persist_grid(a, NW, KC, 2);
    if constexpr (NW == 16 && KC == 64) {
        if (pg) return launch_mom_inst<NW, LP, KC, STATS, PLAIN, BT, true>(st, a, pg);
    }
    launch_mom_inst<NW, LP, KC, STATS, PLAIN, BT, false>(st, a, pgm_all_tiles(a));
}

}  // namespace

#ifndef SA_PGM_MR_TU
template <> int64_t launch_pgm_grad_ifft<float>(hipStream_t st, const PgmColsArgs<float> &a) {
    if (fused_mr_height(a.H)) return launch_pgm_grad_ifft_mr(st, a);
    SA_REQUIRE((a.H == 128 || a.H == 256 || a.H == 512) && a.K >= 1 && a.K <= 64,
               "shape not handled by the fused PGM kernels");
    if (a.H == 128) {
        if (a.K == 64) launch_grad<4, 4, 64>(st, a);
        else launch_grad<4, 4, 0>(st, a);
    } else if (a.H == 256) {
        if (a.K == 64) launch_grad<8, 2, 64>(st, a);
        else launch_grad<8, 2, 0>(st, a);
    } else {
        if (a.K == 64) launch_grad<16, 1, 64>(st, a);
        else launch_grad<16, 1, 0>(st, a);
    }
    SA_HIP(hipGetLastError());
    return (int64_t)(a.W / 2 + 1) * a.CN;
}

template <> int64_t launch_pgm_stats_slabs<float>(hipStream_t st, const PgmColsArgs<float> &a,
                                                  double *partials2) {
    const int64_t ntiles = (int64_t)(a.W / 2 + 1) * a.CN;
    hipLaunchKernelGGL(pgm_stats_slabs_kernel, dim3((unsigned)ntiles), dim3(256), sizeof(double) * 3 * 4,
                       st, a, (int)ceil_div(a.K, 64), partials2);
    SA_HIP(hipGetLastError());
    return ntiles;
}
template <> int64_t launch_pgm_stats_slabs<double>(hipStream_t, const PgmColsArgs<double> &, double *) {
    throw Error(-1, "the fused FISTA kernels are float32 only");
}
template <> int64_t launch_pgm_fft_momentum<float>(hipStream_t st, const PgmColsArgs<float> &a) {
    if (fused_mr_height(a.H)) return launch_pgm_fft_momentum_mr(st, a, false);
    SA_REQUIRE((a.H == 128 || a.H == 256 || a.H == 512) && a.K >= 1 && a.K <= 256,
               "shape not handled by the fused PGM kernels");
    SA_REQUIRE(a.K <= 64 || !a.want_stats || a.qpart, "K > 64 with statistics needs qpart");
    const bool k64 = a.K == 64, st8 = a.H == 256, stats = a.want_stats != 0;
    if (a.H == 128) {
        if (a.ey) { if (k64) launch_mom<4, 4, 64, true, true>(st, a); else launch_mom<4, 4, 0, true, true>(st, a); }
        else if (k64) { if (stats) launch_mom<4, 4, 64, true>(st, a); else launch_mom<4, 4, 64, false>(st, a); }
        else { if (stats) launch_mom<4, 4, 0, true>(st, a); else launch_mom<4, 4, 0, false>(st, a); }
    } else if (a.ey) {      // a backtracking trial
        SA_REQUIRE(stats, "the backtracking sums need want_stats");
        if (st8) { if (k64) launch_mom<8, 2, 64, true, true>(st, a); else launch_mom<8, 2, 0, true, true>(st, a); }
        else { if (k64) launch_mom<16, 1, 64, true, true>(st, a); else launch_mom<16, 1, 0, true, true>(st, a); }
    } else if (st8) {
        if (k64) { if (stats) launch_mom<8, 2, 64, true>(st, a); else launch_mom<8, 2, 64, false>(st, a); }
        else { if (stats) launch_mom<8, 2, 0, true>(st, a); else launch_mom<8, 2, 0, false>(st, a); }
    } else {
        if (k64) { if (stats) launch_mom<16, 1, 64, true>(st, a); else launch_mom<16, 1, 64, false>(st, a); }
        else { if (stats) launch_mom<16, 1, 0, true>(st, a); else launch_mom<16, 1, 0, false>(st, a); }
    }
    SA_HIP(hipGetLastError());
    return (int64_t)(a.W / 2 + 1) * a.CN * ceil_div(a.K, 64);
}

template <int NW, int LP, int KC> static void launch_plain(hipStream_t st, const PgmColsArgs<float> &a) {
    launch_mom<NW, LP, KC, false, false, true>(st, a);
}

template <> int64_t launch_cols_fft<float>(hipStream_t st, const PgmColsArgs<float> &a) {
    if (fused_mr_height(a.H)) return launch_pgm_fft_momentum_mr(st, a, true);
    SA_REQUIRE((a.H == 128 || a.H == 256 || a.H == 512) && a.K >= 1 && a.K <= 256,
               "shape not handled by the fused column kernels");
    if (a.H == 128) {
        if (a.K == 64) launch_plain<4, 4, 64>(st, a);
        else launch_plain<4, 4, 0>(st, a);
    } else if (a.H == 256) {
        if (a.K == 64) launch_plain<8, 2, 64>(st, a);
        else launch_plain<8, 2, 0>(st, a);
    } else {
        if (a.K == 64) launch_plain<16, 1, 64>(st, a);
        else launch_plain<16, 1, 0>(st, a);
    }
    SA_HIP(hipGetLastError());
    return (int64_t)(a.W / 2 + 1) * a.CN;
}
template <> int64_t launch_cols_fft<double>(hipStream_t, const PgmColsArgs<double> &) {
    throw Error(-1, "the fused column kernels are float32 only");
}

template <> bool ccmod_tiled_supported<float>(int H, int K) {
    return ((H == 128 || H == 256 || H == 512) && K >= 1 && K <= 256) || (fused_mr_height(H) && K >= 1 && K <= 64);
}
template <> bool ccmod_tiled_supported<double>(int, int) { return false; }

template <> int64_t launch_ccmod_grad_tiled<float>(hipStream_t st, const CcmodTiledArgs<float> &a) {
    SA_REQUIRE(ccmod_tiled_supported<float>(a.H, a.K), "shape not handled by the tiled D-step kernel");
    if (fused_mr_height(a.H)) return launch_ccmod_grad_tiled_mr(st, a);
    const unsigned grid = (unsigned)((a.W / 2 + 1) * a.G);
    const size_t lds = sizeof(double) * 4 * 16;
    if (a.K > 64) {
        // two passes over zf: the slabs' shares of sum_k zf d, the residual per frequency, then
        // (when a gradient is wanted) conj(zf) r per slab
        SA_REQUIRE(a.qpart && a.rbuf, "the K > 64 tiled D-step needs its exchange buffers");
        const dim3 g2(grid, (unsigned)ceil_div(a.K, 64));
        const int64_t ntiles = (int64_t)(a.W / 2 + 1) * a.CN;
        if (a.H == 128) hipLaunchKernelGGL((ccmod_grad_tiled_kernel<4, 0, 1>), g2, dim3(256), lds, st, a);
        else if (a.H == 256) hipLaunchKernelGGL((ccmod_grad_tiled_kernel<8, 0, 1>), g2, dim3(512), lds, st, a);
        else hipLaunchKernelGGL((ccmod_grad_tiled_kernel<16, 0, 1>), g2, dim3(1024), lds, st, a);
        hipLaunchKernelGGL(ccmod_resid_sum_kernel, dim3((unsigned)ntiles), dim3(256), lds, st, a,
                           (int)g2.y);
        if (a.gpart) {
            if (a.H == 128) hipLaunchKernelGGL((ccmod_grad_tiled_kernel<4, 0, 2>), g2, dim3(256), lds, st, a);
            else if (a.H == 256) hipLaunchKernelGGL((ccmod_grad_tiled_kernel<8, 0, 2>), g2, dim3(512), lds, st, a);
            else hipLaunchKernelGGL((ccmod_grad_tiled_kernel<16, 0, 2>), g2, dim3(1024), lds, st, a);
        }
        SA_HIP(hipGetLastError());
        return ntiles;
    }
    if (a.H == 128) {
        if (a.K == 64) launch_ccmod_ahead<8>(st, a, grid);
        else hipLaunchKernelGGL((ccmod_grad_tiled_kernel<4, 0>), dim3(grid), dim3(256), lds, st, a);
    } else if (a.H == 256) {
        if (a.K == 64) launch_ccmod_ahead<16>(st, a, grid);
        else hipLaunchKernelGGL((ccmod_grad_tiled_kernel<8, 0>), dim3(grid), dim3(512), lds, st, a);
    } else {
        if (a.K == 64) hipLaunchKernelGGL((ccmod_grad_tiled_kernel<16, 64>), dim3(grid), dim3(1024), lds, st, a);
        else hipLaunchKernelGGL((ccmod_grad_tiled_kernel<16, 0>), dim3(grid), dim3(1024), lds, st, a);
    }
    SA_HIP(hipGetLastError());
    return grid;
}
template <> int64_t launch_ccmod_grad_tiled<double>(hipStream_t, const CcmodTiledArgs<double> &) {
    throw Error(-1, "the tiled D-step kernel is float32 only");
}

template <>
void launch_sum_groups<float>(hipStream_t st, const cx<float> *part, cx<float> *out, int64_t n, int G) {
    int64_t grid = ceil_div(n, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(sum_groups_kernel, dim3((unsigned)grid), dim3(256), 0, st, part, out, n, G);
    SA_HIP(hipGetLastError());
}
template <>
void launch_sum_groups<double>(hipStream_t, const cx<double> *, cx<double> *, int64_t, int) {
    throw Error(-1, "float32 only");
}

template <> int64_t launch_pgm_grad_ifft<double>(hipStream_t, const PgmColsArgs<double> &) {
    throw Error(-1, "the fused PGM kernels are float32 only");
}
template <> int64_t launch_pgm_fft_momentum<double>(hipStream_t, const PgmColsArgs<double> &) {
    throw Error(-1, "the fused PGM kernels are float32 only");
}
#endif   // SA_PGM_MR_TU

}  // namespace sporco_amd
