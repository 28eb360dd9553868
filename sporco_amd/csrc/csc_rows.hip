// csc_rows.hip -- register-resident row transforms of the fused ADMM iteration
// (see csc_rows.h for what they fuse and the reference lines they replace).
//
// Layout of the work.  A workgroup of NW waves owns one image row h and 128
// consecutive columns p = (c, n, k) of it: lane l holds the filter pair
// (p, p+1) = 128*tile + 2l packed into one complex line z = x_p + i x_{p+1}, so
// every access to Y / U / X is a 512-byte row of float2 and every access to the
// tile-major spectrum T[wf][cn][h][k] is 16 bytes per lane.  The length-W
// transform is split W = 32 x NW as in csc_fused.hip:
//   spatial side   wave w holds the pixels x = NW*n1 + w, n1 = 0..31;
//   spectral side  wave w holds whole lines k1 of the intermediate
//                  C[k1][n2] (k1 = 0..31: index of the 32-point transform,
//                  n2 = 0..NW-1), namely the lines {w, 32-w} (and {16-w, 16+w}
//                  when NW = 8).  Those sets are closed under k1 -> -k1, which
//                  is what makes the real-transform "untangling"
//                      A[f] = (Z[f] + conj Z[W-f]) / 2,  B[f] = (Z[f] - conj Z[W-f]) / 2i
//                  (and its inverse) a purely per-thread operation: the bins f
//                  and W-f always live in the same thread.  Wave 0 owns the
//                  self-paired lines 0 and 16 (plus 8, 24 for NW = 8).
// One LDS exchange, in two halves of 16 lines (64 KiB each for W = 256, which
// lets two workgroups share a CU; 128 KiB for W = 512), moves the data between
// the two sides; nothing else touches LDS.
#include "csc_rows.h"

#include <cmath>
#include <cstdlib>

#include "csc_ctl_dev.h"
#include "csc_fused_body.h"
#include "regfft.h"

namespace sporco_amd {

namespace {

using namespace regfft;

struct alignas(16) cf2 {
    cf a, b;
};

constexpr int kN1 = 32;
constexpr size_t rows_lds_bytes(int NW) {
    return sizeof(f2) * 16 * NW * 64 + sizeof(double) * 8 * 16;
}

// line k1 held in slot j of spectral-side wave w (see the file header): slots (2 m, 2 m + 1) hold
// a pair {k1, 32 - k1} (wave 0, m = 0: the self-paired lines 0 and 16).  NW = 4 (W = 128) has
// eight slots per wave: the four of the other splits, then {4 + w, 28 - w} and {9 + w, 23 - w}.
// Mixed-radix lines (round 6): N1 = 10 ... 30 points per thread (regfft.h SA_MR_LENGTHS) with NW = 16
// waves (W = 16 N1 = 160 ... 480).  The spatial side is as before (wave w, pixels x = 16 n1 + w,
// n1 < N1); on the spectral side the N1 lines make pairs {k1, N1 - k1} -- wave 0 takes the
// self-paired lines 0 and (N1 even) N1 / 2, wave w >= 1 the pair {w, N1 - w} -- one pair per wave
// for the first NA = (N1 + 1) / 2 waves; the remaining 16 - NA waves idle through the second
// transform and the spectrum loads / stores (a part of a stage that is not what bounds the kernel)
// and take part in the exchange and its barriers only.
template <int N1, int NW> constexpr int spectral_waves() { return regfft::mr_length(N1) ? (N1 + 1) / 2 : NW; }
template <int N1, int NW> constexpr int spectral_lines() { return regfft::mr_length(N1) ? 2 : N1 / NW; }
// (odd N1: wave 0 has no second line)
template <int N1> __device__ __forceinline__ bool second_line(int w) { return (N1 & 1) == 0 || w != 0; }
template <int N1, int NW> __device__ __forceinline__ int line_of(int w, int j) {
    if (j == 0) return w;
    if (j == 1) return w == 0 ? N1 / 2 : N1 - w;
    if (j == 2) return w == 0 ? 8 : 16 - w;
    if (j == 3) return w == 0 ? 24 : 16 + w;
    if (j == 4) return 4 + w;
    if (j == 5) return 28 - w;
    if (j == 6) return 9 + w;
    return 23 - w;
}

// The exchange moves 16 lines at a time (32 KiB for NW = 4, 64 KiB for NW = 8, 128 KiB for
// NW = 16): group_of / kl_of give the half a line travels in and its slot there.  Each half
// holds whole {k1, 32 - k1} pairs: the first half of a wave's slots, or the second.
__host__ __device__ constexpr bool first_half4(int k1) {
    return k1 < 4 || k1 == 8 || (k1 >= 13 && k1 <= 19) || k1 == 24 || k1 > 28;
}
__host__ __device__ constexpr int group_of(int N1, int NW, int k1) {
    if (NW == 4) return first_half4(k1) ? 0 : 1;
    return NW == 16 ? (k1 >= (N1 + 1) / 2 ? 1 : 0) : ((k1 < 8 || k1 == 16 || k1 > 24) ? 0 : 1);
}
__host__ __device__ constexpr int kl_of(int N1, int NW, int k1) {
    if (NW == 4) {       // rank of the line among the 16 of its half
        int r = 0;
        for (int o = 0; o < k1; ++o) r += first_half4(o) == first_half4(k1) ? 1 : 0;
        return r;
    }
    if (NW == 16) return k1 >= (N1 + 1) / 2 ? k1 - (N1 + 1) / 2 : k1;
    if (group_of(N1, NW, k1) == 0) return k1 < 8 ? k1 : (k1 == 16 ? 8 : k1 - 16);
    return k1 < 16 ? k1 - 8 : k1 - 9;
}

__device__ __forceinline__ float soft1(float v, float thr) {
    // sign(v) * max(|v| - thr, 0)            (prox/_lp.py:181), for any threshold
    float m = fabsf(v) - thr;
    m = m > 0.f ? m : 0.f;
    return __builtin_copysignf(m, v);
}
// the same for a threshold known to be >= 0 (the scalar lambda / rho: the API layer sends
// negative lambdas to the generic chain): v minus v clamped to [-thr, thr], two instructions
__device__ __forceinline__ float soft1_pos(float v, float thr) { return v - sa_med3(v, -thr, thr); }
// MODE 1 carries a weight array, whose entries may have either sign
template <int MODE> __device__ __forceinline__ float soft1_m(float v, float thr) {
    if constexpr (MODE == 1) return soft1(v, thr);
    else return soft1_pos(v, thr);
}

// Spatial side -> spectral side: v[n1] = z(x = NW n1 + w) (destroyed) is transformed
// along W, untangled into the spectra of the two packed real lines, and the bins
// f <= W/2 are stored tile-major at t[f][cn][h][k..k+1].
// COH (here and below): the spectrum changes hands between workgroups of the same launch
// (admm_persist_kernel) -- agent-scope accesses instead of the streaming ones.
template <int NW, bool COH = false, int N1 = kN1>
__device__ __forceinline__ void spatial_to_spectral(cf (&v)[N1], const cf *twA, cf *t, int CN, int H,
                                                    int K, int cn, int k, int h, bool pv, int w,
                                                    int lane, f2 *L, int &token) {
    constexpr int J = spectral_lines<N1, NW>(), NA = spectral_waves<N1, NW>();
    constexpr int NG = 2;                    // exchange halves
    constexpr int LPG = J / NG;
    constexpr int LBW = ilog2(NW);
    const bool act = NA >= NW || w < NA;     // (wave-uniform; always true for the power-of-two lines)
    dif1<N1, false>(v, 0);
#pragma unroll
    for (int i = 1; i < N1; ++i) {
        cf tw;
        sa_uload2(reinterpret_cast<const float *>(twA + w * N1 + i), tw.re, tw.im);
        v[i] = cmul(v[i], tw);
    }
    reg_fence<N1>(v, 0, token);

    // ---- exchange to the spectral side: z[NW j + n2] = C[line_of<NW>(w, j)][n2] ------------
    cf z[J * NW];
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) {
            if (group_of(N1, NW, k1) != g) continue;
            const cf x = v[pos1<N1>(k1)];   // C[k1][n2 = w]
            f2 t;
            t.x = x.re;
            t.y = x.im;
            L[(kl_of(N1, NW, k1) * NW + w) * 64 + lane] = t;
        }
        __syncthreads();
        if (act && (g == 0 || second_line<N1>(w))) {
#pragma unroll
            for (int jl = 0; jl < LPG; ++jl) {
                const int j = g * LPG + jl;
                const int kl = kl_of(N1, NW, line_of<N1, NW>(w, j));
#pragma unroll
                for (int n2 = 0; n2 < NW; ++n2) {
                    const f2 t = L[(kl * NW + n2) * 64 + lane];
                    z[NW * j + n2] = mk<float>(t.x, t.y);
                }
            }
        } else if (act) {
#pragma unroll
            for (int i = 0; i < LPG * NW; ++i) z[NW * g * LPG + i] = mk<float>(0.f, 0.f);
        }
        if (g + 1 < NG) __syncthreads();
    });
    if (!act) return;

    // ---- transform over n2: z[NW j + i] = Z[k1 + N1 brev(i)] -----------------------------
#pragma unroll
    for (int j = 0; j < J; ++j) dif<NW, false>(z, NW * j);

    // ---- untangle the two real spectra and store the bins f <= W/2 ------------------------
    const int64_t tline = (int64_t)CN * H * K;
    cf *Tl = t + (int64_t)cn * H * K + (int64_t)h * K + k;
    auto store_unit = [&](int f, cf zf, cf zp) {
        // A = (Zf + conj Zp) / 2,  B = (Zf - conj Zp) / (2i)
        cf2 ab;
        ab.a = mk<float>(0.5f * (zf.re + zp.re), 0.5f * (zf.im - zp.im));
        ab.b = mk<float>(0.5f * (zf.im + zp.im), 0.5f * (zp.re - zf.re));
        if (pv) {
            const float q[4] = {ab.a.re, ab.a.im, ab.b.re, ab.b.im};
            if constexpr (COH) sa_coh_store4(reinterpret_cast<float *>(Tl + (int64_t)f * tline), q);
            else sa_stream_store4(reinterpret_cast<float *>(Tl + (int64_t)f * tline), q);
        }
    };
#pragma unroll
    for (int pr = 0; pr < J / 2; ++pr) {
        const int ja = 2 * pr, jb = 2 * pr + 1;
        const int k1a = line_of<N1, NW>(w, ja), k1b = line_of<N1, NW>(w, jb);
        if (pr == 0 && w == 0) {
            // self-paired lines 0 and N1 / 2: f and W - f sit in the same line
#pragma unroll
            for (int k2 = 0; k2 <= NW / 2; ++k2) {
                const cf zf = z[NW * ja + brev(k2 % NW, LBW)];
                const cf zp = z[NW * ja + brev((NW - k2) % NW, LBW)];
                store_unit(N1 * k2, zf, zp);
            }
            if constexpr ((N1 & 1) == 0) {
#pragma unroll
            for (int k2 = 0; k2 < NW / 2; ++k2) {
                const cf zf = z[NW * jb + brev(k2, LBW)];
                const cf zp = z[NW * jb + brev(NW - 1 - k2, LBW)];
                store_unit(N1 / 2 + N1 * k2, zf, zp);
            }
            }
        } else {
            // W - (k1a + N1 k2) = k1b + N1 (NW - 1 - k2)
#pragma unroll
            for (int k2 = 0; k2 < NW / 2; ++k2) {
                store_unit(k1a + N1 * k2, z[NW * ja + brev(k2, LBW)],
                           z[NW * jb + brev(NW - 1 - k2, LBW)]);
                store_unit(k1b + N1 * k2, z[NW * jb + brev(k2, LBW)],
                           z[NW * ja + brev(NW - 1 - k2, LBW)]);
            }
        }
    }
}

// Spectral side -> spatial side: the bins f <= W/2 of t[f][cn][h][k..k+1] are loaded,
// the packed spectrum Z is rebuilt, and v[n1] receives the unnormalised inverse
// transform at x = NW n1 + w: (re, im) = (filter k, filter k+1).
// (t_odd: the planes arrive in two buffers, csc_rows.h RowsPostArgs::t_odd)
// before_last: called when the exchange is over and only the in-register FFT-32 is left -- the
// point where the spectral-side registers have just died (the parked epilogue requests its first
// batch of the previous iterate there, under the transform).
struct NoHook {
    __device__ __forceinline__ void operator()() const {}
};
template <int NW, bool COH = false, int N1 = kN1, typename Hook = NoHook>
__device__ __forceinline__ void spectral_to_spatial(cf (&v)[N1], const cf *twW, const cf *t, int CN,
                                                    int H, int K, int cn, int k, int h, bool pv, int w,
                                                    int lane, f2 *L, int &token, const cf *t_odd = nullptr,
                                                    Hook before_last = Hook()) {
    constexpr int J = spectral_lines<N1, NW>(), NA = spectral_waves<N1, NW>();
    constexpr int NG = 2;
    constexpr int LPG = J / NG;
    constexpr int LBW = ilog2(NW);
    const bool act = NA >= NW || w < NA;     // (see spatial_to_spectral)
    const cf zero = mk<float>(0.f, 0.f);
    // ---- spectral side: load the bins f <= W/2 of this thread's lines, rebuild Z -------
    const int64_t tline = (int64_t)CN * H * K;
    const cf *Tl = t + (int64_t)cn * H * K + (int64_t)h * K + k;
    // striped planes: plane f at index f >> 1 of the buffer of its parity (wave-uniform arithmetic)
    const int64_t odd_delta = t_odd ? t_odd - t : 0;
    auto load_unit = [&](int f) {
        cf2 ab;
        ab.a = zero;
        ab.b = zero;
        if (pv) {
            float q[4];
            const int64_t poff = t_odd ? (int64_t)(f >> 1) * tline + ((f & 1) ? odd_delta : 0) : (int64_t)f * tline;
            if constexpr (COH) sa_coh_load4(reinterpret_cast<const float *>(Tl + poff), q);
            else sa_stream_load4(reinterpret_cast<const float *>(Tl + poff), q);
            ab.a = mk<float>(q[0], q[1]);
            ab.b = mk<float>(q[2], q[3]);
        }
        return ab;
    };
    cf z[J * NW];   // z[NW j + i] = Z[line_of<N1, NW>(w, j) + N1 brev(i)]
    if (act) {
#pragma unroll
    for (int pr = 0; pr < J / 2; ++pr) {
        const int ja = 2 * pr, jb = 2 * pr + 1;
        const int k1a = line_of<N1, NW>(w, ja), k1b = line_of<N1, NW>(w, jb);
        if (pr == 0 && w == 0) {
#pragma unroll
            for (int k2 = 0; k2 <= NW / 2; ++k2) {
                const cf2 ab = load_unit(N1 * k2);
                if (k2 == 0 || k2 == NW / 2) {
                    // DC / Nyquist: imaginary parts ignored, as numpy.fft.irfft does
                    z[NW * ja + brev(k2, LBW)] = mk<float>(ab.a.re, ab.b.re);
                } else {
                    z[NW * ja + brev(k2, LBW)] = mk<float>(ab.a.re - ab.b.im, ab.a.im + ab.b.re);
                    z[NW * ja + brev(NW - k2, LBW)] = mk<float>(ab.a.re + ab.b.im, ab.b.re - ab.a.im);
                }
            }
            if constexpr ((N1 & 1) == 0) {
#pragma unroll
            for (int k2 = 0; k2 < NW / 2; ++k2) {
                const cf2 ab = load_unit(N1 / 2 + N1 * k2);
                z[NW * jb + brev(k2, LBW)] = mk<float>(ab.a.re - ab.b.im, ab.a.im + ab.b.re);
                z[NW * jb + brev(NW - 1 - k2, LBW)] = mk<float>(ab.a.re + ab.b.im, ab.b.re - ab.a.im);
            }
            } else {
#pragma unroll
                for (int i = 0; i < NW; ++i) z[NW * jb + i] = zero;
            }
        } else {
#pragma unroll
            for (int k2 = 0; k2 < NW / 2; ++k2) {
                const cf2 ua = load_unit(k1a + N1 * k2);
                z[NW * ja + brev(k2, LBW)] = mk<float>(ua.a.re - ua.b.im, ua.a.im + ua.b.re);
                z[NW * jb + brev(NW - 1 - k2, LBW)] = mk<float>(ua.a.re + ua.b.im, ua.b.re - ua.a.im);
                const cf2 ub = load_unit(k1b + N1 * k2);
                z[NW * jb + brev(k2, LBW)] = mk<float>(ub.a.re - ub.b.im, ub.a.im + ub.b.re);
                z[NW * ja + brev(NW - 1 - k2, LBW)] = mk<float>(ub.a.re + ub.b.im, ub.b.re - ub.a.im);
            }
        }
    }
    reg_fence<J * NW>(z, 0, token);
    }

    // ---- inverse transform over k2, conj twiddle, exchange to the spatial side --------------
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if (act && (g == 0 || second_line<N1>(w))) {
#pragma unroll
        for (int jl = 0; jl < LPG; ++jl) {
            const int j = g * LPG + jl;
            const int k1 = line_of<N1, NW>(w, j);
            const int kl = kl_of(N1, NW, k1);
            dit<NW, true>(z, NW * j);
#pragma unroll
            for (int n2 = 0; n2 < NW; ++n2) {
                cf x = z[NW * j + n2];
                if (n2 > 0) {
                    // (n2 k1 < NW N1 = W: the table index needs no reduction)
                    cf tw;
                    sa_uload2(reinterpret_cast<const float *>(twW + n2 * k1), tw.re, tw.im);
                    x = cmulc(tw, x);
                }
                f2 t;
                t.x = x.re;
                t.y = x.im;
                L[(kl * NW + n2) * 64 + lane] = t;
            }
        }
        }
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) {
            if (group_of(N1, NW, k1) != g) continue;
            const f2 t = L[(kl_of(N1, NW, k1) * NW + w) * 64 + lane];
            v[pos1<N1>(k1)] = mk<float>(t.x, t.y);
        }
        if (g + 1 < NG) __syncthreads();
    });
    reg_fence<N1>(v, 0, token);
    before_last();
    dit1<N1, true>(v, 0);   // v[n1] = (X_p, X_{p+1}) at x = NW n1 + w, unnormalised
}

// ---------------------------------------------------------------------------
// rows_fwd: T = rfft_W(Y - s2 U), tile-major
// ---------------------------------------------------------------------------
// One tile (image row `h`, 128-column block `bx`) of rows_fwd.
// VFORM: the iterate arrives as V = AX + U of the iteration that produced it (csc_rows.h):
// Y = prox(V; thr_prev) (+ NonNeg), U = V - Y per element, then Y - s2 U as before.
// JOINT (with VFORM): Y = prox_sl1l2(V) couples the channels, so the tile is the joint epilogue's
// -- one image, 32 filters, all C <= 4 channels, lane = (channel, filter pair).
// MODE (with VFORM, as rows_inv_post): 1 = L1Weight array (+ NoBndryCross, AddMaskSim), 2 =
// NoBndryCross and / or AddMaskSim without a weight array -- the derivation of Y repeats them.
template <int NW, bool BCAST, bool VFORM, bool JOINT, int MODE, bool COH = false, int N1 = kN1, typename AP>
__device__ __forceinline__ void rows_fwd_tile(AP a, int bx, int h) {
    constexpr int W = N1 * NW;
    constexpr bool GENERAL = MODE != 0;
    static_assert(!(BCAST && VFORM), "the broadcast form reads a dictionary-sized Y");
    static_assert(!JOINT || VFORM, "only the V form needs the joint tiling");
    static_assert(!GENERAL || (VFORM && !JOINT), "options of the derivation: plain V form only");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    float s2 = a->s2, thr_p = a->thr_prev, thr21_p = a->thr21_prev;
    if (a->ctl) {       // device-driven solve
        s2 = a->ctl->u_scale_f;
        thr_p = a->ctl->thr_prev_f;
        thr21_p = a->ctl->thr21_prev_f;
    }
    const bool nonneg = VFORM && (a->flags & F_NONNEG);
    // (NonNegCoef as max(y, 0) in one instruction: med3(y, lo, +inf) with lo = 0, or -inf when off)
    const float nn_lo = nonneg ? 0.f : -__builtin_inff();
    int64_t p;
    bool pv;
    int cn, k;
    if constexpr (JOINT) {
        const int kbn = a->K >> 5, n = bx / kbn, kb = bx % kbn;
        const int c = lane >> 4;
        pv = c < a->C;
        cn = pv ? c * a->N + n : 0;
        k = pv ? kb * 32 + 2 * (lane & 15) : 0;
        p = (int64_t)cn * a->K + k;
    } else {
        p = (int64_t)bx * 128 + 2 * lane;
        pv = p < a->P;
        cn = pv ? (int)(p / a->K) : 0;
        k = pv ? (int)(p % a->K) : 0;
    }
    f2 *L = dyn_lds<f2>();
    int token = 0;

    // ---- spatial side: z[n1] = (Y - s2 U)(h, x = NW n1 + w, p..p+1) -------------------
    const int64_t rowoff = (int64_t)h * W * a->P;
    const uint32_t rowbytes = (uint32_t)((int64_t)W * a->P * sizeof(float));
    // (u may be null: a zero-length buffer then reads as zeros)
    // BCAST: y is (H, W, K) and is broadcast over the (c, n) blocks of K filters -- the
    // consensus dictionary update transforms Y[.., k] - s U[.., n, k] (admm/ccmod.py:768)
    const BufRsrc Yb = BCAST ? make_rsrc(a->y + (int64_t)h * W * a->K,
                                         (uint32_t)((int64_t)W * a->K * sizeof(float)))
                             : make_rsrc(a->y + rowoff, rowbytes);
    const BufRsrc Ub = VFORM ? make_rsrc(a->v + rowoff, rowbytes)
                             : (a->u ? make_rsrc(a->u + rowoff, rowbytes) : make_rsrc(a->y, 0u));
    const int voff = pv ? (int)(p * (int64_t)sizeof(float)) : (int)0x80000000;  // masked lanes read 0
    const int pixbytes = (int)(a->P * (int64_t)sizeof(float));
    const int yvoff = BCAST ? (pv ? k * (int)sizeof(float) : (int)0x80000000) : voff;
    const int ypixbytes = BCAST ? a->K * (int)sizeof(float) : pixbytes;
    // the per-element constants of the epilogue's prox (rows_inv_post_tile), when the options need them
    int wlane = 0;
    const int ws4 = GENERAL ? (int)a->wl1.stride[4] : 0;
    bool hkill = false, am_e[2] = {false, false};
    int x0kill = 0;
    uint32_t mbits = 0u;
    if constexpr (GENERAL) {
        const bool nob = a->flags & F_NOBNDRY;
        if (MODE == 1) {
            const int c = cn / a->N, n = cn % a->N;
            wlane = (int)(c * a->wl1.stride[2] + n * a->wl1.stride[3] + k * a->wl1.stride[4]);
        }
        hkill = nob && h >= ((a->dH > 1) ? a->H - (a->dH - 1) : 0);
        x0kill = nob ? ((a->dW > 1) ? W - (a->dW - 1) : 0) : W;
        const bool aml = a->ams_bits != nullptr && pv && (k | 1) == (a->ams_k | 1);
        am_e[0] = aml && !(a->ams_k & 1);
        am_e[1] = aml && (a->ams_k & 1);
        const BufRsrc Mb = make_rsrc(a->ams_bits, a->ams_bits ? (uint32_t)((int64_t)a->H * a->CN * NW * 4) : 0u);
        const int mvoff = aml ? cn * NW * 4 : (int)0x80000000;
        mbits = __builtin_bit_cast(uint32_t, sa_buf_load1(Mb, mvoff, (h * a->CN * NW + w) * 4));
    }
    cf v[N1];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // (two halves of (N1 + 1) / 2 and N1 / 2 pixels: HB = the first half's length)
        constexpr int HB = (N1 + 1) / 2;
        cf yv[HB], uv[HB];
#pragma unroll
        for (int i = 0; i < HB; ++i) {
            if (half * HB + i >= N1) continue;
            const int n1 = half * HB + i;
            const int soff = (NW * n1 + w) * pixbytes;
            if constexpr (!VFORM) yv[i] = buf_load_cf(Yb, yvoff, (NW * n1 + w) * ypixbytes);
            uv[i] = buf_load_cf(Ub, voff, soff);
        }
        if constexpr (VFORM) {
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                // (no fused multiply-adds here or in the epilogue: the (Y, U) and the V form of
                // an iteration must round alike, and which product of a sum the compiler
                // fuses depends on the code around it)
#pragma clang fp contract(off)
                if (half * HB + i >= N1) continue;
                const cf vv = uv[i];
                float t0 = thr_p, t1 = thr_p;
                if constexpr (GENERAL) {
                    float w0 = 1.f, w1 = 1.f;
                    if (MODE == 1) {
                        const int xw = NW * (half * HB + i) + w;
                        const float *wrow = a->wl1.ptr + (int64_t)h * a->wl1.stride[0] +
                                            (int64_t)xw * a->wl1.stride[1];
                        w0 = wrow[wlane];
                        w1 = wrow[wlane + ws4];
                    }
                    t0 = thr_p * (am_e[0] ? 0.f : w0);
                    t1 = thr_p * (am_e[1] ? 0.f : w1);
                }
                float y0 = soft1_m<MODE>(vv.re, t0), y1 = soft1_m<MODE>(vv.im, t1);
                if constexpr (JOINT) {      // the l2 shrinkage over the channels, as the epilogue
                    float q0 = y0 * y0, q1 = y1 * y1;
                    sum_over_rows2(q0, q1);
                    float f0 = sa_fma(-thr21_p, sa_rsq(q0), 1.f);
                    float f1 = sa_fma(-thr21_p, sa_rsq(q1), 1.f);
                    f0 = f0 > 0.f ? f0 : 0.f;
                    f1 = f1 > 0.f ? f1 : 0.f;
                    y0 = f0 * y0;
                    y1 = f1 * y1;
                }
                y0 = sa_med3(y0, am_e[0] ? -__builtin_inff() : nn_lo, __builtin_inff());
                y1 = sa_med3(y1, am_e[1] ? -__builtin_inff() : nn_lo, __builtin_inff());
                if constexpr (GENERAL) {
                    const int n1 = half * HB + i;
                    const float keep = (hkill || NW * n1 + w >= x0kill) ? 0.f : 1.f;
                    const float mkeep = ((mbits >> n1) & 1u) ? 0.f : 1.f;
                    y0 *= am_e[0] ? mkeep : keep;
                    y1 *= am_e[1] ? mkeep : keep;
                }
                yv[i] = mk<float>(y0, y1);
                uv[i] = mk<float>(vv.re - y0, vv.im - y1);
            }
        }
#pragma unroll
        for (int i = 0; i < HB; ++i) {
#pragma clang fp contract(off)
            if (half * HB + i >= N1) continue;
            v[half * HB + i] = mk<float>(sa_fma(-s2, uv[i].re, yv[i].re), sa_fma(-s2, uv[i].im, yv[i].im));
        }
        // (one call where the halves are alike: a branch on the unrolled loop's counter around the
        // fence cost the joint V form 13 registers and 52 bytes of scratch)
        if constexpr ((N1 & 1) == 0) reg_fence<HB>(v, half * HB, token);
        else if (half == 0) reg_fence<HB>(v, 0, token);
        else reg_fence<N1 - HB>(v, HB, token);
    }
    spatial_to_spectral<NW, COH, N1>(v, a->twA, a->t, a->CN, a->H, a->Ks ? a->Ks : a->K, cn, k, h, pv, w, lane, L,
                                     token);
}

// Tile loop shared by the row kernels: a 1-D grid of G workgroups, workgroup b takes the tiles
// b, b + G, ...  A persistent launch (G = what the device runs at once) pays no workgroup
// launch per tile (16 waves + 128 KiB of LDS + the argument loads: several microseconds on a
// CU that holds one workgroup) and lets a tile's trailing stores drain under the next tile's
// loads; G = number of tiles gives one tile per workgroup.  The tile code reads the kernel
// arguments through an opaque pointer obtained per tile (sa_args_reload): nothing derived from
// them is hoisted in front of the loop, where it would hold registers throughout.
template <typename A, typename F>
__device__ __forceinline__ void rows_tile_loop(const A &a_in, int tiles_x, int tiles_y, F &&tile_fn) {
    const int64_t ntiles = (int64_t)tiles_x * tiles_y;
    if (a_in.persist) {
        const int ph = (int)(blockIdx.x % (unsigned)a_in.stagger_groups);
        for (int i = 0; i < ph * a_in.stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        tile_fn(sa_args_reload(a_in), (int)(t % tiles_x), (int)(t / tiles_x));
        __syncthreads();     // the exchange buffer is reused by the next tile
    }
}

template <int NW, bool BCAST, bool VFORM = false, bool JOINT = false, int MODE = 0, int N1 = kN1>
__global__ void __launch_bounds__(NW * 64) rows_fwd_kernel(const RowsFwdArgs<float> a_in) {
    // device-driven solve: nothing to do once the stopping test is met, or when the previous
    // epilogue already left this spectrum behind
    if (a_in.ctl && (a_in.ctl->stop | a_in.ctl->skip_fwd)) return;
    const int tiles_x = JOINT ? a_in.N * (a_in.K >> 5) : (int)((a_in.P + 127) / 128);
    rows_tile_loop(a_in, tiles_x, a_in.H,
                   [](auto a, int bx, int h) { rows_fwd_tile<NW, BCAST, VFORM, JOINT, MODE, false, N1>(a, bx, h); });
}

// ---------------------------------------------------------------------------
// rows_inv_post: X = irfft_W(T) / (H W); relax, shrink, dual update, sums
// ---------------------------------------------------------------------------
// MODE: 0 = plain epilogue; 1 = L1Weight array (+ NoBndryCross, AddMaskSim); 2 = NoBndryCross
// and / or AddMaskSim without a weight array (no weight loads).
// SF (state form, csc_rows.h): 0 = (Y, U) in and out; 1 = (Y, U) in, V' out; 2 = V in, V' out.
// largest divisor of n that is <= want (pixels per batch of the epilogue: N1 = 20 ... 30 have other
// divisors than the powers of two)
constexpr int batch_of(int n, int want) {
    int b = 1;
    for (int d = 1; d <= want; ++d)
        if (n % d == 0) b = d;
    return b;
}
template <int NW, bool WRITE_X, int MODE, bool EMIT_T, bool JOINT, int SF, bool COH = false, int N1 = kN1,
          typename AP>
__device__ __forceinline__ void rows_inv_post_tile(AP a, int bx, int h, int tiles_x) {
    constexpr bool GENERAL = MODE != 0;
    constexpr bool VIN = SF == 2, VOUT = SF != 0;
    static_assert(!JOINT || MODE == 0, "the joint epilogue takes scalar weights only");
    static_assert(SF == 0 || !WRITE_X, "V form: no X output");
    constexpr int W = N1 * NW;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    float thr = a->thr, usc = a->u_scale, thr_p = a->thr_prev;
    if (a->ctl) {     // device-driven solve
        thr = a->ctl->thr_f;
        usc = a->ctl->u_scale_f;
        thr_p = a->ctl->thr_prev_f;
    }
    // columns of this thread: 128 consecutive ones of (c, n, k) per workgroup -- or, JOINT,
    // (channel lane >> 4, image blockIdx / (K/32), filters 32 (blockIdx % (K/32)) + 2 (lane & 15))
    const int CN = a->C * a->N;
    int64_t p;
    bool pv;
    int cn, k;
    if constexpr (JOINT) {
        const int kbn = a->K >> 5, n = bx / kbn, kb = bx % kbn;
        const int c = lane >> 4;
        pv = c < a->C;
        cn = pv ? c * a->N + n : 0;
        k = pv ? kb * 32 + 2 * (lane & 15) : 0;
        p = (int64_t)cn * a->K + k;
    } else {
        p = (int64_t)bx * 128 + 2 * lane;
        pv = p < a->P;
        cn = pv ? (int)(p / a->K) : 0;
        k = pv ? (int)(p % a->K) : 0;
    }
    f2 *L = dyn_lds<f2>();
    double *scratch = reinterpret_cast<double *>(L + 16 * NW * 64);
    int token = 0;

    // pixels per batch (the previous iterate of the next batch is in flight); the emitting variants
    // keep the tile for the forward transform and have fewer registers to spare
    // (V form reads one array instead of two: twice the pixels per batch for the same registers)
#ifndef SA_POST_B_VIN
#define SA_POST_B_VIN 8
#endif
    // PARK (emitting V-form epilogues): the second half of the tile -- x at the pixels n1 >= 16 --
    // waits in the idle exchange buffer while the first half is worked on, each value in the slot
    // column its own thread reads and writes in the exchanges (no barrier), and each result takes the
    // place of the value it came from; the half comes back into registers for the forward transform.
    // The 32 registers this frees hold the previous iterate of more pixels in flight: the joint
    // variant had room for ONE pixel per thread (8 KiB per CU requested at a time against the
    // ~64 KiB that keep a CU's share of the memory pipe full), now SA_POST_B_JOINT_PARK.
#ifndef SA_PARK_JOINT
#define SA_PARK_JOINT 1
#endif
#ifndef SA_PARK_PLAIN
#define SA_PARK_PLAIN 0
#endif
#ifndef SA_POST_B_JOINT_PARK
#define SA_POST_B_JOINT_PARK 4
#endif
#ifndef SA_POST_B_PLAIN_PARK
#define SA_POST_B_PLAIN_PARK 8
#endif
    constexpr bool PARK = EMIT_T && VIN && !regfft::mr_length(N1) && (JOINT ? SA_PARK_JOINT != 0 : SA_PARK_PLAIN != 0);
    constexpr int NP = N1 / 2;      // parked pixels per thread
    constexpr int B = batch_of(PARK ? NP : N1, EMIT_T ? (JOINT ? (PARK ? SA_POST_B_JOINT_PARK : 1)
                                                               : (VIN ? (PARK ? SA_POST_B_PLAIN_PARK : 4) : 2))
                                                      : ((VIN && !JOINT) ? SA_POST_B_VIN : 4));
    cf yb[2][B], ub[2][B];
    cf v[N1];
    if constexpr (PARK) {
        // the first batch of the previous iterate is requested under the last in-register transform
        auto first_fetch = [&]() {
            const int64_t rowoff0 = (int64_t)h * W * a->P;
            const BufRsrc Vb0 = make_rsrc(a->v_in + rowoff0, (uint32_t)((int64_t)W * a->P * sizeof(float)));
            const int voff0 = pv ? (int)(p * (int64_t)sizeof(float)) : (int)0x80000000;
            const int pixbytes0 = (int)(a->P * (int64_t)sizeof(float));
#pragma unroll
            for (int i = 0; i < B; ++i) yb[0][i] = buf_load_cf(Vb0, voff0, (NW * i + w) * pixbytes0);
        };
        spectral_to_spatial<NW, COH, N1>(v, a->twW, a->t, CN, a->H, a->Ks ? a->Ks : a->K, cn, k, h, pv, w, lane, L,
                                         token, a->t_odd, first_fetch);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            f2 t;
            t.x = v[NP + i].re;
            t.y = v[NP + i].im;
            L[(i * NW + w) * 64 + lane] = t;
        }
    } else {
        spectral_to_spatial<NW, COH, N1>(v, a->twW, a->t, CN, a->H, a->Ks ? a->Ks : a->K, cn, k, h, pv, w, lane, L,
                                         token, a->t_odd);
    }

    // ---- ADMM epilogue on the 32 pixels of this thread ---------------------------------------
    const int64_t rowoff = (int64_t)h * W * a->P;
    const uint32_t rowbytes = (uint32_t)((int64_t)W * a->P * sizeof(float));
    // (V form: the one input array through Yb, the one output array through Yo)
    const BufRsrc Yb = make_rsrc((VIN ? a->v_in : a->y) + rowoff, rowbytes);
    const BufRsrc Ub = VIN ? Yb : make_rsrc(a->u + rowoff, rowbytes);
    const BufRsrc Yo = make_rsrc((VOUT ? a->v_out : a->y_out) + rowoff, rowbytes);
    const BufRsrc Uo = VOUT ? Yo : make_rsrc(a->u_out + rowoff, rowbytes);
    const BufRsrc Xb = make_rsrc(WRITE_X ? a->x + rowoff : a->y_out + rowoff, rowbytes);
    const int voff = pv ? (int)(p * (int64_t)sizeof(float)) : (int)0x80000000;
    const int pixbytes = (int)(a->P * (int64_t)sizeof(float));
    const float al = a->rlx, oma = 1.f - a->rlx, scale = a->scale;
    const bool nonneg = a->flags & F_NONNEG, nob = a->flags & F_NOBNDRY, gy = a->flags & F_GEVAL_Y;
    const float nn_lo = nonneg ? 0.f : -__builtin_inff();     // (see rows_fwd_tile)
    // weight of element (h, x, c, n, k): wave-uniform row pointer + 32-bit lane offset
    int wlane = 0;
    const int ws4 = (int)a->wl1.stride[4];
    if (MODE == 1) {
        const int c = cn / a->N, n = cn % a->N;
        wlane = (int)(c * a->wl1.stride[2] + n * a->wl1.stride[3] + k * a->wl1.stride[4]);
    }
    const bool hkill = GENERAL && nob && h >= ((a->dH > 1) ? a->H - (a->dH - 1) : 0);
    const int x0kill = (a->dW > 1) ? W - (a->dW - 1) : 0;
    // AddMaskSim (cbpdn.py:2378-2412): the lanes whose filter pair holds the impulse slice
    // ams_k read the mask of their (c, n) and row: one 32-bit word per thread, bit n1 for the
    // pixel x = NW n1 + w (ams_bits, packed by launch_ams_pack); every other lane (and every
    // lane without a mask) sends an out-of-range offset, which costs no memory traffic and
    // returns 0.
    const bool aml = GENERAL && a->ams_bits != nullptr && pv && (k | 1) == (a->ams_k | 1);
    const bool am_e[2] = {aml && !(a->ams_k & 1), aml && (a->ams_k & 1)};
    uint32_t mbits = 0u;
    if (GENERAL) {
        const BufRsrc Mb = make_rsrc(a->ams_bits, a->ams_bits ? (uint32_t)((int64_t)a->H * CN * NW * 4) : 0u);
        const int mvoff = aml ? cn * NW * 4 : (int)0x80000000;
        mbits = __builtin_bit_cast(uint32_t, sa_buf_load1(Mb, mvoff, (h * CN * NW + w) * 4));
    }
    float s_r2 = 0.f, s_s2 = 0.f, s_x2 = 0.f, s_y2 = 0.f, s_u2 = 0.f, s_l1 = 0.f, s_l21 = 0.f;
    float thr21 = a->thr21, thr21_p = a->thr21_prev;
    if (JOINT && a->ctl) {
        thr21 = a->ctl->thr21_f;
        thr21_p = a->ctl->thr21_prev_f;
    }
    const float l21w = lane < 16 ? 1.f : 0.f;  // the l2,1 sum counts each channel group once
    const float emit_y = (EMIT_T && SF == 0 && a->emit_u) ? 0.f : 1.f;
    const float emit_s = (EMIT_T && SF == 0 && a->emit_u) ? -1.f : 1.f;
    auto fetch = [&](int slot, int b) {
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int soff = (NW * (b * B + i) + w) * pixbytes;
            yb[slot][i] = buf_load_cf(Yb, voff, soff);
            if constexpr (!VIN) ub[slot][i] = buf_load_cf(Ub, voff, soff);
        }
    };
    if constexpr (!PARK) fetch(0, 0);
    static_for<N1 / B>([&](auto bc) {
        // (every product rounded: the state forms of csc_rows.h must agree bit for bit, see
        // rows_fwd_tile)
#pragma clang fp contract(off)
        constexpr int b = decltype(bc)::value;
        if constexpr (b + 1 < N1 / B) fetch((b + 1) & 1, b + 1);
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int n1 = b * B + i;
            const int xw = NW * n1 + w;
            const int soff = xw * pixbytes;
            // (a batch lies in one half of the tile: B divides N1 / 2)
            constexpr bool parked = PARK && b * B >= NP;
            cf xraw = v[parked ? 0 : n1];
            if constexpr (parked) {
                const f2 t = L[((n1 - NP) * NW + w) * 64 + lane];
                xraw = mk<float>(t.x, t.y);
            }
            const float xs[2] = {xraw.re * scale, xraw.im * scale};
            float yo[2] = {yb[b & 1][i].re, yb[b & 1][i].im};
            float uraw[2];
            // NoBndryCross as a multiplicative mask (a uniform branch here splits the unrolled
            // epilogue into dozens of blocks and the register allocator spills the tile)
            const float keep = (GENERAL && (hkill || (nob && xw >= x0kill))) ? 0.f : 1.f;
            const float mkeep = (GENERAL && ((mbits >> n1) & 1u)) ? 0.f : 1.f;
            float wte[2] = {1.f, 1.f};
            if constexpr (!JOINT && GENERAL) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (MODE == 1) {
                        const float *wrow = a->wl1.ptr + (int64_t)h * a->wl1.stride[0] +
                                            (int64_t)xw * a->wl1.stride[1];
                        wte[e] = wrow[wlane + e * ws4];
                    }
                    wte[e] = am_e[e] ? 0.f : wte[e];
                }
            }
            if constexpr (VIN && JOINT) {
                // the previous iterate from its V, both elements of the pixel together: the
                // channel sums of their squares travel through one set of swaps (sum_over_rows2)
                const float vp[2] = {yo[0], yo[1]};
                float yp[2] = {soft1_pos(vp[0], thr_p), soft1_pos(vp[1], thr_p)};
                float qp[2] = {yp[0] * yp[0], yp[1] * yp[1]};
                sum_over_rows2(qp[0], qp[1]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float fp = sa_fma(-thr21_p, sa_rsq(qp[e]), 1.f);
                    fp = fp > 0.f ? fp : 0.f;
                    yp[e] = sa_med3(fp * yp[e], nn_lo, __builtin_inff());
                    yo[e] = yp[e];
                    uraw[e] = vp[e] - yp[e];
                }
            } else if constexpr (VIN) {
                // the previous iterate from its V: Y = prox(V; thr_prev) (+ the options), U = V - Y
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float vp = yo[e];
                    float yp = soft1_m<MODE>(vp, thr_p * wte[e]);
                    yp = sa_med3(yp, (GENERAL && am_e[e]) ? -__builtin_inff() : nn_lo, __builtin_inff());
                    if constexpr (GENERAL) yp *= am_e[e] ? mkeep : keep;
                    yo[e] = yp;
                    uraw[e] = vp - yp;
                }
            } else {
                uraw[0] = ub[b & 1][i].re;
                uraw[1] = ub[b & 1][i].im;
            }
            float yn[2], un[2], vn[2] = {0.f, 0.f};
            if constexpr (JOINT) {
                // prox_sl1l2 over the channel axis (cbpdn.py:785-794): soft threshold, then the
                // channel vector of each (pixel, image, filter) shrunk in l2 norm,
                // y = s max(0, 1 - thr21 / ||s||) (prox/_lp.py:283-290, zero where ||s|| = 0).
                // The channels sit 16 lanes apart (idle lanes hold zeros); the two elements of the
                // pixel share each of the three channel sums' swaps (sum_over_rows2).  Stage by
                // stage with fences between: the emitting variant keeps the whole tile live for the
                // forward transform and has no registers for the scheduler's interleavings (it
                // spilled 544 bytes of scratch in round 2).
                float sv[2], q[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float ax = sa_fma(al, xs[e], oma * yo[e]);
                    vn[e] = sa_fma(usc, uraw[e], ax);
                    sv[e] = soft1_pos(vn[e], thr);
                    q[e] = sv[e] * sv[e];
                }
                if constexpr (EMIT_T) {
                    SA_VGPR_FENCE3(sv[0], sv[1], vn[0]);
                    SA_VGPR_FENCE3(q[0], q[1], vn[1]);
                }
                sum_over_rows2(q[0], q[1]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    float fac = sa_fma(-thr21, sa_rsq(q[e]), 1.f);   // (q = 0: -inf, or NaN when thr21 = 0)
                    fac = fac > 0.f ? fac : 0.f;
                    const float y1 = sa_med3(fac * sv[e], nn_lo, __builtin_inff());
                    const float u1 = vn[e] - y1;
                    yn[e] = y1;
                    un[e] = u1;
                    const float dr = xs[e] - y1, ds = y1 - yo[e];
                    s_r2 = sa_fma(dr, dr, s_r2);
                    s_s2 = sa_fma(ds, ds, s_s2);
                    s_x2 = sa_fma(xs[e], xs[e], s_x2);
                    s_y2 = sa_fma(y1, y1, s_y2);
                    s_u2 = sa_fma(u1, u1, s_u2);
                }
                if constexpr (EMIT_T) {
                    SA_VGPR_FENCE3(s_r2, s_s2, s_x2);
                    SA_VGPR_FENCE3(s_y2, s_u2, s_l1);
                }
                // (always formed: a branch on F_OBJ here would split the unrolled epilogue
                // into blocks and spill the tile, see the NoBndryCross note above)
                const float gv0 = gy ? yn[0] : xs[0], gv1 = gy ? yn[1] : xs[1];
                s_l1 += fabsf(gv0);
                s_l1 += fabsf(gv1);
                float g2[2] = {gv0 * gv0, gv1 * gv1};
                sum_over_rows2(g2[0], g2[1]);
                s_l21 = sa_fma(l21w, sa_sqrt(g2[0]), s_l21);
                s_l21 = sa_fma(l21w, sa_sqrt(g2[1]), s_l21);
                if constexpr (EMIT_T) SA_VGPR_FENCE3(s_l21, s_l1, s_r2);
            } else {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ax = sa_fma(al, xs[e], oma * yo[e]);
                const bool am = GENERAL && am_e[e];
                const float wt = wte[e];
                // V' = AX + U: the new iterate is a function of it alone (Y' = prox(V'),
                // U' = V' - Y'), which is what the V form stores
                const float vv = sa_fma(usc, uraw[e], ax);
                float y1 = soft1_m<MODE>(vv, thr * wt);
                y1 = sa_med3(y1, am ? -__builtin_inff() : nn_lo, __builtin_inff());
                if (GENERAL) y1 *= am ? mkeep : keep;
                const float u1 = vv - y1;
                yn[e] = y1;
                un[e] = u1;
                vn[e] = vv;
                const float dr = xs[e] - y1, ds = y1 - yo[e];
                s_r2 = sa_fma(dr, dr, s_r2);
                s_s2 = sa_fma(ds, ds, s_s2);
                s_x2 = sa_fma(xs[e], xs[e], s_x2);
                s_y2 = sa_fma(y1, y1, s_y2);
                s_u2 = sa_fma(u1, u1, s_u2);
                s_l1 += fabsf(wt * (gy ? y1 : xs[e]));
            }
            }
            if constexpr (VOUT) {
                buf_store_cf(Yo, voff, soff, mk<float>(vn[0], vn[1]));
            } else {
                buf_store_cf(Yo, voff, soff, mk<float>(yn[0], yn[1]));
                buf_store_cf(Uo, voff, soff, mk<float>(un[0], un[1]));
            }

            if (WRITE_X) buf_store_cf(Xb, voff, soff, mk<float>(xs[0], xs[1]));
            if constexpr (EMIT_T && SF == 0) {
                // (emit_u: the spectrum of U' alone; 1 * y - u rounds as y - u does)
                v[n1] = mk<float>(emit_y * yn[0] - emit_s * un[0], emit_y * yn[1] - emit_s * un[1]);
            } else if constexpr (PARK && b * B >= NP) {
                f2 t;
                t.x = yn[0] - un[0];
                t.y = yn[1] - un[1];
                L[((n1 - NP) * NW + w) * 64 + lane] = t;
            } else if (EMIT_T) {
                v[n1] = mk<float>(yn[0] - un[0], yn[1] - un[1]);
            }
        }
    });
    if constexpr (PARK) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const f2 t = L[(i * NW + w) * 64 + lane];
            v[NP + i] = mk<float>(t.x, t.y);
        }
    }

    // (pin the six sums here: left alone, the compiler sinks their accumulation below the
    // transform that follows and keeps every per-element term alive until then)
    SA_VGPR_FENCE3(s_r2, s_s2, s_x2);
    SA_VGPR_FENCE3(s_y2, s_u2, s_l1);
    if (EMIT_T) {
        // Speculation on an unchanged rho: the row spectra of Y' - U' that the next
        // iteration's rows_fwd would compute from these very values, stored over the
        // units this thread consumed (same spectral-side ownership: in place is safe).
        reg_fence<N1>(v, 0, token);
        spatial_to_spectral<NW, COH, N1>(v, a->twA, a->t_next, CN, a->H, a->Ks ? a->Ks : a->K, cn, k, h, pv, w, lane,
                                         L, token);
        __syncthreads();   // the reduction scratch below sits next to the exchange buffer
    }

    // masked lanes contributed zeros everywhere except possibly the threshold of 0: their
    // inputs are all zero, so every term above is exactly 0
    double acc[8] = {(double)s_r2, (double)s_s2, (double)s_x2,   (double)s_y2,
                     (double)s_u2, (double)s_l1, (double)s_l21, 0.0};
    const int64_t tile = (int64_t)h * tiles_x + bx;
    block_sum_store<8, COH>(acc, scratch, a->partials + tile * 8);
}

// (two 8-wave workgroups share a CU only at <= 128 registers: the joint emitting variants land a
// register or two above that on their own, so they are told -- second argument = waves per SIMD)
template <int NW, bool WRITE_X, int MODE, bool EMIT_T, bool JOINT = false, int SF = 0, int N1 = kN1>
__global__ void __launch_bounds__(NW * 64, (JOINT && EMIT_T && NW == 8) ? 4 : 1)
rows_inv_post_kernel(const RowsPostArgs<float> a_in) {
    // device-driven solve: both variants are enqueued every iteration and the one whose EMIT_T
    // matches the speculation decision runs (an idle launch costs about 12 us; the emitting
    // variant keeps half as many Y / U loads in flight, so it is not the one to run when
    // nothing is emitted)
    if (a_in.ctl && (a_in.ctl->stop | (a_in.ctl->emit != (EMIT_T ? 1 : 0)))) return;
    const int tiles_x = JOINT ? a_in.N * (a_in.K >> 5) : (int)((a_in.P + 127) / 128);
    rows_tile_loop(a_in, tiles_x, a_in.H, [tiles_x](auto a, int bx, int h) {
        rows_inv_post_tile<NW, WRITE_X, MODE, EMIT_T, JOINT, SF, false, N1>(a, bx, h, tiles_x);
    });
}

// ---------------------------------------------------------------------------
// rows_inv_prox_fwd: X = prox_l1(irfft_W(T_in) / (H W)); T_out = rfft_W(X)
// ---------------------------------------------------------------------------
template <int NW, bool GENERAL, int N1 = kN1, typename AP>
__device__ __forceinline__ void rows_inv_prox_fwd_tile(AP ap, int bx, int h, int tiles_x) {
    constexpr int W = N1 * NW;
    const auto &a = *ap;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int64_t p = (int64_t)bx * 128 + 2 * lane;
    const bool pv = p < a.P;
    const int CN = a.C * a.N;
    const int cn = pv ? (int)(p / a.K) : 0, k = pv ? (int)(p % a.K) : 0;
    f2 *L = dyn_lds<f2>();
    double *scratch = reinterpret_cast<double *>(L + 16 * NW * 64);
    int token = 0;

    cf v[N1];
    spectral_to_spatial<NW, false, N1>(v, a.twW, a.t_in, CN, a.H, a.Ks ? a.Ks : a.K, cn, k, h, pv, w, lane, L,
                                       token);

    // ---- proximal step on the 32 pixels of this thread -----------------------------------------
    const int64_t rowoff = (int64_t)h * W * a.P;
    const uint32_t rowbytes = (uint32_t)((int64_t)W * a.P * sizeof(float));
    // no X requested: a zero-length buffer drops every store
    const BufRsrc Xb = a.x ? make_rsrc(a.x + rowoff, rowbytes) : make_rsrc(a.t_in, 0u);
    const int voff = pv ? (int)(p * (int64_t)sizeof(float)) : (int)0x80000000;
    const int pixbytes = (int)(a.P * (int64_t)sizeof(float));
    const float scale = a.scale;
    const bool nonneg = a.flags & F_NONNEG, nob = a.flags & F_NOBNDRY;
    int wlane = 0;
    const int ws4 = (int)a.wl1.stride[4];
    if (GENERAL) {
        const int c = cn / a.N, n = cn % a.N;
        wlane = (int)(c * a.wl1.stride[2] + n * a.wl1.stride[3] + k * a.wl1.stride[4]);
    }
    const bool hkill = GENERAL && nob && h >= ((a.dH > 1) ? a.H - (a.dH - 1) : 0);
    const int x0kill = (a.dW > 1) ? W - (a.dW - 1) : 0;
    const float nn_lo = nonneg ? 0.f : -__builtin_inff();     // (see rows_fwd_tile)
    float s_l1 = 0.f;
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) {
        const int xw = NW * n1 + w;
        const float keep = (GENERAL && (hkill || (nob && xw >= x0kill))) ? 0.f : 1.f;
        float y[2] = {v[n1].re * scale, v[n1].im * scale};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float wt = 1.f;
            if (GENERAL) {
                const float *wrow = a.wl1.ptr + (int64_t)h * a.wl1.stride[0] +
                                    (int64_t)xw * a.wl1.stride[1];
                wt = wrow[wlane + e * ws4];
            }
            // (the plain variant is launched with a threshold >= 0 only: launch_prox_nw)
            float y1 = GENERAL ? soft1(y[e], a.thr * wt) : soft1_pos(y[e], a.thr);
            y1 = sa_med3(y1, nn_lo, __builtin_inff());
            if (GENERAL) y1 *= keep;
            s_l1 += fabsf(wt * y1);
            y[e] = y1;
        }
        v[n1] = mk<float>(y[0], y[1]);
        buf_store_cf(Xb, voff, xw * pixbytes, v[n1]);
    }
    double acc[1] = {(double)s_l1};
    const int64_t tile = (int64_t)h * tiles_x + bx;
    block_sum_store<1>(acc, scratch, a.partials + tile);
    if (!a.t_out) return;
    reg_fence<N1>(v, 0, token);

    spatial_to_spectral<NW, false, N1>(v, a.twA, a.t_out, CN, a.H, a.Ks ? a.Ks : a.K, cn, k, h, pv, w, lane, L,
                                       token);
}

template <int NW, bool GENERAL, int N1 = kN1>
__global__ void __launch_bounds__(NW * 64) rows_inv_prox_fwd_kernel(const RowsProxArgs<float> a_in) {
    const int tiles_x = (int)((a_in.P + 127) / 128);
    rows_tile_loop(a_in, tiles_x, a_in.H, [tiles_x](auto a, int bx, int h) {
        rows_inv_prox_fwd_tile<NW, GENERAL, N1>(a, bx, h, tiles_x);
    });
}


// ---------------------------------------------------------------------------
// admm_persist: a run of iterations in one launch (csc_rows.h)
// ---------------------------------------------------------------------------
constexpr int kBarGroup0 = 16, kBarStride = 16;   // group counters: bar[16 + 16 g], g = 0..7; the top one at g = 8
// Barrier across the grid.  What changes hands between workgroups -- the tile-major spectrum
// and the tile sums -- is written and read with agent-scope accesses (COH above: written
// through, read past the non-coherent cache levels), so no cache is written back or dropped
// here (an agent-scope release / acquire pair per workgroup does that to the whole L2 and
// costs more than the passes themselves): every wave waits for its stores to be acknowledged,
// one thread per workgroup arrives on a counter and waits for the generation to change.
// Everything else an iteration touches is private to a workgroup (its tiles of V -- the same
// tiles in every pass --, its control block, its argument copies).  The grid is never
// larger than the device holds at once, so whoever waits, waits for a resident workgroup; a
// wait that does not complete (2^22 polls) raises bar[2] and later barriers do not wait.
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned nblk) {
    sa_wait_stores();
    __syncthreads();
    if (threadIdx.x == 0 && sa_load_agent(bar + 2) == 0u) {
        // two levels (256 arrivals on one word are served one after the other, a few
        // microseconds in all): eight groups of nblk / 8 workgroups, then the eight groups; the
        // counters only ever count up (the last of a group is the one that completes a multiple
        // of the group size), so nothing is reset and nothing can be reset late
        const unsigned gen = sa_load_agent(bar + 1);
        const unsigned g = blockIdx.x & 7u, per = nblk >> 3;
        bool last = false;
        if ((sa_atomic_inc_agent(bar + kBarGroup0 + kBarStride * g) + 1u) % per == 0u)
            last = (sa_atomic_inc_agent(bar + kBarGroup0 + kBarStride * 8) + 1u) % 8u == 0u;
        if (last) {
            sa_store_agent(bar + 1, gen + 1);
        } else {
            int polls = 0;
            while (sa_load_agent(bar + 1) == gen) {
                sa_spin_pause();
                if (++polls > (1 << 22)) {
                    sa_store_agent(bar + 2, 1u);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// The sums of one iteration from the tile partials, each in the association of finalize_kernel
// (csc_kernels.hip: 256 strided sums, then a fixed tree), all values side by side.
// scratch: 7 * kFinalizeThreads doubles; sums: the 16 output slots.
__device__ __forceinline__ void persist_finalize(const double *prow, int nrow, const double *pcol, int ncol,
                                                 bool dfid, double dfid_scale, double *scratch, double *sums) {
    constexpr int FT = kFinalizeThreads;
    const int nv = 6 + (dfid ? 1 : 0);
    const int tid = threadIdx.x, nth = blockDim.x;
    if (tid < 16) sums[tid] = 0.0;
    for (int idx = tid; idx < nv * FT; idx += nth) {
        const int v = idx / FT, t = idx % FT;
        const double *p = v < 6 ? prow + v : pcol;
        const int n = v < 6 ? nrow : ncol, st = v < 6 ? 8 : 1;
        double s = 0.0;
        for (int b = t; b < n; b += FT) s = s + sa_load_agent(p + (int64_t)b * st);
        scratch[idx] = s;
    }
    __syncthreads();
    for (int w = FT / 2; w > 0; w >>= 1) {
        for (int idx = tid; idx < nv * w; idx += nth) {
            const int v = idx / w, t = idx % w;
            scratch[v * FT + t] = scratch[v * FT + t] + scratch[v * FT + t + w];
        }
        __syncthreads();
    }
    if (tid < nv) {
        const int slots[7] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_S2, SPORCO_AMD_OUT_AX2, SPORCO_AMD_OUT_Y2,
                              SPORCO_AMD_OUT_U2, SPORCO_AMD_OUT_L1, SPORCO_AMD_OUT_DFID};
        sums[slots[tid]] = scratch[tid * FT] * (tid == 6 ? dfid_scale : 1.0);
    }
    __syncthreads();
}

template <int NW, int LP>
__global__ void __launch_bounds__(NW * 64) admm_persist_kernel(const AdmmPersistArgs<float> pa) {
    const int tid = threadIdx.x;
    const unsigned nblk = gridDim.x;
    AdmmCtl *c = pa.ctl_blk + blockIdx.x;
    // between the passes the exchange buffer holds the reduction scratch and the sums
    double *scratch = dyn_lds<double>();
    double *sums = scratch + 7 * kFinalizeThreads;
    // this workgroup's copies of the arguments (one per parity) and of the control block
    if (tid == 0) {
        *c = *pa.ctl;
        for (int par = 0; par < 2; ++par) {
            PersistIterArgs<float> *b = pa.blk + (size_t)par * nblk + blockIdx.x;
            *b = pa.iter[par];
            b->fwd.ctl = c;
            b->cols.ctl = c;
            b->post.ctl = c;
        }
    }
    sa_wait_stores();
    sa_scalar_cache_inv();
    __syncthreads();
    const int tiles_x = (int)((pa.iter[0].fwd.P + 127) / 128), H = pa.iter[0].fwd.H;
    const int64_t ntiles = (int64_t)tiles_x * H;
#ifdef SA_PERSIST_TIMING
    // (measurement builds only: where workgroup 0 spends its time, in ticks of the 100 MHz clock,
    // summed over the iterations into bar[8 ..])
    unsigned long long tl = sa_wall_clock(), tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SA_PT(i)                                   \
    {                                              \
        const unsigned long long t_ = sa_wall_clock(); \
        tacc[i] += t_ - tl;                        \
        tl = t_;                                   \
    }
#else
#define SA_PT(i)
#endif
    int it = 0;
    for (; it < pa.max_iter; ++it) {
        if (c->stop) break;
        const int par = (pa.index0 + it) & 1;
        SA_ARGS_PTR_T(PersistIterArgs<float>) pb = sa_opaque_sptr(
            (SA_ARGS_PTR_T(PersistIterArgs<float>))(pa.blk + (size_t)par * nblk + blockIdx.x));
        if (!c->skip_fwd) {     // rho moved: the emitted spectrum of Y - U is void
            for (int64_t t = blockIdx.x; t < ntiles; t += nblk) {
                rows_fwd_tile<NW, false, true, false, 0, true>(&pb->fwd, (int)(t % tiles_x), (int)(t / tiles_x));
                __syncthreads();
            }
            SA_PT(0)
            grid_barrier(pa.bar, nblk);
            SA_PT(1)
        }
        fused_cols_body<32, NW, LP, 0, false, false, false, 0, true, -1>(pa.iter[0].cols, &pb->cols);
        SA_PT(2)
        grid_barrier(pa.bar, nblk);
        SA_PT(3)
        for (int64_t t = blockIdx.x; t < ntiles; t += nblk) {
            rows_inv_post_tile<NW, false, 0, true, false, 2, true>(&pb->post, (int)(t % tiles_x), (int)(t / tiles_x),
                                                               tiles_x);
            __syncthreads();
        }
        SA_PT(4)
        grid_barrier(pa.bar, nblk);
        SA_PT(5)
        if (pa.want_sums) {
            persist_finalize(pb->post.partials, pa.n_row_tiles, pb->cols.partials, pa.n_col_tiles,
                             pa.want_dfid != 0, pa.dfid_scale, scratch, sums);
        } else {
            if (tid < 16) sums[tid] = 0.0;
            __syncthreads();
        }
        SA_PT(6)
        // (the host-visible record -- two system-scope fences -- is written by the LAST workgroup:
        // it has no tile in the column pass that follows, so nobody waits for it)
        if (tid == 0)
            admm_ctl_update_dev<float>(c, sums, blockIdx.x == nblk - 1 ? pa.rec + it : nullptr, pa.index0 + it);
        sa_wait_stores();
        sa_scalar_cache_inv();
        __syncthreads();
        SA_PT(7)
    }
    if (blockIdx.x == 0 && tid == 0) {
        *pa.ctl = *c;
        pa.bar[3] = (unsigned)it;
#ifdef SA_PERSIST_TIMING
        for (int i = 0; i < 8; ++i) pa.bar[8 + i] = (unsigned)tacc[i];
#endif
    }
#undef SA_PT
}

template <int NW, typename K>
void set_lds_attr(K kernel) {
    SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)rows_lds_bytes(NW)));
}

}  // namespace

// points per thread of a supported line length: 32 for the powers of two, W / 16 for the
// mixed-radix lengths 320, 384, 448, 480
static int rows_n1(int W) { return rows_mr_width(W) ? W / 16 : kN1; }
static int rows_rev(int N1, int i) {
    switch (N1) {
#define SA_MR_CASE(n) case n: return regfft::mr_rev<n>(i);
    SA_MR_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: return regfft::brev(i, 5);
    }
}

// a device-resident 1.0f: the weight array of the GENERAL epilogue when none is set
static const float *device_one() {
    static float *one = nullptr;
    if (!one) {
        const float v = 1.0f;
        SA_HIP(hipMalloc((void **)&one, sizeof(float)));
        SA_HIP(hipMemcpy(one, &v, sizeof(float), hipMemcpyHostToDevice));
    }
    return one;
}

#ifndef SA_ROWS_MR_TU
bool rows_mr_width(int W) { return W % 16 == 0 && regfft::mr_length(W / 16); }
template <> bool rows_supported<float>(int W, int K) {
    return (W == 128 || W == 256 || W == 512 || rows_mr_width(W)) && K >= 2 && K % 2 == 0;
}
template <> bool rows_supported<double>(int, int) { return false; }
template <> bool rows_joint_supported<float>(int W, int C, int K) {
    return rows_supported<float>(W, K) && C >= 1 && C <= 4 && K % 32 == 0;
}
template <> bool rows_joint_supported<double>(int, int, int) { return false; }

template <typename T> void rows_twiddles(int W, cx<T> *twA) {
    const int N1 = rows_n1(W), NW = W / N1;
    const double two_pi = 6.283185307179586476925286766559;
    for (int w = 0; w < NW; ++w)
        for (int i = 0; i < N1; ++i) {
            const double ang = -two_pi * (double)(w * rows_rev(N1, i)) / (double)W;
            twA[w * N1 + i] = mk<T>((T)std::cos(ang), (T)std::sin(ang));
        }
}
template void rows_twiddles<float>(int, cx<float> *);
template void rows_twiddles<double>(int, cx<double> *);
#endif

// Workgroups of a persistent row-kernel launch: what the device holds at once (one 16-wave
// workgroup per CU, two 8-wave ones).
static int64_t rows_persistent_grid(int NW) {
    return (int64_t)current_device_cus() * (NW == 16 ? 1 : NW == 8 ? 2 : 4);
}
// want: 1 = persistent; 0 = one workgroup per tile
template <typename A>
static dim3 rows_grid(A &a, int NW, int64_t tiles_x, int64_t tiles_y, int want) {
    const int64_t g = rows_persistent_grid(NW), n = tiles_x * tiles_y;
    SA_REQUIRE(n < ((int64_t)1 << 31), "too many tiles for one launch");
    a.persist = (want && g > 0 && n > g) ? 1 : 0;
    a.stagger_groups = 1;      // (a start-up stagger of the row kernels measured no gain)
    a.stagger_sleeps = 0;
    return dim3((unsigned)(a.persist ? g : n), 1);
}

#ifndef SA_ROWS_MR_TU
// ---- the one-launch solve ---------------------------------------------------------------------
template <> bool admm_persist_supported<float>(int H, int W, int K) {
    return H == W && (W == 128 || W == 256) && K >= 2 && K % 2 == 0 && K <= 64;
}
template <> bool admm_persist_supported<double>(int, int, int) { return false; }

static int persist_cus() { return current_device_cus(); }
template <> int admm_persist_grid<float>(int H, int W, int K, int CN) {
    (void)W;
    // one workgroup per CU at most, a multiple of 8 (the column pass gives workgroup b the row
    // frequencies = b mod 8), and no more than the larger pass has tiles
#ifdef SPORCO_AMD_HOSTSIM
    int g = 8;      // (the CPU test simulator runs the whole grid side by side: hostsim::set_coop)
#else
    int g = persist_cus() & ~7;
#endif
    const int64_t row_tiles = (int64_t)H * (((int64_t)CN * K + 127) / 128);
    const int64_t col_tiles = (int64_t)(W / 2 + 1) * CN;
    const int64_t most = std::max(row_tiles, (col_tiles + 7) / 8 * 8);
    if (g > most) g = (int)((most + 7) / 8 * 8);
    return g < 8 ? 8 : g;
}
template <> int admm_persist_grid<double>(int, int, int, int) { return 0; }

template <int NW, int LP> static size_t persist_prepare() {
    const size_t lds = std::max<size_t>(std::max(rows_lds_bytes(NW), fused_lds_bytes(NW, LP)),
                                        sizeof(double) * (7 * kFinalizeThreads + 16));
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&admm_persist_kernel<NW, LP>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    return lds;
}
template <int NW, int LP>
static void launch_persist_inst(hipStream_t st, const AdmmPersistArgs<float> &a, int grid) {
    const size_t lds = persist_prepare<NW, LP>();
#ifdef SPORCO_AMD_HOSTSIM
    hostsim::set_coop(grid);
#endif
    hipLaunchKernelGGL((admm_persist_kernel<NW, LP>), dim3((unsigned)grid), dim3(NW * 64), lds, st, a);
    SA_HIP(hipGetLastError());
}
// The grid barrier inside the one-launch solve needs every workgroup resident at once: what the
// device can hold of this kernel (registers, LDS) times its CUs must cover the grid -- asked of
// the runtime here, before the solve commits to the form, rather than found out by a barrier that
// times out.  (Work of OTHER streams or processes on the device can still starve it: the form is
// opt-in, for exclusive use of a device, and a timed-out barrier comes back as SPORCO_AMD_EHIP.)
template <> bool admm_persist_resident<float>(int H, int W, int K, int CN) {
#ifdef SPORCO_AMD_HOSTSIM
    (void)H; (void)W; (void)K; (void)CN;
    return true;
#else
    if (!admm_persist_supported<float>(H, W, K)) return false;
    const int grid = admm_persist_grid<float>(H, W, K, CN);
    int per_cu = 0;
    if (W == 128) {
        const size_t lds = persist_prepare<4, 4>();
        SA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, reinterpret_cast<const void *>(&admm_persist_kernel<4, 4>), 4 * 64, lds));
    } else {
        const size_t lds = persist_prepare<8, 2>();
        SA_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &per_cu, reinterpret_cast<const void *>(&admm_persist_kernel<8, 2>), 8 * 64, lds));
    }
    return (int64_t)per_cu * current_device_cus() >= grid;
#endif
}
template <> bool admm_persist_resident<double>(int, int, int, int) { return false; }

template <> void launch_admm_persist<float>(hipStream_t st, const AdmmPersistArgs<float> &a, int grid) {
    const RowsFwdArgs<float> &f = a.iter[0].fwd;
    SA_REQUIRE(admm_persist_supported<float>(f.H, f.W, f.K), "shape not handled by the one-launch solve");
    SA_REQUIRE(grid >= 8 && grid % 8 == 0 && grid <= std::max(8, persist_cus()), "grid of the one-launch solve");
    if (f.W == 128) launch_persist_inst<4, 4>(st, a, grid);
    else launch_persist_inst<8, 2>(st, a, grid);
}
template <> void launch_admm_persist<double>(hipStream_t, const AdmmPersistArgs<double> &, int) {
    throw Error(-1, "the one-launch solve is float32 only");
}

template <> void launch_rows_fwd<float>(hipStream_t st, const RowsFwdArgs<float> &a_in) {
    RowsFwdArgs<float> a = a_in;
    SA_REQUIRE(rows_supported<float>(a.W, a.K), "shape not handled by the fused row kernels");
    SA_REQUIRE(a.H <= 65535, "too many rows for one launch");
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<4>(&rows_fwd_kernel<4, false>);
        set_lds_attr<8>(&rows_fwd_kernel<8, false>);
        set_lds_attr<16>(&rows_fwd_kernel<16, false>);
        set_lds_attr<4>(&rows_fwd_kernel<4, true>);
        set_lds_attr<8>(&rows_fwd_kernel<8, true>);
        set_lds_attr<16>(&rows_fwd_kernel<16, true>);
        set_lds_attr<4>(&rows_fwd_kernel<4, false, true>);
        set_lds_attr<8>(&rows_fwd_kernel<8, false, true>);
        set_lds_attr<16>(&rows_fwd_kernel<16, false, true>);
        set_lds_attr<4>(&rows_fwd_kernel<4, false, true, true>);
        set_lds_attr<8>(&rows_fwd_kernel<8, false, true, true>);
        set_lds_attr<16>(&rows_fwd_kernel<16, false, true, true>);
    }
    SA_REQUIRE(!(a.v && a.y_bcast), "the broadcast row pass has no V form");
    if (rows_mr_width(a.W)) {
        launch_rows_fwd_mr(st, a);
        return;
    }
    if (a.v && !(a.flags & F_JOINT) &&
        (a.wl1.ptr || (a.flags & F_NOBNDRY) || a.ams_bits)) {
        // the V form under an L1Weight array / NoBndryCross / AddMaskSim
        static PerDeviceOnce gattr;
        if (gattr.first()) {
            set_lds_attr<4>(&rows_fwd_kernel<4, false, true, false, 1>);
            set_lds_attr<8>(&rows_fwd_kernel<8, false, true, false, 1>);
            set_lds_attr<16>(&rows_fwd_kernel<16, false, true, false, 1>);
            set_lds_attr<4>(&rows_fwd_kernel<4, false, true, false, 2>);
            set_lds_attr<8>(&rows_fwd_kernel<8, false, true, false, 2>);
            set_lds_attr<16>(&rows_fwd_kernel<16, false, true, false, 2>);
        }
        SA_REQUIRE(a.C * a.N == a.CN, "the derivation needs the channel / image split");
        const dim3 grid = rows_grid(a, a.W / kN1, ceil_div(a.P, 128), a.H, 0);
        const bool m1 = a.wl1.ptr != nullptr;
        if (a.W == 128) {
            if (m1) hipLaunchKernelGGL((rows_fwd_kernel<4, false, true, false, 1>), grid, dim3(4 * 64), rows_lds_bytes(4), st, a);
            else hipLaunchKernelGGL((rows_fwd_kernel<4, false, true, false, 2>), grid, dim3(4 * 64), rows_lds_bytes(4), st, a);
        } else if (a.W == 256) {
            if (m1) hipLaunchKernelGGL((rows_fwd_kernel<8, false, true, false, 1>), grid, dim3(8 * 64), rows_lds_bytes(8), st, a);
            else hipLaunchKernelGGL((rows_fwd_kernel<8, false, true, false, 2>), grid, dim3(8 * 64), rows_lds_bytes(8), st, a);
        } else {
            if (m1) hipLaunchKernelGGL((rows_fwd_kernel<16, false, true, false, 1>), grid, dim3(16 * 64), rows_lds_bytes(16), st, a);
            else hipLaunchKernelGGL((rows_fwd_kernel<16, false, true, false, 2>), grid, dim3(16 * 64), rows_lds_bytes(16), st, a);
        }
        SA_HIP(hipGetLastError());
        return;
    }
    if (a.v && (a.flags & F_JOINT)) {
        // the V form of ConvBPDNJoint: tiles as the joint epilogue (one image, 32 filters, all
        // channels per workgroup)
        SA_REQUIRE(rows_joint_supported<float>(a.W, a.C, a.K) && a.C * a.N == a.CN,
                   "configuration not handled by the joint row pass");
        const dim3 jgrid = rows_grid(a, a.W / kN1, (int64_t)a.N * (a.K / 32), a.H, 0);
        if (a.W == 128) hipLaunchKernelGGL((rows_fwd_kernel<4, false, true, true>), jgrid, dim3(4 * 64), rows_lds_bytes(4), st, a);
        else if (a.W == 256) hipLaunchKernelGGL((rows_fwd_kernel<8, false, true, true>), jgrid, dim3(8 * 64), rows_lds_bytes(8), st, a);
        else hipLaunchKernelGGL((rows_fwd_kernel<16, false, true, true>), jgrid, dim3(16 * 64), rows_lds_bytes(16), st, a);
        SA_HIP(hipGetLastError());
        return;
    }
    // (measured at config 2: the tile loop gains nothing for this kernel)
    const dim3 grid = rows_grid(a, a.W / kN1, ceil_div(a.P, 128), a.H, 0);
    if (a.W == 128) {
        if (a.y_bcast) hipLaunchKernelGGL((rows_fwd_kernel<4, true>), grid, dim3(4 * 64), rows_lds_bytes(4), st, a);
        else if (a.v) hipLaunchKernelGGL((rows_fwd_kernel<4, false, true>), grid, dim3(4 * 64), rows_lds_bytes(4), st, a);
        else hipLaunchKernelGGL((rows_fwd_kernel<4, false>), grid, dim3(4 * 64), rows_lds_bytes(4), st, a);
    } else if (a.W == 256) {
        if (a.y_bcast) hipLaunchKernelGGL((rows_fwd_kernel<8, true>), grid, dim3(8 * 64), rows_lds_bytes(8), st, a);
        else if (a.v) hipLaunchKernelGGL((rows_fwd_kernel<8, false, true>), grid, dim3(8 * 64), rows_lds_bytes(8), st, a);
        else hipLaunchKernelGGL((rows_fwd_kernel<8, false>), grid, dim3(8 * 64), rows_lds_bytes(8), st, a);
    } else {
        if (a.y_bcast) hipLaunchKernelGGL((rows_fwd_kernel<16, true>), grid, dim3(16 * 64), rows_lds_bytes(16), st, a);
        else if (a.v) hipLaunchKernelGGL((rows_fwd_kernel<16, false, true>), grid, dim3(16 * 64), rows_lds_bytes(16), st, a);
        else hipLaunchKernelGGL((rows_fwd_kernel<16, false>), grid, dim3(16 * 64), rows_lds_bytes(16), st, a);
    }
    SA_HIP(hipGetLastError());
}
template <> void launch_rows_fwd<double>(hipStream_t, const RowsFwdArgs<double> &) {
    throw Error(-1, "the fused row kernels are float32 only");
}

template <int NW, bool EMIT>
static void launch_post_nw(hipStream_t st, const RowsPostArgs<float> &a, dim3 grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 0, EMIT>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, true, 0, EMIT>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 1, EMIT>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, true, 1, EMIT>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 2, EMIT>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, true, 2, EMIT>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 0, EMIT, false, 1>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 0, EMIT, false, 2>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 1, EMIT, false, 1>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 1, EMIT, false, 2>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 2, EMIT, false, 1>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 2, EMIT, false, 2>);
    }
    const int mode = a.wl1.ptr != nullptr ? 1 : (((a.flags & F_NOBNDRY) || a.ams_bits) ? 2 : 0);
    const dim3 block(NW * 64);
    const size_t lds = rows_lds_bytes(NW);
    if (a.v_out) {      // single-array state (csc_rows.h)
        SA_REQUIRE(!a.x, "the V form has no X output");
        if (mode == 1) {
            if (a.v_in) hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 1, EMIT, false, 2>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 1, EMIT, false, 1>), grid, block, lds, st, a);
        } else if (mode == 2) {
            if (a.v_in) hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 2, EMIT, false, 2>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 2, EMIT, false, 1>), grid, block, lds, st, a);
        } else {
            if (a.v_in) hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 0, EMIT, false, 2>), grid, block, lds, st, a);
            else hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 0, EMIT, false, 1>), grid, block, lds, st, a);
        }
        return;
    }
    SA_REQUIRE(!a.v_in, "a V-form input needs a V-form output");
    if (a.x) {
        if (mode == 1) hipLaunchKernelGGL((rows_inv_post_kernel<NW, true, 1, EMIT>), grid, block, lds, st, a);
        else if (mode == 2) hipLaunchKernelGGL((rows_inv_post_kernel<NW, true, 2, EMIT>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((rows_inv_post_kernel<NW, true, 0, EMIT>), grid, block, lds, st, a);
    } else {
        if (mode == 1) hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 1, EMIT>), grid, block, lds, st, a);
        else if (mode == 2) hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 2, EMIT>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 0, EMIT>), grid, block, lds, st, a);
    }
}

// ams_bits[(h * CN + cn) * NW + w], bit n1 = (mask(h, NW n1 + w, c, n) != 0)
__global__ void __launch_bounds__(256) ams_pack_kernel(Weight<float> m, uint32_t *bits, int H, int W,
                                                       int C, int N, int NW) {
    const int64_t total = (int64_t)H * C * N * NW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % NW);
        const int cn = (int)((i / NW) % (C * N));
        const int h = (int)(i / ((int64_t)NW * C * N));
        const int c = cn / N, n = cn % N;
        uint32_t b = 0u;
        for (int n1 = 0; n1 < W / NW; ++n1) {
            const int x = NW * n1 + w;
            const float v = m.ptr[h * m.stride[0] + x * m.stride[1] + c * m.stride[2] + n * m.stride[3]];
            if (v != 0.f) b |= 1u << n1;
        }
        bits[i] = b;
    }
}

template <> void launch_ams_pack<float>(hipStream_t st, const Weight<float> &mask, uint32_t *bits,
                                        int H, int W, int C, int N) {
    const int NW = rows_mr_width(W) ? 16 : W / kN1;     // (waves of the row kernels: one word per wave)
    const int64_t total = (int64_t)H * C * N * NW;
    hipLaunchKernelGGL(ams_pack_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(total, 256), 65535)),
                       dim3(256), 0, st, mask, bits, H, W, C, N, NW);
    SA_HIP(hipGetLastError());
}
template <> void launch_ams_pack<double>(hipStream_t, const Weight<double> &, uint32_t *, int, int,
                                         int, int) {
    throw Error(-1, "the fused row kernels are float32 only");
}


template <int NW, bool EMIT>
static void launch_post_joint_nw(hipStream_t st, const RowsPostArgs<float> &a, dim3 grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 0, EMIT, true>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 0, EMIT, true, 1>);
        set_lds_attr<NW>(&rows_inv_post_kernel<NW, false, 0, EMIT, true, 2>);
    }
    if (a.v_out && a.v_in)
        hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 0, EMIT, true, 2>), grid, dim3(NW * 64),
                           rows_lds_bytes(NW), st, a);
    else if (a.v_out)
        hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 0, EMIT, true, 1>), grid, dim3(NW * 64),
                           rows_lds_bytes(NW), st, a);
    else
        hipLaunchKernelGGL((rows_inv_post_kernel<NW, false, 0, EMIT, true>), grid, dim3(NW * 64),
                           rows_lds_bytes(NW), st, a);
}

template <> int64_t launch_rows_inv_post<float>(hipStream_t st, const RowsPostArgs<float> &a_in) {
    RowsPostArgs<float> a = a_in;
    SA_REQUIRE(rows_supported<float>(a.W, a.K), "shape not handled by the fused row kernels");
    SA_REQUIRE(a.H <= 65535, "too many rows for one launch");
    if (rows_mr_width(a.W)) return launch_rows_inv_post_mr(st, a);
    if (a.flags & F_JOINT) {
        SA_REQUIRE(!a.v_in || a.v_out, "a V-form input needs a V-form output");
        SA_REQUIRE(rows_joint_supported<float>(a.W, a.C, a.K) && !a.wl1.ptr && !a.ams_bits &&
                       !(a.flags & F_NOBNDRY) && !a.x,
                   "configuration not handled by the joint row epilogue");
        const int64_t jtx = (int64_t)a.N * (a.K / 32);
        const dim3 jgrid = rows_grid(a, a.W / kN1, jtx, a.H, a.t_next != nullptr);
        if (a.W == 128) {
            if (a.t_next) launch_post_joint_nw<4, true>(st, a, jgrid);
            else launch_post_joint_nw<4, false>(st, a, jgrid);
        } else if (a.W == 256) {
            if (a.t_next) launch_post_joint_nw<8, true>(st, a, jgrid);
            else launch_post_joint_nw<8, false>(st, a, jgrid);
        } else {
            if (a.t_next) launch_post_joint_nw<16, true>(st, a, jgrid);
            else launch_post_joint_nw<16, false>(st, a, jgrid);
        }
        SA_HIP(hipGetLastError());
        return jtx * a.H;
    }
    const int64_t tx = ceil_div(a.P, 128);
    // persistent for the emitting variant (2.39 -> 2.21 ms at config 2: a tile's spectrum stores
    // drain under the next tile's loads); the plain epilogue is faster one tile per workgroup
    const dim3 grid = rows_grid(a, a.W / kN1, tx, a.H, a.t_next != nullptr);
    if (a.W == 128) {
        if (a.t_next) launch_post_nw<4, true>(st, a, grid);
        else launch_post_nw<4, false>(st, a, grid);
    } else if (a.W == 256) {
        if (a.t_next) launch_post_nw<8, true>(st, a, grid);
        else launch_post_nw<8, false>(st, a, grid);
    } else {
        if (a.t_next) launch_post_nw<16, true>(st, a, grid);
        else launch_post_nw<16, false>(st, a, grid);
    }
    SA_HIP(hipGetLastError());
    return tx * a.H;
}

template <int NW>
static void launch_prox_nw(hipStream_t st, const RowsProxArgs<float> &a_in, dim3 grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<NW>(&rows_inv_prox_fwd_kernel<NW, false>);
        set_lds_attr<NW>(&rows_inv_prox_fwd_kernel<NW, true>);
    }
    // (a negative threshold -- a negative lambda: meaningless, but defined -- takes the variant
    // whose soft threshold makes no assumption about its sign)
    const bool general = a_in.wl1.ptr != nullptr || (a_in.flags & F_NOBNDRY) || a_in.thr < 0.f;
    RowsProxArgs<float> a = a_in;
    if (general && !a.wl1.ptr) a.wl1.ptr = device_one();
    const dim3 block(NW * 64);
    if (general)
        hipLaunchKernelGGL((rows_inv_prox_fwd_kernel<NW, true>), grid, block, rows_lds_bytes(NW), st, a);
    else
        hipLaunchKernelGGL((rows_inv_prox_fwd_kernel<NW, false>), grid, block, rows_lds_bytes(NW), st, a);
}

template <> int64_t launch_rows_inv_prox_fwd<float>(hipStream_t st, const RowsProxArgs<float> &a) {
    SA_REQUIRE(rows_supported<float>(a.W, a.K), "shape not handled by the fused row kernels");
    SA_REQUIRE(a.H <= 65535, "too many rows for one launch");
    // (the forward half of this kernel is the emitting epilogue's, which gained 8 % from a
    // persistent launch; this one does not -- config 4 244.7-245.8 it/s persistent against
    // 240.7-246.2 per tile, profiles/r03g_config4_prox_persist.jsonl: one tile per workgroup)
    if (rows_mr_width(a.W)) return launch_rows_inv_prox_fwd_mr(st, a);
    RowsProxArgs<float> ap = a;
    const int64_t tx = ceil_div(a.P, 128);
    const dim3 grid = rows_grid(ap, a.W / kN1, tx, a.H, 0);
    if (a.W == 128)
        launch_prox_nw<4>(st, ap, grid);
    else if (a.W == 256)
        launch_prox_nw<8>(st, ap, grid);
    else
        launch_prox_nw<16>(st, ap, grid);
    SA_HIP(hipGetLastError());
    return tx * a.H;
}
template <> int64_t launch_rows_inv_prox_fwd<double>(hipStream_t, const RowsProxArgs<double> &) {
    throw Error(-1, "the fused row kernels are float32 only");
}

template <> int64_t launch_rows_inv_post<double>(hipStream_t, const RowsPostArgs<double> &) {
    throw Error(-1, "the fused row kernels are float32 only");
}

#endif   // SA_ROWS_MR_TU

}  // namespace sporco_amd
