// regfft.h -- device building blocks of the register-resident transforms
// (csc_fused.hip, csc_rows.hip): fully unrolled radix-2 FFTs on per-thread
// arrays, the transposing wave reduction, buffer addressing, register fences.
// Forward transforms are decimation-in-frequency (natural in, bit-reversed
// out), inverse ones decimation-in-time (bit-reversed in, natural out), so no
// reordering pass exists anywhere.
#pragma once

#include <gfx950_intrin.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace sporco_amd {
namespace regfft {

typedef cx<float> cf;

struct alignas(8) f2 {
    float x, y;
};

// cos(2 pi t / 64), t = 0..16
__host__ __device__ constexpr double cos64_q(int t) {
    constexpr double c[17] = {1.0,
                              0.99518472667219688624,
                              0.98078528040323044913,
                              0.95694033573220886494,
                              0.92387953251128675613,
                              0.88192126434835502971,
                              0.83146961230254523708,
                              0.77301045336273696081,
                              0.70710678118654752440,
                              0.63439328416364549822,
                              0.55557023301960222474,
                              0.47139673682599764856,
                              0.38268343236508977173,
                              0.29028467725446236764,
                              0.19509032201612826785,
                              0.09801714032956060199,
                              0.0};
    return c[t];
}
__host__ __device__ constexpr double cos64(int t) {
    t = ((t % 64) + 64) % 64;
    if (t > 32) t = 64 - t;
    return t <= 16 ? cos64_q(t) : -cos64_q(32 - t);
}
__host__ __device__ constexpr double sin64(int t) { return cos64(t - 16); }

__host__ __device__ constexpr int brev(int x, int bits) {
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((x >> b) & 1) << (bits - 1 - b);
    return r;
}
__host__ __device__ constexpr int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

// d * exp(-/+ 2 pi i t / 64), t in [0, 32), t known at compile time after unrolling
template <bool INV> __device__ __forceinline__ cf tw64_mul(cf d, int t) {
    const float h = 0.70710678118654752440f;
    if (t == 0) return d;
    if (t == 16) return INV ? mul_pi(d) : mul_mi(d);
    if (t == 8)
        return INV ? mk<float>(h * (d.re - d.im), h * (d.re + d.im))
                   : mk<float>(h * (d.re + d.im), h * (d.im - d.re));
    if (t == 24)
        return INV ? mk<float>(-h * (d.re + d.im), h * (d.re - d.im))
                   : mk<float>(h * (d.im - d.re), -h * (d.re + d.im));
    // (fused multiply-adds named, see common.h fma1)
    const float c = (float)cos64(t), s = (float)sin64(t);
    return INV ? mk<float>(fma1(d.re, c, -(d.im * s)), fma1(d.im, c, d.re * s))
               : mk<float>(fma1(d.re, c, d.im * s), fma1(d.im, c, -(d.re * s)));
}

// Decimation in frequency: natural-order input v[off .. off+N), output X[brev(i)] at v[off+i].
template <int N, bool INV, int TOT> __device__ __forceinline__ void dif(cf (&v)[TOT], int off) {
#pragma unroll
    for (int len = N; len >= 2; len >>= 1) {
#pragma unroll
        for (int blk = 0; blk < N; blk += len) {
#pragma unroll
            for (int j = 0; j < len / 2; ++j) {
                const cf x = v[off + blk + j], y = v[off + blk + j + len / 2];
                v[off + blk + j] = x + y;
                v[off + blk + j + len / 2] = tw64_mul<INV>(x - y, j * (64 / len));
            }
        }
    }
}

// Decimation in time: input x[brev(i)] at v[off+i], natural-order output.
template <int N, bool INV, int TOT> __device__ __forceinline__ void dit(cf (&v)[TOT], int off) {
#pragma unroll
    for (int len = 2; len <= N; len <<= 1) {
#pragma unroll
        for (int blk = 0; blk < N; blk += len) {
#pragma unroll
            for (int j = 0; j < len / 2; ++j) {
                const cf x = v[off + blk + j];
                const cf y = tw64_mul<INV>(v[off + blk + j + len / 2], j * (64 / len));
                v[off + blk + j] = x + y;
                v[off + blk + j + len / 2] = x - y;
            }
        }
    }
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// ---------------------------------------------------------------------------------------------
// Mixed-radix lengths (round 6): the in-register transform of N1 = 20, 24, 28 or 30 points, so
// that lines of 16 N1 = 320, 384, 448, 480 points run on the register-resident kernels (the
// second factor stays 16 = the waves of the workgroup).  Same conventions as the radix-2 pair
// above: the forward transform is decimation in frequency -- natural in, DIGIT-reversed out:
// position i holds X[mr_rev<N>(i)] --, the inverse one decimation in time from that order back to
// natural; no reordering pass.  Radices are applied odd ones first (20 = 5.2.2, 24 = 3.2.2.2,
// 28 = 7.2.2, 30 = 5.3.2), which leaves the fewest non-trivial twiddles; every twiddle is a
// compile-time constant of the instruction stream.
// ---------------------------------------------------------------------------------------------
constexpr double kPiD = 3.14159265358979323846264338327950288;
constexpr double mr_tsin(double x) {     // |x| <= pi / 4
    double x2 = x * x, term = x, sum = x;
    for (int n = 1; n <= 11; ++n) {
        term *= -x2 / (double)((2 * n) * (2 * n + 1));
        sum += term;
    }
    return sum;
}
constexpr double mr_tcos(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int n = 1; n <= 11; ++n) {
        term *= -x2 / (double)((2 * n - 1) * (2 * n));
        sum += term;
    }
    return sum;
}
// cos / sin of 2 pi num / den (exact at the multiples of a quarter turn)
constexpr double mr_cos(int num, int den) {
    const int a = ((num % den) + den) % den;
    const int q = (8 * a + den) / (2 * den);                    // nearest quarter turn
    const double phi = (kPiD / 2.0) * (double)(4 * a - q * den) / (double)den;
    const int qq = q & 3;
    return qq == 0 ? mr_tcos(phi) : qq == 1 ? -mr_tsin(phi) : qq == 2 ? -mr_tcos(phi) : mr_tsin(phi);
}
constexpr double mr_sin(int num, int den) { return mr_cos(4 * num - den, 4 * den); }

// The supported lengths (one list for the kernels' instantiations, the launchers' switches and the
// host tables): X(n) for every n.  16 n = 160, 192, 224, 240, 288, 320, 336, 384, 400, 432, 448, 480.
#define SA_MR_LENGTHS(X) X(10) X(12) X(14) X(15) X(18) X(20) X(21) X(24) X(25) X(27) X(28) X(30)
constexpr bool mr_length(int n) {
#define SA_MR_IS(m) if (n == m) return true;
    SA_MR_LENGTHS(SA_MR_IS)
#undef SA_MR_IS
    return false;
}
// the radices of a length, in the order the forward transform applies them: odd ones first
constexpr int mr_radix_at(int n, int s) {
    const int rs[4] = {7, 5, 3, 2};
    for (int i = 0; i < 4; ++i)
        while (n % rs[i] == 0) {
            if (s == 0) return rs[i];
            --s;
            n /= rs[i];
        }
    return 1;
}
constexpr int mr_stages(int n) {
    int s = 0;
    while (mr_radix_at(n, s) != 1) ++s;
    return s;
}
template <int N> struct MrPlan {
    static constexpr int S = mr_stages(N);
    static constexpr int r(int s) { return mr_radix_at(N, s); }
};

// position i of the forward transform's output -> the frequency it holds (digits of i, most
// significant first, are the output indices p_1, p_2, ... of the stages; k = p_1 + r_1 p_2 + ...)
template <int N> constexpr int mr_rev(int i) {
    int m = N, k = 0, w = 1;
    for (int s = 0; s < MrPlan<N>::S; ++s) {
        const int r = MrPlan<N>::r(s);
        m /= r;
        k += (i / m) * w;
        i %= m;
        w *= r;
    }
    return k;
}
template <int N> constexpr int mr_pos(int k) {      // ... and the position that holds frequency k
    for (int i = 0; i < N; ++i)
        if (mr_rev<N>(i) == k) return i;
    return -1;
}
// one table for every in-register length: rev / pos of the power-of-two lengths are the bit reversal
template <int N> constexpr int rev1(int i) {
    if constexpr (mr_length(N)) return mr_rev<N>(i);
    else return brev(i, ilog2(N));
}
template <int N> constexpr int pos1(int k) {
    if constexpr (mr_length(N)) return mr_pos<N>(k);
    else return brev(k, ilog2(N));
}

// d * exp(-/+ 2 pi i T / LEN); the constants are evaluated by the compiler (constexpr variables:
// left to the optimiser, the Taylor loops of mr_cos survived into the 28-point kernels)
template <bool INV, int TT, int LEN> __device__ __forceinline__ cf mr_tw(cf d) {
    constexpr int t = TT % LEN;
    if constexpr (t == 0) return d;
    else if constexpr (4 * t == LEN) return INV ? mul_pi(d) : mul_mi(d);
    else if constexpr (2 * t == LEN) return mk<float>(-d.re, -d.im);
    else if constexpr (4 * t == 3 * LEN) return INV ? mul_mi(d) : mul_pi(d);
    else {
        constexpr float c = (float)mr_cos(t, LEN), s = (float)mr_sin(t, LEN);
        return INV ? mk<float>(fma1(d.re, c, -(d.im * s)), fma1(d.im, c, d.re * s))
                   : mk<float>(fma1(d.re, c, d.im * s), fma1(d.im, c, -(d.re * s)));
    }
}
// x -> -/+ i x (the quarter turn of the odd-radix butterflies: forward -i, inverse +i)
template <bool INV> __device__ __forceinline__ cf mr_quarter(cf a) { return INV ? mul_pi(a) : mul_mi(a); }

// out_q = sum_s v_s W_R^(s q), W_R = exp(-/+ 2 pi i / R), in place on R values (fused
// multiply-adds named: these files are compiled without contraction)
template <int R, bool INV> struct MrBfly;
template <bool INV> struct MrBfly<2, INV> {
    static __device__ __forceinline__ void run(cf (&v)[2]) {
        const cf a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};
template <bool INV> struct MrBfly<3, INV> {
    static __device__ __forceinline__ void run(cf (&v)[3]) {
        const float h = 0.86602540378443864676f;        // sin(2 pi / 3)
        const cf t1 = v[1] + v[2], d = v[1] - v[2];
        const cf m1 = mk<float>(fma1(-0.5f, t1.re, v[0].re), fma1(-0.5f, t1.im, v[0].im));
        const cf dq = mr_quarter<INV>(mk<float>(h * d.re, h * d.im));
        v[0] = v[0] + t1;
        v[1] = m1 + dq;
        v[2] = m1 - dq;
    }
};
template <bool INV> struct MrBfly<5, INV> {
    static __device__ __forceinline__ void run(cf (&v)[5]) {
        const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
        const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
        const cf t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
        const cf a1 = mk<float>(fma1(c2, t2.re, fma1(c1, t1.re, v[0].re)), fma1(c2, t2.im, fma1(c1, t1.im, v[0].im)));
        const cf a2 = mk<float>(fma1(c1, t2.re, fma1(c2, t1.re, v[0].re)), fma1(c1, t2.im, fma1(c2, t1.im, v[0].im)));
        const cf b1 = mr_quarter<INV>(mk<float>(fma1(s2, t4.re, s1 * t3.re), fma1(s2, t4.im, s1 * t3.im)));
        const cf b2 = mr_quarter<INV>(mk<float>(fma1(-s1, t4.re, s2 * t3.re), fma1(-s1, t4.im, s2 * t3.im)));
        v[0] = v[0] + t1 + t2;
        v[1] = a1 + b1;
        v[4] = a1 - b1;
        v[2] = a2 + b2;
        v[3] = a2 - b2;
    }
};
template <bool INV> struct MrBfly<7, INV> {
    static __device__ __forceinline__ cf comb(cf x0, float ca, cf xa, float cb, cf xb, float cc, cf xc) {
        return mk<float>(fma1(cc, xc.re, fma1(cb, xb.re, fma1(ca, xa.re, x0.re))),
                         fma1(cc, xc.im, fma1(cb, xb.im, fma1(ca, xa.im, x0.im))));
    }
    static __device__ __forceinline__ void run(cf (&v)[7]) {
        const float c1 = 0.62348980185873353053f, c2 = -0.22252093395631440429f, c3 = -0.90096886790241912624f;
        const float s1 = 0.78183148246802980871f, s2 = 0.97492791218182360702f, s3 = 0.43388373911755812048f;
        const cf t1 = v[1] + v[6], t2 = v[2] + v[5], t3 = v[3] + v[4];
        const cf d1 = v[1] - v[6], d2 = v[2] - v[5], d3 = v[3] - v[4];
        const cf z = mk<float>(0.f, 0.f);
        const cf a1 = comb(v[0], c1, t1, c2, t2, c3, t3), a2 = comb(v[0], c2, t1, c3, t2, c1, t3),
                 a3 = comb(v[0], c3, t1, c1, t2, c2, t3);
        const cf b1 = mr_quarter<INV>(comb(z, s1, d1, s2, d2, s3, d3));
        const cf b2 = mr_quarter<INV>(comb(z, s2, d1, -s3, d2, -s1, d3));
        const cf b3 = mr_quarter<INV>(comb(z, s3, d1, -s1, d2, s2, d3));
        v[0] = v[0] + t1 + t2 + t3;
        v[1] = a1 + b1;
        v[6] = a1 - b1;
        v[2] = a2 + b2;
        v[5] = a2 - b2;
        v[3] = a3 + b3;
        v[4] = a3 - b3;
    }
};

// one stage of the forward (DIF) flow on every block of length LEN = R * M of v[off .. off + N):
// butterfly over the stride-M elements, then the twiddle W_LEN^(j p) on output p of column j
template <int N, int LEN, int R, bool INV, int TOT>
__device__ __forceinline__ void mr_dif_stage(cf (&v)[TOT], int off) {
    constexpr int M = LEN / R;
    static_for<N / LEN>([&](auto bc) {
        constexpr int blk = decltype(bc)::value * LEN;
        static_for<M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            cf x[R];
#pragma unroll
            for (int q = 0; q < R; ++q) x[q] = v[off + blk + j + q * M];
            MrBfly<R, INV>::run(x);
            static_for<R>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                v[off + blk + j + p * M] = mr_tw<INV, j * p, LEN>(x[p]);
            });
        });
    });
}
// ... and of the inverse (DIT) flow: the conjugate twiddle first, then the butterfly
template <int N, int LEN, int R, bool INV, int TOT>
__device__ __forceinline__ void mr_dit_stage(cf (&v)[TOT], int off) {
    constexpr int M = LEN / R;
    static_for<N / LEN>([&](auto bc) {
        constexpr int blk = decltype(bc)::value * LEN;
        static_for<M>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            cf x[R];
            static_for<R>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                x[p] = mr_tw<INV, j * p, LEN>(v[off + blk + j + p * M]);
            });
            MrBfly<R, INV>::run(x);
#pragma unroll
            for (int q = 0; q < R; ++q) v[off + blk + j + q * M] = x[q];
        });
    });
}
template <int N, int S, int LEN, bool INV, int TOT>
__device__ __forceinline__ void mr_dif_from(cf (&v)[TOT], int off) {
    if constexpr (S < MrPlan<N>::S) {
        constexpr int R = MrPlan<N>::r(S);
        mr_dif_stage<N, LEN, R, INV>(v, off);
        mr_dif_from<N, S + 1, LEN / R, INV>(v, off);
    }
}
template <int N, int S, int LEN, bool INV, int TOT>
__device__ __forceinline__ void mr_dit_from(cf (&v)[TOT], int off) {
    // stage S of the forward plan works on blocks of LEN; the inverse runs the stages last to first
    if constexpr (S < MrPlan<N>::S) {
        constexpr int R = MrPlan<N>::r(S);
        mr_dit_from<N, S + 1, LEN / R, INV>(v, off);
        mr_dit_stage<N, LEN, R, INV>(v, off);
    }
}
// The in-register transform of any supported length: the radix-2 pair for powers of two (the
// instruction streams of rounds 1 ... 5, unchanged), the mixed-radix flow otherwise.
template <int N, bool INV, int TOT> __device__ __forceinline__ void dif1(cf (&v)[TOT], int off) {
    if constexpr (mr_length(N)) mr_dif_from<N, 0, N, INV>(v, off);
    else dif<N, INV>(v, off);
}
template <int N, bool INV, int TOT> __device__ __forceinline__ void dit1(cf (&v)[TOT], int off) {
    if constexpr (mr_length(N)) mr_dit_from<N, 0, N, INV>(v, off);
    else dit<N, INV>(v, off);
}

// Sum r[i] over the 64 lanes of the wave for all 8 i at once ("transposing"
// reduction: each of the first three exchanges halves the number of live
// values); lane l returns the total of r[l >> 3].  Everything stays on the VALU:
// permlane swaps across the 32- and 16-lane halves, DPP inside a 16-lane row
// (row_mirror pairs l with l^15, row_half_mirror with l^7: any pairing that
// flips the selecting bit works, and {15, 7, 2, 1} generate all 16 lanes).
__device__ __forceinline__ float reduce8_across_lanes(const float (&r)[8], int lane) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = r[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        sa_swap32(a[i], a[i + 4]);
        a[i] += a[i + 4];          // lanes < 32: sum of r[i]; lanes >= 32: sum of r[i + 4]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        sa_swap16(a[i], a[i + 2]);
        a[i] += a[i + 2];          // even rows: r[i] (or r[i+4]); odd rows: r[i+2] (or r[i+6])
    }
    const bool up = lane & 8;
    const float keep = up ? a[1] : a[0], send = up ? a[0] : a[1];
    float t = keep + sa_lane_xor15(send);
    t += sa_lane_xor7(t);
    t += sa_lane_xor2(t);
    t += sa_lane_xor1(t);
    return t;
}

// v summed over the four 16-lane rows of the wave, in every lane (two VALU permlane swaps, no
// LDS traffic): with both operands of a swap set to v, the swapped pair holds v and v of the
// lane 16 (32) away in SOME order in every lane, so their sum needs no select.
__device__ __forceinline__ float sum_over_rows(float v) {
    float a = v, b = v;
    sa_swap16(a, b);
    const float t = a + b;      // v[l] + v[l ^ 16]
    float c = t, d = t;
    sa_swap32(c, d);
    return c + d;               // ... + the same of the lane 32 away
}

// The same for TWO values at once: q0 and q1 each summed over the four rows, each total in every
// lane, in 7 instructions instead of 2 x 8 (a transposing reduction, then the transposition
// undone): the first swap leaves q0's partial sums in the even rows and q1's in the odd ones of
// ONE register, the second completes both, the third spreads them back.  Same association as
// sum_over_rows -- (r0 + r1) + (r2 + r3) -- so the results are bit-identical to it.
__device__ __forceinline__ void sum_over_rows2(float &q0, float &q1) {
    sa_swap16(q0, q1);          // q0 = [q0.r0, q1.r0, q0.r2, q1.r2], q1 = [q0.r1, q1.r1, q0.r3, q1.r3]
    const float t = q0 + q1;    // rows [Q0 r0+r1, Q1 r0+r1, Q0 r2+r3, Q1 r2+r3]
    float c = t, d = t;
    sa_swap32(c, d);            // c = [t0, t1, t0, t1], d = [t2, t3, t2, t3]
    const float s = c + d;      // rows [Q0, Q1, Q0, Q1]
    float x = s, y = s;
    sa_swap16(x, y);            // x = [Q0, Q0, Q0, Q0], y = [Q1, Q1, Q1, Q1]
    q0 = x;
    q1 = y;
}

typedef SaBuf BufRsrc;
__device__ __forceinline__ BufRsrc make_rsrc(const void *base, uint32_t bytes) {
    return sa_make_buf(base, bytes);
}
__device__ __forceinline__ cf buf_load_cf(BufRsrc r, int voff, int soff) {
    cf x;
    sa_buf_load2(r, voff, soff, x.re, x.im);
    return x;
}
// for operands that are re-read by other workgroups (Df): default cache policy
__device__ __forceinline__ cf buf_load_cf_cached(BufRsrc r, int voff, int soff) {
    cf x;
    sa_buf_load2_cached(r, voff, soff, x.re, x.im);
    return x;
}
__device__ __forceinline__ void buf_store_cf(BufRsrc r, int voff, int soff, cf x) {
    sa_buf_store2(r, voff, soff, x.re, x.im);
}
// COH: data that other workgroups of the SAME launch wrote / will read (gfx950_intrin.h)
template <bool COH> __device__ __forceinline__ cf buf_load_cf_x(BufRsrc r, int voff, int soff) {
    cf x;
    if constexpr (COH) sa_buf_load2_coh(r, voff, soff, x.re, x.im);
    else sa_buf_load2(r, voff, soff, x.re, x.im);
    return x;
}
template <bool COH> __device__ __forceinline__ void buf_store_cf_x(BufRsrc r, int voff, int soff, cf x) {
    if constexpr (COH) sa_buf_store2_coh(r, voff, soff, x.re, x.im);
    else sa_buf_store2(r, voff, soff, x.re, x.im);
}

// Register fence: every element passes through an (empty) volatile asm, and a
// token chained through all of them and back makes everything after the fence
// depend on everything before it.  No instruction is emitted; it only stops the
// scheduler from overlapping two stages of the unrolled transform, which is what
// drives its register demand far above the tile itself.
template <int N, int TOT> __device__ __forceinline__ void reg_fence(cf (&v)[TOT], int off, int &token) {
#pragma unroll
    for (int i = 0; i < N; ++i) SA_VGPR_FENCE3(v[off + i].re, v[off + i].im, token);
#pragma unroll
    for (int i = 0; i < N; ++i) SA_VGPR_FENCE3(v[off + i].re, v[off + i].im, token);
}


}  // namespace regfft
}  // namespace sporco_amd
