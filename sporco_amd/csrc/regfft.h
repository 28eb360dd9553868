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

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
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
