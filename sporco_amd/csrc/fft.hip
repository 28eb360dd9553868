// fft.hip -- generic-length batched line FFT kernels (Stockham autosort in LDS).
//
// One kernel template serves every transform on the path:
//   MODE_C2C  complex lines (the H axis of rfftn/irfftn),
//   MODE_R2C  real -> half spectrum (the W axis of rfftn, sporco/fft.py:257-286),
//   MODE_C2R  half spectrum -> real (the W axis of irfftn, sporco/fft.py:288-314).
// For real transforms two ADJACENT batch columns (p, p+1) are packed into one
// complex line z = x_p + i x_{p+1}; with the filter-fastest layout that pair is a
// single aligned 8-byte (f32) load, and the two half spectra are separated
// ("untangled") on the way out.  Odd batch counts use one real column per
// complex line.
//
// A workgroup holds `cols` adjacent columns x all n points in LDS
// ([i][col], col fastest => conflict-free b64 accesses and 128-byte global
// segments), runs the radix passes ping-pong between two LDS buffers and
// streams the result out.  Radices 8/4/2 use in-register butterflies; any
// other factor (3, 5, 7, 17, ...) goes through a direct O(R) per-output pass,
// so every length that fits LDS is supported.
#include "fft.h"

#include "csc_post_elem.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

namespace sporco_amd {

enum { MODE_C2C = 0, MODE_R2C = 1, MODE_C2R = 2 };

template <typename T> struct LineArgs {
    const void *in;
    const void *in2;
    void *out;
    const cx<T> *tw;
    T s2, scale;
    int n, nfreq, cols, inverse, nrad;
    int radix[kMaxRadixPasses];
    int64_t ncols;
    int64_t in_outer, in_line, out_outer, out_line;
    // R2C only: `in` is broadcast over column blocks of bc_mod columns (column c reads
    // in[o * bc_outer + i * bc_line + c % bc_mod]); 0 = plain.  The consensus dictionary
    // update transforms Y[.., k] - s U[.., n, k] this way (admm/ccmod.py:768).
    int64_t bc_mod, bc_outer, bc_line;
    // Column groups on the complex side of a real transform (0 = plain): column p
    // lives at (p / grp) * grp_stride + p % grp instead of p.  This is how the row
    // transforms write / read the tile-major layout of csc_fused.h.
    int64_t grp, grp_stride;
    // C2R fused with the ADMM epilogue in the single-array state (FUSE > 0; fft.h fft_c2r_vpost): the
    // transform's output is X of element (o, i, p) of an (n_outer, n, P) array and never leaves the
    // workgroup; post.v_in / post.v_out carry the iterate (vthr / vnonneg below derive Y, U from
    // it), every workgroup writes 8 partial sums, and with FUSE == 2 the line Y' - U' is transformed
    // forward again and stored as the next iteration's row spectrum (emit_*).
    PostParams<T> post;
    double *partials;
    int64_t postP;
    cx<T> *emit_out;
    int64_t emit_outer, emit_line;
    // R2C only: `in` is the ADMM iterate in its single-array form V = AX + U (csc_rows.h): the line
    // transformed is Y - s2 U with Y = prox_l1(V; vthr) (+ NonNegCoef), U = V - Y, derived per
    // element as the epilogue derived them (csc_post_elem.h admm_post_elem)
    int vform;
    T vthr;
    int vnonneg;
};

template <typename T> __device__ __forceinline__ T yu_from_v(const LineArgs<T> &a, T v) {
    T y = soft(v, a.vthr);
    if (a.vnonneg && y < T(0)) y = T(0);
    const T u = v - y;
    return y - a.s2 * u;
}

template <typename T> __device__ __forceinline__ int64_t col_off(const LineArgs<T> &a, int64_t p) {
    return a.grp ? (p / a.grp) * a.grp_stride + p % a.grp : p;
}

// two adjacent complex values moved as one vector access
template <typename T> struct alignas(2 * sizeof(cx<T>)) cx2 {
    cx<T> a, b;
};

// ---------------------------------------------------------------------------
// in-register butterflies; INV selects exp(+2 pi i ...) kernels
// ---------------------------------------------------------------------------
template <typename T, bool INV> __device__ __forceinline__ cx<T> quarter(cx<T> a) {
    // multiply by W4 = exp(-/+ i pi/2)
    return INV ? mul_pi(a) : mul_mi(a);
}

template <typename T, bool INV>
__device__ __forceinline__ void dft4(cx<T> &a0, cx<T> &a1, cx<T> &a2, cx<T> &a3) {
    const cx<T> t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = quarter<T, INV>(a1 - a3);
    a0 = t0 + t2;
    a1 = t1 + t3;
    a2 = t0 - t2;
    a3 = t1 - t3;
}

template <typename T, int R, bool INV> struct Butterfly;

template <typename T, bool INV> struct Butterfly<T, 2, INV> {
    static __device__ __forceinline__ void run(cx<T> (&v)[2]) {
        const cx<T> a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <typename T, bool INV> struct Butterfly<T, 4, INV> {
    static __device__ __forceinline__ void run(cx<T> (&v)[4]) { dft4<T, INV>(v[0], v[1], v[2], v[3]); }
};

template <typename T, bool INV> struct Butterfly<T, 8, INV> {
    static __device__ __forceinline__ void run(cx<T> (&v)[8]) {
        cx<T> e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        cx<T> o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
        dft4<T, INV>(e0, e1, e2, e3);
        dft4<T, INV>(o0, o1, o2, o3);
        const T h = T(0.70710678118654752440);
        // W8^1 = (1 -/+ i)/sqrt2, W8^2 = -/+ i, W8^3 = (-1 -/+ i)/sqrt2
        const cx<T> w1 = INV ? mk<T>(h * (o1.re - o1.im), h * (o1.re + o1.im))
                             : mk<T>(h * (o1.re + o1.im), h * (o1.im - o1.re));
        const cx<T> w2 = quarter<T, INV>(o2);
        const cx<T> w3 = INV ? mk<T>(-h * (o3.re + o3.im), h * (o3.re - o3.im))
                             : mk<T>(h * (o3.im - o3.re), -h * (o3.re + o3.im));
        v[0] = e0 + o0;
        v[4] = e0 - o0;
        v[1] = e1 + w1;
        v[5] = e1 - w1;
        v[2] = e2 + w2;
        v[6] = e2 - w2;
        v[3] = e3 + w3;
        v[7] = e3 - w3;
    }
};

// Radix 3 and 5 (the factors of 240-, 320-, 360-, 384-, 480-point lines) with their constants in
// the instruction stream: out_q = sum_s v_s W_R^(s q), W_R = exp(-/+ 2 pi i / R).
template <typename T, bool INV> struct Butterfly<T, 3, INV> {
    static __device__ __forceinline__ void run(cx<T> (&v)[3]) {
        const T h = T(0.86602540378443864676);        // sin(2 pi / 3)
        const cx<T> t1 = v[1] + v[2];
        const cx<T> m1 = mk<T>(v[0].re - T(0.5) * t1.re, v[0].im - T(0.5) * t1.im);
        const cx<T> dq = quarter<T, INV>(cscale(v[1] - v[2], h));     // -/+ i h (v1 - v2)
        v[0] = v[0] + t1;
        v[1] = m1 + dq;
        v[2] = m1 - dq;
    }
};

template <typename T, bool INV> struct Butterfly<T, 5, INV> {
    static __device__ __forceinline__ void run(cx<T> (&v)[5]) {
        const T c1 = T(0.30901699437494742410), c2 = T(-0.80901699437494742410);   // cos(2 pi/5), cos(4 pi/5)
        const T s1 = T(0.95105651629515357212), s2 = T(0.58778525229247312917);    // sin(2 pi/5), sin(4 pi/5)
        const cx<T> t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
        const cx<T> a1 = mk<T>(v[0].re + c1 * t1.re + c2 * t2.re, v[0].im + c1 * t1.im + c2 * t2.im);
        const cx<T> a2 = mk<T>(v[0].re + c2 * t1.re + c1 * t2.re, v[0].im + c2 * t1.im + c1 * t2.im);
        const cx<T> b1 = quarter<T, INV>(mk<T>(s1 * t3.re + s2 * t4.re, s1 * t3.im + s2 * t4.im));
        const cx<T> b2 = quarter<T, INV>(mk<T>(s2 * t3.re - s1 * t4.re, s2 * t3.im - s1 * t4.im));
        v[0] = v[0] + t1 + t2;
        v[1] = a1 + b1;           // forward: a1 - i b1
        v[4] = a1 - b1;
        v[2] = a2 + b2;
        v[3] = a2 - b2;
    }
};

template <typename T, bool INV> struct Butterfly<T, 7, INV> {
    static __device__ __forceinline__ void run(cx<T> (&v)[7]) {
        // cos / sin of 2 pi m / 7, m = 1, 2, 3 (224-, 448-, 336-point lines)
        const T c1 = T(0.62348980185873353053), c2 = T(-0.22252093395631440429), c3 = T(-0.90096886790241912624);
        const T s1 = T(0.78183148246802980871), s2 = T(0.97492791218182360702), s3 = T(0.43388373911755812048);
        const cx<T> t1 = v[1] + v[6], t2 = v[2] + v[5], t3 = v[3] + v[4];
        const cx<T> d1 = v[1] - v[6], d2 = v[2] - v[5], d3 = v[3] - v[4];
        auto comb = [](cx<T> x0, T ca, cx<T> xa, T cb, cx<T> xb, T cc, cx<T> xc) {
            return mk<T>(x0.re + ca * xa.re + cb * xb.re + cc * xc.re, x0.im + ca * xa.im + cb * xb.im + cc * xc.im);
        };
        const cx<T> z = mk<T>(T(0), T(0));
        const cx<T> a1 = comb(v[0], c1, t1, c2, t2, c3, t3), a2 = comb(v[0], c2, t1, c3, t2, c1, t3),
                    a3 = comb(v[0], c3, t1, c1, t2, c2, t3);
        const cx<T> b1 = quarter<T, INV>(comb(z, s1, d1, s2, d2, s3, d3));
        const cx<T> b2 = quarter<T, INV>(comb(z, s2, d1, -s3, d2, -s1, d3));
        const cx<T> b3 = quarter<T, INV>(comb(z, s3, d1, -s1, d2, s2, d3));
        v[0] = v[0] + t1 + t2 + t3;
        v[1] = a1 + b1;
        v[6] = a1 - b1;
        v[2] = a2 + b2;
        v[5] = a2 - b2;
        v[3] = a3 + b3;
        v[4] = a3 - b3;
    }
};

// Composite radices with coprime factors, R = N1 N2 (6 = 2*3, 10 = 2*5, 12 = 4*3), by the
// prime-factor map: inputs taken at n = (N2 e2 n1 + N1 e1 n2) mod R with
// e2 = N2^-1 mod N1, e1 = N1^-1 mod N2, outputs delivered at k = (N2 k1 + N1 k2) mod R make
// W_R^(n k) = W_N1^(n1 k1) W_N2^(n2 k2): N1 transforms of length N2, then N2 of length N1, no
// twiddles in between.  One LDS round trip then covers two prime factors of the line length.
template <int A, int M> constexpr int inv_mod() {
    for (int x = 1; x < M; ++x)
        if ((A * x) % M == 1) return x;
    return 1;
}
template <typename T, int N1, int N2, bool INV> struct PfaButterfly {
    static __device__ __forceinline__ void run(cx<T> (&v)[N1 * N2]) {
        constexpr int R = N1 * N2;
        constexpr int C1 = N2 * inv_mod<N2 % N1, N1>(), C2 = N1 * inv_mod<N1 % N2, N2>();
        cx<T> a[N1][N2];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
#pragma unroll
            for (int n2 = 0; n2 < N2; ++n2) a[n1][n2] = v[(C1 * n1 + C2 * n2) % R];
            Butterfly<T, N2, INV>::run(a[n1]);
        }
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            cx<T> b[N1];
#pragma unroll
            for (int n1 = 0; n1 < N1; ++n1) b[n1] = a[n1][k2];
            Butterfly<T, N1, INV>::run(b);
#pragma unroll
            for (int k1 = 0; k1 < N1; ++k1) v[(N2 * k1 + N1 * k2) % R] = b[k1];
        }
    }
};
template <typename T, bool INV> struct Butterfly<T, 6, INV> : PfaButterfly<T, 2, 3, INV> {};
template <typename T, bool INV> struct Butterfly<T, 10, INV> : PfaButterfly<T, 2, 5, INV> {};
template <typename T, bool INV> struct Butterfly<T, 12, INV> : PfaButterfly<T, 4, 3, INV> {};

// One Stockham pass of radix R over the `cols` columns held by the workgroup.
//   read  src[j + r*n/R],  twiddle exp(-/+ 2 pi i k r /(Ns R)), k = j mod Ns,
//   write dst[(j div Ns) Ns R + k + q Ns]
template <typename T, int R, bool INV>
__device__ __forceinline__ void radix_pass(const cx<T> *__restrict__ src, cx<T> *__restrict__ dst,
                                           const cx<T> *__restrict__ tw, int n, int Ns, int cols,
                                           int col, int lane, int lpc) {
    const int nb = n / R;
    const int tmul = n / (Ns * R);
    for (int j = lane; j < nb; j += lpc) {
        const int k = j % Ns;
        const int j0 = (j - k) * R + k;
        const int ts = k * tmul;
        cx<T> v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const cx<T> x = src[(j + r * nb) * cols + col];
            v[r] = (r == 0 || Ns == 1) ? x : cmul(x, tw[r * ts]);
        }
        Butterfly<T, R, INV>::run(v);
#pragma unroll
        for (int q = 0; q < R; ++q) dst[(j0 + q * Ns) * cols + col] = v[q];
    }
}

// Direct pass for an arbitrary radix R: every output is an R-term sum.
template <typename T>
__device__ __forceinline__ void direct_pass(const cx<T> *__restrict__ src, cx<T> *__restrict__ dst,
                                            const cx<T> *__restrict__ tw, int n, int R, int Ns,
                                            int cols, int col, int lane, int lpc) {
    const int nb = n / R;
    const int tmul = n / (Ns * R);
    for (int item = lane; item < n; item += lpc) {
        const int j = item % nb, q = item / nb;
        const int k = j % Ns;
        const int j0 = (j - k) * R + k;
        int step = k * tmul + q * nb;
        if (step >= n) step -= n;
        int idx = 0;
        cx<T> acc = mk<T>(T(0), T(0));
        for (int r = 0; r < R; ++r) {
            acc = acc + cmul(src[(j + r * nb) * cols + col], tw[idx]);
            idx += step;
            if (idx >= n) idx -= n;
        }
        dst[(j0 + q * Ns) * cols + col] = acc;
    }
}

// (BIG: the kernels that also carry the 6-, 10- and 12-point butterflies -- they need more
// registers, so lines without those radices run the lean instantiation)
template <typename T, bool INV, bool BIG>
__device__ __forceinline__ void stockham_pass_r(int R, const cx<T> *src, cx<T> *dst, const cx<T> *tw, int n,
                                                int Ns, int cols, int col, int lane, int lpc) {
    if constexpr (BIG) {
        switch (R) {
        case 6: radix_pass<T, 6, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); return;
        case 10: radix_pass<T, 10, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); return;
        case 12: radix_pass<T, 12, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); return;
        default: break;
        }
    }
    switch (R) {
    case 8: radix_pass<T, 8, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); break;
    case 4: radix_pass<T, 4, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); break;
    case 2: radix_pass<T, 2, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); break;
    case 3: radix_pass<T, 3, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); break;
    case 5: radix_pass<T, 5, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); break;
    case 7: radix_pass<T, 7, INV>(src, dst, tw, n, Ns, cols, col, lane, lpc); break;
    default: direct_pass<T>(src, dst, tw, n, R, Ns, cols, col, lane, lpc); break;
    }
}

// FUSE (MODE_C2R with packed columns only): 0 = plain store; 1 = the ADMM epilogue of the
// single-array state on every output element instead (csc_post_elem.h; plain l1 term); 2 = ... and
// the forward transform of Y' - U' after it (the next iteration's fft_r2c for an unchanged rho).
template <typename T, int MODE, bool PACK, int FUSE = 0>
__global__ void __launch_bounds__(1024) fft_lines_kernel(const LineArgs<T> a) {
    const int n = a.n, cols = a.cols;
    cx<T> *buf0 = dyn_lds<cx<T>>();
    cx<T> *buf1 = buf0 + (size_t)n * cols;
    cx<T> *tw = buf1 + (size_t)n * cols;

    const int tid = threadIdx.x;
    const int col = tid % cols, lane = tid / cols, lpc = blockDim.x / cols;
    const int64_t c = (int64_t)blockIdx.x * cols + col;
    const int64_t o = blockIdx.y;
    const bool valid = c < a.ncols;
    const cx<T> zero = mk<T>(T(0), T(0));

    for (int t = tid; t < n; t += blockDim.x) {
        cx<T> w = a.tw[t];
        if (a.inverse) w.im = -w.im;
        tw[t] = w;
    }

    // ---------------- load ----------------
    if (MODE == MODE_C2C) {
        const cx<T> *in = static_cast<const cx<T> *>(a.in) + o * a.in_outer + c;
        for (int i = lane; i < n; i += lpc) buf0[i * cols + col] = valid ? in[i * a.in_line] : zero;
    } else if (MODE == MODE_R2C) {
        if (PACK) {
            const cx<T> *in = a.bc_mod
                                  ? static_cast<const cx<T> *>(a.in) + o * a.bc_outer + c % a.bc_mod
                                  : static_cast<const cx<T> *>(a.in) + o * a.in_outer + c;
            const int64_t in_ln = a.bc_mod ? a.bc_line : a.in_line;
            const cx<T> *in2 = a.in2 ? static_cast<const cx<T> *>(a.in2) + o * a.in_outer + c : nullptr;
            for (int i = lane; i < n; i += lpc) {
                cx<T> v = zero;
                if (valid) {
                    v = in[i * in_ln];
                    if (a.vform) v = mk<T>(yu_from_v(a, v.re), yu_from_v(a, v.im));
                    if (in2) {
                        const cx<T> u = in2[i * a.in_line];
                        v = mk<T>(v.re - a.s2 * u.re, v.im - a.s2 * u.im);
                    }
                }
                buf0[i * cols + col] = v;
            }
        } else {
            const T *in = a.bc_mod ? static_cast<const T *>(a.in) + o * a.bc_outer + c % a.bc_mod
                                   : static_cast<const T *>(a.in) + o * a.in_outer + c;
            const int64_t in_ln = a.bc_mod ? a.bc_line : a.in_line;
            const T *in2 = a.in2 ? static_cast<const T *>(a.in2) + o * a.in_outer + c : nullptr;
            for (int i = lane; i < n; i += lpc) {
                T v = T(0);
                if (valid) {
                    v = in[i * in_ln];
                    if (a.vform) v = yu_from_v(a, v);
                    if (in2) v -= a.s2 * in2[i * a.in_line];
                }
                buf0[i * cols + col] = mk<T>(v, T(0));
            }
        }
    } else {  // MODE_C2R: rebuild the full spectrum of the packed line
        const int nyq = (n % 2 == 0) ? n / 2 : -1;
        for (int f = lane; f < a.nfreq; f += lpc) {
            cx<T> A = zero, B = zero;
            if (valid) {
                if (PACK) {
                    const cx2<T> ab = *reinterpret_cast<const cx2<T> *>(
                        static_cast<const cx<T> *>(a.in) + o * a.in_outer + f * a.in_line +
                        col_off(a, 2 * c));
                    A = ab.a;
                    B = ab.b;
                } else {
                    A = (static_cast<const cx<T> *>(a.in) + o * a.in_outer + col_off(a, c))[f * a.in_line];
                }
            }
            if (f == 0 || f == nyq) {
                buf0[f * cols + col] = mk<T>(A.re, B.re);
            } else {
                buf0[f * cols + col] = mk<T>(A.re - B.im, A.im + B.re);
                buf0[(n - f) * cols + col] = mk<T>(A.re + B.im, B.re - A.im);
            }
        }
    }
    __syncthreads();

    // ---------------- radix passes ----------------
    cx<T> *src = buf0, *dst = buf1;
    int Ns = 1;
    for (int p = 0; p < a.nrad; ++p) {
        const int R = a.radix[p];
        if (a.inverse) {
            stockham_pass_r<T, true, false>(R, src, dst, tw, n, Ns, cols, col, lane, lpc);
        } else {
            stockham_pass_r<T, false, false>(R, src, dst, tw, n, Ns, cols, col, lane, lpc);
        }
        __syncthreads();
        cx<T> *t = src;
        src = dst;
        dst = t;
        Ns *= R;
    }

    if constexpr (FUSE != 0) {
        static_assert(MODE == MODE_C2R && PACK, "the fused epilogue belongs to the packed c2r pass");
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (valid) {
            const int64_t P = a.postP;
            for (int i = lane; i < n; i += lpc) {
                const cx<T> z = src[i * cols + col];
                const int64_t idx = (o * n + i) * P + 2 * c;
                const T x0 = z.re * a.scale, x1 = z.im * a.scale;
                // (Y, U) of the iterate from V, as admm_post_kernel derives them
                const cx<T> vv = *reinterpret_cast<const cx<T> *>(a.post.v_in + idx);
                T y0 = soft(vv.re, a.post.thr_prev), y1 = soft(vv.im, a.post.thr_prev);
                if ((a.post.flags & F_NONNEG) && y0 < T(0)) y0 = T(0);
                if ((a.post.flags & F_NONNEG) && y1 < T(0)) y1 = T(0);
                T u0 = vv.re - y0, u1 = vv.im - y1, vn0, vn1;
                admm_post_elem<T, false>(a.post, idx, P, x0, y0, u0, acc, &vn0);
                admm_post_elem<T, false>(a.post, idx + 1, P, x1, y1, u1, acc, &vn1);
                *reinterpret_cast<cx<T> *>(a.post.v_out + idx) = mk<T>(vn0, vn1);
                // (the line the next fft_r2c would form from V' with s2 = 1: Y' - U')
                if (FUSE == 2) src[i * cols + col] = mk<T>(y0 - u0, y1 - u1);
            }
        }
        if constexpr (FUSE == 2) {
            __syncthreads();
            for (int t = tid; t < n; t += blockDim.x) tw[t].im = -tw[t].im;     // forward twiddles
            __syncthreads();
            Ns = 1;
            for (int p = 0; p < a.nrad; ++p) {
                stockham_pass_r<T, false, false>(a.radix[p], src, dst, tw, n, Ns, cols, col, lane, lpc);
                __syncthreads();
                cx<T> *t = src;
                src = dst;
                dst = t;
                Ns *= a.radix[p];
            }
            if (valid) {
                for (int f = lane; f < a.nfreq; f += lpc) {
                    const cx<T> zf = src[f * cols + col];
                    const cx<T> zn = src[(f == 0 ? 0 : n - f) * cols + col];
                    cx2<T> ab;
                    ab.a = mk<T>(T(0.5) * (zf.re + zn.re), T(0.5) * (zf.im - zn.im));
                    ab.b = mk<T>(T(0.5) * (zf.im + zn.im), T(0.5) * (zn.re - zf.re));
                    *reinterpret_cast<cx2<T> *>(a.emit_out + o * a.emit_outer + f * a.emit_line + 2 * c) = ab;
                }
            }
        }
        __syncthreads();      // the transform buffers become the reduction scratch
        block_sum_store<8>(acc, reinterpret_cast<double *>(buf0),
                           a.partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8);
        return;
    }
    // ---------------- store ----------------
    if (!valid) return;
    if (MODE == MODE_C2C) {
        cx<T> *out = static_cast<cx<T> *>(a.out) + o * a.out_outer + c;
        for (int i = lane; i < n; i += lpc) out[i * a.out_line] = cscale(src[i * cols + col], a.scale);
    } else if (MODE == MODE_R2C) {
        for (int f = lane; f < a.nfreq; f += lpc) {
            const cx<T> zf = src[f * cols + col];
            if (PACK) {
                const cx<T> zn = src[(f == 0 ? 0 : n - f) * cols + col];
                cx2<T> ab;
                ab.a = mk<T>(T(0.5) * (zf.re + zn.re), T(0.5) * (zf.im - zn.im));
                ab.b = mk<T>(T(0.5) * (zf.im + zn.im), T(0.5) * (zn.re - zf.re));
                *reinterpret_cast<cx2<T> *>(static_cast<cx<T> *>(a.out) + o * a.out_outer +
                                            f * a.out_line + col_off(a, 2 * c)) = ab;
            } else {
                (static_cast<cx<T> *>(a.out) + o * a.out_outer + col_off(a, c))[f * a.out_line] = zf;
            }
        }
    } else {
        if (PACK) {
            cx<T> *out = static_cast<cx<T> *>(a.out) + o * a.out_outer + c;
            for (int i = lane; i < n; i += lpc) out[i * a.out_line] = cscale(src[i * cols + col], a.scale);
        } else {
            T *out = static_cast<T *>(a.out) + o * a.out_outer + c;
            for (int i = lane; i < n; i += lpc) out[i * a.out_line] = src[i * cols + col].re * a.scale;
        }
    }
}

// ---------------------------------------------------------------------------
// Column pass of the generic ADMM X-step in ONE kernel: forward transform along H, the
// Sherman-Morrison solve (linalg.solvedbi_sm, sporco/linalg.py:232-297) and the inverse
// transform, for one (wf, cn) tile of all n = H frequencies x K filters held in LDS -- two passes
// over the spectrum instead of the six of fft_c2c + launch_sm_solve + fft_c2c.  The tile has ONE
// buffer (a Stockham pass needs two): the forward transform is decimation in frequency in place
// (natural order in, digit-reversed order out: a radix-R butterfly reads and writes the same R
// positions, so a barrier between passes is all the synchronisation there is), the solve works
// at the digit-reversed positions (plan.drev maps a position to its frequency), and the inverse
// is the transposed flow (decimation in time: conj twiddle, then butterfly; passes in reverse
// order), which takes the digit-reversed order back to the natural one.
// ---------------------------------------------------------------------------
template <typename T> struct ColsSmArgs {
    cx<T> *xf;          // (n, Wf, CN, K): rfft_W(Y - s U) on entry, the column-inverse-transformed
                        // solution spectrum on return (the c2r row pass is what remains of irfftn)
    const cx<T> *df;    // (n, Wf, K)
    const cx<T> *sf;    // (n, Wf, CN)
    const T *gram;      // (n, Wf)
    const cx<T> *tw;    // W_n^t
    const int *drev;    // position -> frequency after the in-place forward transform
    T rho;
    int n, Wf, CN, K, Kp, W, nrad, want_obj;
    int Ks;             // cols_sm_slab_kernel: filters per slab (a power of two)
    int radix[kMaxRadixPasses];
    double *partials;   // one per tile: Parseval-weighted sum of |Df.xf - Sf|^2
};

// R-point DFT of v (exp(-/+ 2 pi i s q / R)), any R, from the table of W_n^t (n a multiple of R)
template <typename T, int R, bool INV>
__device__ __forceinline__ void small_dft(cx<T> (&v)[R], const cx<T> *tw, int n) {
    if constexpr (R == 2 || R == 3 || R == 4 || R == 5 || R == 6 || R == 7 || R == 8 || R == 10 || R == 12) {
        Butterfly<T, R, INV>::run(v);
    } else {
        const int step = n / R;
        cx<T> o[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            cx<T> acc = v[0];
#pragma unroll
            for (int s = 1; s < R; ++s) {
                const cx<T> w = tw[((s * q) % R) * step];
                acc = acc + (INV ? cmulc(w, v[s]) : cmul(w, v[s]));
            }
            o[q] = acc;
        }
#pragma unroll
        for (int q = 0; q < R; ++q) v[q] = o[q];
    }
}

// One in-place pass over blocks of length m: the R elements j + s m/R of every block.
// Forward (decimation in frequency): butterfly, then output q times W_m^(j q).
// Inverse (decimation in time):      input s times conj W_m^(j s), then the conjugate butterfly.
template <typename T, int R, bool INV>
__device__ __forceinline__ void inplace_pass(cx<T> *buf, const cx<T> *tw, int n, int m, int K, int ncol,
                                             int col, int lane, int lpc) {
    // (K: columns per row of the buffer; ncol <= K of them hold data -- a filter count below the
    // lane group, the last slab of a tile: the other lanes idle)
    const int sub = m / R, nb = n / R, tstep = n / m;
    if (col >= ncol) return;
    for (int b = lane; b < nb; b += lpc) {
        const int blk = b / sub, j = b - blk * sub;
        const int base = blk * m + j;
        cx<T> v[R];
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const cx<T> x = buf[(base + s * sub) * K + col];
            v[s] = (INV && s > 0 && j > 0) ? cmulc(tw[j * s * tstep], x) : x;
        }
        small_dft<T, R, INV>(v, tw, n);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const cx<T> x = (!INV && q > 0 && j > 0) ? cmul(tw[j * q * tstep], v[q]) : v[q];
            buf[(base + q * sub) * K + col] = x;
        }
    }
}

template <typename T, bool INV, bool BIG>
__device__ __forceinline__ void inplace_pass_r(int R, cx<T> *buf, const cx<T> *tw, int n, int m, int K,
                                               int ncol, int col, int lane, int lpc) {
    if constexpr (BIG) {
        switch (R) {
        case 6: inplace_pass<T, 6, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); return;
        case 10: inplace_pass<T, 10, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); return;
        case 12: inplace_pass<T, 12, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); return;
        default: break;
        }
    }
    switch (R) {
    case 8: inplace_pass<T, 8, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); break;
    case 4: inplace_pass<T, 4, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); break;
    case 2: inplace_pass<T, 2, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); break;
    case 3: inplace_pass<T, 3, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); break;
    case 5: inplace_pass<T, 5, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); break;
    default: inplace_pass<T, 7, INV>(buf, tw, n, m, K, ncol, col, lane, lpc); break;
    }
}

// US: rows of solve operands (Df, Sf, gram) a thread requests together; when that covers all
// its rows they are requested before the forward passes and arrive behind them.
template <typename T, int US, bool BIG = false>
__global__ void __launch_bounds__(1024) cols_sm_kernel(const ColsSmArgs<T> a) {
    const int n = a.n, K = a.K;
    cx<T> *buf = dyn_lds<cx<T>>();
    cx<T> *tw = buf + (size_t)n * K;
    double *scratch = reinterpret_cast<double *>(tw + n);
    int *drev = reinterpret_cast<int *>(scratch + 16);
    const int tid = threadIdx.x;
    // Kp = K rounded up to a power of two: the lanes of one row (the sum over the filters is a
    // butterfly over them); columns K .. Kp - 1 stay idle
    const int Kp = a.Kp;
    const int col = tid % Kp, lane = tid / Kp, lpc = blockDim.x / Kp;
    const bool cvalid = col < K;
    const int colc = cvalid ? col : 0;
    // Workgroup b runs on XCD b % 8 (observed dispatch order; used for speed only): every XCD
    // takes a contiguous run of tiles, so that the CN tiles that share a row frequency's slice
    // of Df find it in that XCD's L2.
    const int ntiles = a.Wf * a.CN, per_xcd = (ntiles + 7) / 8;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || tile >= ntiles) return;
    const int wf = tile / a.CN, cn = tile - wf * a.CN;
    const int64_t rowstride = (int64_t)a.Wf * a.CN * K;
    cx<T> *x = a.xf + ((int64_t)wf * a.CN + cn) * K + (cvalid ? col : 0);
    const cx<T> zero = mk<T>(T(0), T(0));
    for (int t = tid; t < n; t += blockDim.x) {
        tw[t] = a.tw[t];
        drev[t] = a.drev[t];
    }
    cx<T> d[US];
    // (Df of the thread's rows travels across the forward passes when one batch covers them; the
    // per-row scalars Sf and gram are requested when the solve starts -- holding them too spilt
    // the 12-row variant)
    auto load_d = [&](int pos0, const int *dr) {
#pragma unroll
        for (int u = 0; u < US; ++u) {
            const int pos = pos0 + u * lpc + lane;
            const int64_t pix = pos < n ? (int64_t)dr[pos] * a.Wf + wf : 0;
            // (an unconditional load from a clamped address, then a select: a predicated load
            // would split the batch into blocks and serialise the requests)
            const cx<T> dl = a.df[pix * K + colc];
            d[u] = cvalid ? dl : zero;
        }
    };
    const bool one_batch = US * lpc >= n;
    if (one_batch) load_d(0, a.drev);
    // (batches of independent loads: one row per thread in flight leaves the pass waiting out a
    // memory round trip per row)
    constexpr int UL = 8;
    for (int r0 = lane; r0 < n; r0 += UL * lpc) {
        cx<T> t[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int r = r0 + u * lpc;
            t[u] = r < n ? x[r * rowstride] : zero;       // (x points at column colc)
        }
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int r = r0 + u * lpc;
            if (r < n && cvalid) buf[r * K + col] = t[u];
        }
    }
    __syncthreads();
    // ---- forward, in place ------------------------------------------------------------------
    int m = n;
    for (int p = 0; p < a.nrad; ++p) {
        inplace_pass_r<T, false, BIG>(a.radix[p], buf, tw, n, m, K, K, col, lane, lpc);
        m /= a.radix[p];
        __syncthreads();
    }
    // ---- xf = yuf + conj(Df) (Sf - sum_k Df yuf) / (sum_k |Df|^2 + rho) ------------------------
    double acc[1] = {0.0};
    // (half-spectrum Parseval weights 1, 2, ..., 2, 1 or 2: fft.py:476-484)
    const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == a.Wf - 1)) ? 1.0 : 2.0;
    // (every lane of a wave takes part in the shuffles: the trip count is the workgroup's)
    for (int pos0 = 0; pos0 < n; pos0 += US * lpc) {
        if (!one_batch) load_d(pos0, drev);
        cx<T> sv[US];
        T gv[US];
#pragma unroll
        for (int u = 0; u < US; ++u) {
            const int pos = pos0 + u * lpc + lane;
            const int64_t pix = pos < n ? (int64_t)drev[pos] * a.Wf + wf : 0;
            sv[u] = a.sf[pix * a.CN + cn];
            gv[u] = a.gram[pix];
        }
#pragma unroll
        for (int u = 0; u < US; ++u) {
            const int pos = pos0 + u * lpc + lane;
            if (pos0 + u * lpc >= n) break;        // (uniform over the workgroup)
            const bool valid = pos < n && cvalid;
            const cx<T> yu = valid ? buf[pos * K + col] : zero;
            cx<T> q = cmul(d[u], yu);
            for (int s = Kp >> 1; s > 0; s >>= 1) {    // (the Kp <= 64 lanes of a row share pos)
                q.re += __shfl_xor(q.re, s, kWave);
                q.im += __shfl_xor(q.im, s, kWave);
            }
            const cx<T> coef = cscale(sv[u] - q, T(1) / (gv[u] + a.rho));
            if (valid) buf[pos * K + col] = yu + cmulc(d[u], coef);
            // Df.xf - Sf = rho (q - Sf) / (gram + rho)
            if (a.want_obj && valid && col == 0)
                acc[0] += pw * (double)cabs2(coef) * (double)a.rho * (double)a.rho;
        }
    }
    __syncthreads();
    // ---- inverse, in place: the transposed flow ---------------------------------------------------
    for (int p = a.nrad - 1; p >= 0; --p) {
        m *= a.radix[p];
        inplace_pass_r<T, true, BIG>(a.radix[p], buf, tw, n, m, K, K, col, lane, lpc);
        __syncthreads();
    }
    if (cvalid)
        for (int r = lane; r < n; r += lpc) x[r * rowstride] = buf[r * K + col];
    if (a.want_obj) block_sum_store<1>(acc, scratch, a.partials + tile);
}

// The same pass for a tile that does not fit LDS (384 x 64 float32, 256 x 64 float64, ...): the
// filters go through in slabs of Ks.  Phase A, per slab: load, forward transform, add the slab's
// share of sum_k Df yuf to q (n values in LDS), write the transformed slab back over xf.  Then
// the multiplier of every frequency from q.  Phase B, per slab: re-load the transformed slab (it
// was written moments ago by this CU: L2 / MALL), apply the solve, inverse transform, store.
// Four passes over the spectrum (the re-load mostly out of cache) instead of the six of the
// three kernels.
template <typename T, bool BIG = false>
__global__ void __launch_bounds__(1024) cols_sm_slab_kernel(const ColsSmArgs<T> a) {
    const int n = a.n, K = a.K, Ks = a.Ks;
    cx<T> *buf = dyn_lds<cx<T>>();
    cx<T> *tw = buf + (size_t)n * Ks;
    cx<T> *q = tw + n;
    double *scratch = reinterpret_cast<double *>(q + n);
    int *drev = reinterpret_cast<int *>(scratch + 16);
    const int tid = threadIdx.x;
    const int col = tid % Ks, lane = tid / Ks, lpc = blockDim.x / Ks;
    const int ntiles = a.Wf * a.CN, per_xcd = (ntiles + 7) / 8;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || tile >= ntiles) return;
    const int wf = tile / a.CN, cn = tile - wf * a.CN;
    const int64_t rowstride = (int64_t)a.Wf * a.CN * K;
    cx<T> *xt = a.xf + ((int64_t)wf * a.CN + cn) * K;
    const cx<T> zero = mk<T>(T(0), T(0));
    for (int t = tid; t < n; t += blockDim.x) {
        tw[t] = a.tw[t];
        drev[t] = a.drev[t];
        q[t] = zero;
    }
    constexpr int UL = 8;
    const int nslab = (K + Ks - 1) / Ks;
    auto load_slab = [&](int k0, bool cv) {
        for (int r0 = lane; r0 < n; r0 += UL * lpc) {
            cx<T> t[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int r = r0 + u * lpc;
                t[u] = (r < n && cv) ? xt[r * rowstride + k0 + col] : zero;
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int r = r0 + u * lpc;
                if (r < n && cv) buf[r * Ks + col] = t[u];
            }
        }
    };
    __syncthreads();
    // ---- phase A ----------------------------------------------------------------------------
    for (int s = 0; s < nslab; ++s) {
        const int k0 = s * Ks, kw = (K - k0) < Ks ? (K - k0) : Ks;
        const bool cv = col < kw;
        load_slab(k0, cv);
        __syncthreads();
        int m = n;
        for (int p = 0; p < a.nrad; ++p) {
            inplace_pass_r<T, false, BIG>(a.radix[p], buf, tw, n, m, Ks, kw, col, lane, lpc);
            m /= a.radix[p];
            __syncthreads();
        }
        constexpr int US = 4;
        for (int pos0 = 0; pos0 < n; pos0 += US * lpc) {
            cx<T> d[US];
#pragma unroll
            for (int u = 0; u < US; ++u) {
                const int pos = pos0 + u * lpc + lane;
                const int64_t pix = pos < n ? (int64_t)drev[pos] * a.Wf + wf : 0;
                d[u] = cv ? a.df[pix * K + k0 + col] : zero;
            }
#pragma unroll
            for (int u = 0; u < US; ++u) {
                const int pos = pos0 + u * lpc + lane;
                if (pos0 + u * lpc >= n) break;        // (uniform over the workgroup)
                const bool valid = pos < n && cv;
                const cx<T> v = valid ? buf[pos * Ks + col] : zero;
                cx<T> pq = cmul(d[u], v);
                for (int sh = Ks >> 1; sh > 0; sh >>= 1) {
                    pq.re += __shfl_xor(pq.re, sh, kWave);
                    pq.im += __shfl_xor(pq.im, sh, kWave);
                }
                if (pos < n && col == 0) q[pos] = q[pos] + pq;     // (one lane group per row)
                if (valid) xt[pos * rowstride + k0 + col] = v;     // the transformed slab, kept in xf
            }
        }
        __syncthreads();
    }
    // ---- the multiplier of every frequency: (Sf - sum_k Df yuf) / (sum_k |Df|^2 + rho) -----------
    double acc[1] = {0.0};
    const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == a.Wf - 1)) ? 1.0 : 2.0;
    for (int pos = tid; pos < n; pos += blockDim.x) {
        const int64_t pix = (int64_t)drev[pos] * a.Wf + wf;
        const cx<T> coef = cscale(a.sf[pix * a.CN + cn] - q[pos], T(1) / (a.gram[pix] + a.rho));
        q[pos] = coef;
        if (a.want_obj) acc[0] += pw * (double)cabs2(coef) * (double)a.rho * (double)a.rho;
    }
    __syncthreads();
    // ---- phase B ----------------------------------------------------------------------------
    for (int s = 0; s < nslab; ++s) {
        const int k0 = s * Ks, kw = (K - k0) < Ks ? (K - k0) : Ks;
        const bool cv = col < kw;
        constexpr int US = 4;
        for (int pos0 = lane; pos0 < n; pos0 += US * lpc) {
            cx<T> d[US], v[US];
#pragma unroll
            for (int u = 0; u < US; ++u) {
                const int pos = pos0 + u * lpc;
                const bool valid = pos < n && cv;
                const int64_t pix = pos < n ? (int64_t)drev[pos] * a.Wf + wf : 0;
                d[u] = valid ? a.df[pix * K + k0 + col] : zero;
                v[u] = valid ? xt[pos * rowstride + k0 + col] : zero;
            }
#pragma unroll
            for (int u = 0; u < US; ++u) {
                const int pos = pos0 + u * lpc;
                if (pos < n && cv) buf[pos * Ks + col] = v[u] + cmulc(d[u], q[pos]);
            }
        }
        __syncthreads();
        int m = 1;
        for (int p = a.nrad - 1; p >= 0; --p) {
            m *= a.radix[p];
            inplace_pass_r<T, true, BIG>(a.radix[p], buf, tw, n, m, Ks, kw, col, lane, lpc);
            __syncthreads();
        }
        if (cv)
            for (int r = lane; r < n; r += lpc) xt[r * rowstride + k0 + col] = buf[r * Ks + col];
        __syncthreads();
    }
    if (a.want_obj) block_sum_store<1>(acc, scratch, a.partials + tile);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
void FftPlan::init(int n_) {
    SA_REQUIRE(n_ >= 1, "FFT length must be >= 1");
    destroy();
    n = n_;
    nrad = 0;
    nrad_ip = 0;
    int m = n;
    {
        int e2 = 0, e3 = 0, e5 = 0, e7 = 0;
        while (m % 2 == 0) m /= 2, ++e2;
        while (m % 3 == 0) m /= 3, ++e3;
        while (m % 5 == 0) m /= 5, ++e5;
        while (m % 7 == 0) m /= 7, ++e7;
        auto push = [&](int r) {
            SA_REQUIRE(nrad < kMaxRadixPasses, "FFT length has too many factors");
            radix[nrad++] = r;
        };
        auto push_ip = [&](int r) {
            SA_REQUIRE(nrad_ip < kMaxRadixPasses, "FFT length has too many factors");
            radix_ip[nrad_ip++] = r;
        };
        {
            int a2 = e2;
            for (; a2 >= 3; a2 -= 3) push(8);
            for (; a2 >= 2; a2 -= 2) push(4);
            for (; a2 >= 1; --a2) push(2);
            for (int i = 0; i < e3; ++i) push(3);
            for (int i = 0; i < e5; ++i) push(5);
            for (int i = 0; i < e7; ++i) push(7);
        }
        // (the composite passes pay in the in-place kernel -- 10-20 % off the fused column pass at
        // 240 / 320 / 384 / 480 points -- and cost the Stockham line kernels registers and occupancy:
        // profiles/r04y9_fft_radices.jsonl)
        int a2 = e2, a3 = e3, a5 = e5;
        {
            while (a3 > 0 && a2 >= 2) push_ip(12), --a3, a2 -= 2;
            while (a3 > 0 && a2 >= 1) push_ip(6), --a3, --a2;
            while (a5 > 0 && a2 >= 1) push_ip(10), --a5, --a2;
        }
        for (; a2 >= 3; a2 -= 3) push_ip(8);
        for (; a2 >= 2; a2 -= 2) push_ip(4);
        for (; a2 >= 1; --a2) push_ip(2);
        for (; a3 > 0; --a3) push_ip(3);
        for (; a5 > 0; --a5) push_ip(5);
        for (int i = 0; i < e7; ++i) push_ip(7);
        std::sort(radix_ip, radix_ip + nrad_ip, [](int x, int y) { return x > y; });
    }
    for (int p = 11; m > 1; p += 2)
        while (m % p == 0) {
            SA_REQUIRE(nrad < kMaxRadixPasses && nrad_ip < kMaxRadixPasses, "FFT length has too many factors");
            radix[nrad++] = p;
            radix_ip[nrad_ip++] = p;
            m /= p;
        }
    std::vector<cx<float>> t32(n);
    std::vector<cx<double>> t64(n);
    const double two_pi = 6.283185307179586476925286766559;
    for (int t = 0; t < n; ++t) {
        const double ang = -two_pi * (double)t / (double)n;
        t64[t] = mk<double>(std::cos(ang), std::sin(ang));
        t32[t] = mk<float>((float)t64[t].re, (float)t64[t].im);
    }
    SA_HIP(hipMalloc((void **)&tw32, sizeof(cx<float>) * n));
    SA_HIP(hipMalloc((void **)&tw64, sizeof(cx<double>) * n));
    SA_HIP(hipMemcpy(tw32, t32.data(), sizeof(cx<float>) * n, hipMemcpyHostToDevice));
    SA_HIP(hipMemcpy(tw64, t64.data(), sizeof(cx<double>) * n, hipMemcpyHostToDevice));
    // position -> frequency of the in-place forward transform: position q0 n/r0 + q1 n/(r0 r1) +
    // ... holds frequency q0 + r0 (q1 + r1 (...))
    std::vector<int> dr(n);
    for (int pos = 0; pos < n; ++pos) {
        int rem = pos, f = 0, weight = 1, mcur = n;
        for (int p = 0; p < nrad_ip; ++p) {
            const int sub = mcur / radix_ip[p];
            f += (rem / sub) * weight;
            rem %= sub;
            weight *= radix_ip[p];
            mcur = sub;
        }
        dr[pos] = f;
    }
    SA_HIP(hipMalloc((void **)&drev, sizeof(int) * n));
    SA_HIP(hipMemcpy(drev, dr.data(), sizeof(int) * n, hipMemcpyHostToDevice));
}

void FftPlan::destroy() {
    if (tw32) (void)hipFree(tw32);
    if (tw64) (void)hipFree(tw64);
    if (drev) (void)hipFree(drev);
    drev = nullptr;
    tw32 = nullptr;
    tw64 = nullptr;
    n = 0;
    nrad = 0;
}

template <> const cx<float> *FftPlan::tw<float>() const { return tw32; }
template <> const cx<double> *FftPlan::tw<double>() const { return tw64; }

namespace {

constexpr size_t kLdsBudget = 160 * 1024;  // one workgroup may own the whole CU LDS

struct Cfg {
    int cols, threads;
    size_t lds;
};

template <typename T> Cfg pick_cfg(int n, int64_t ncols) {
    const size_t esz = sizeof(cx<T>);
    int cols = (int)(128 / esz);  // 128-byte rows: 16 columns f32, 8 columns f64
    while (cols > 1 && (2 * (size_t)n * cols + n) * esz > kLdsBudget) cols >>= 1;
    SA_REQUIRE((2 * (size_t)n * cols + n) * esz <= kLdsBudget,
               "transform length too large for the single-pass LDS FFT");
    while (cols > 1 && cols / 2 >= ncols) cols >>= 1;
    // float32 lines of 192 < n < 512 points: 64-byte rows (8 columns).  At 16 columns such a tile
    // takes 50-126 KiB of LDS and a CU holds one or two workgroups, whose load, transform and store
    // phases then have nothing to overlap with; at 8 columns it holds three to five.  Measured
    // (profiles/r04o_fft_cols.jsonl): the row and column passes of 240-, 320-, 384- and 480-point
    // lines 17-28 % faster; 512-point columns and the float64 lines slower, so they keep 128 bytes.
    if (sizeof(T) == 4 && n < 512 && cols == 16 && (2 * (size_t)n * cols + n) * esz > 48 * 1024) {
        cols = 8;
    }
    int lpc = 1;
    while (lpc * 8 < n && lpc * 2 * cols <= 1024) lpc <<= 1;
    int threads = cols * lpc;
    if (threads < kWave) threads = kWave;
    Cfg c;
    c.cols = cols;
    c.threads = threads;
    c.lds = (2 * (size_t)n * cols + n) * esz;
    return c;
}

template <typename T, int MODE, bool PACK, int FUSE = 0>
int64_t launch_lines(hipStream_t st, const FftPlan &plan, LineArgs<T> &a, int64_t n_outer) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fft_lines_kernel<T, MODE, PACK, FUSE>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
    }
    if (a.ncols <= 0 || n_outer <= 0) return 0;
    const Cfg cfg = pick_cfg<T>(plan.n, a.ncols);
    a.n = plan.n;
    a.nfreq = plan.n / 2 + 1;
    a.cols = cfg.cols;
    a.nrad = plan.nrad;
    for (int i = 0; i < plan.nrad; ++i) a.radix[i] = plan.radix[i];
    a.tw = plan.tw<T>();
    SA_REQUIRE(n_outer <= 65535, "too many outer slices for one launch");
    const dim3 grid((unsigned)ceil_div(a.ncols, cfg.cols), (unsigned)n_outer, 1);
    hipLaunchKernelGGL((fft_lines_kernel<T, MODE, PACK, FUSE>), grid, dim3(cfg.threads), cfg.lds, st, a);
    SA_HIP(hipGetLastError());
    return (int64_t)grid.x * grid.y;
}

}  // namespace

template <typename T>
void fft_c2c(hipStream_t st, const FftPlan &plan, bool inverse, const cx<T> *in, cx<T> *out,
             int64_t n_outer, int64_t ncols, int64_t in_outer, int64_t in_line,
             int64_t out_outer, int64_t out_line, T scale) {
    LineArgs<T> a{};
    a.in = in;
    a.in2 = nullptr;
    a.out = out;
    a.s2 = T(0);
    a.scale = scale;
    a.inverse = inverse ? 1 : 0;
    a.ncols = ncols;
    a.in_outer = in_outer;
    a.in_line = in_line;
    a.out_outer = out_outer;
    a.out_line = out_line;
    launch_lines<T, MODE_C2C, false>(st, plan, a, n_outer);
}

template <typename T>
void fft_r2c(hipStream_t st, const FftPlan &plan, const T *in, const T *in2, T s2, cx<T> *out,
             int64_t n_outer, int64_t P, int64_t in_outer, int64_t in_line, int64_t out_outer,
             int64_t out_line, int64_t grp, int64_t grp_stride, int64_t bc_mod, const VformIn<T> *vf) {
    LineArgs<T> a{};
    if (vf) {
        a.vform = 1;
        a.vthr = vf->thr;
        a.vnonneg = vf->nonneg ? 1 : 0;
    }
    // broadcast `in`: (n_outer, n, bc_mod) contiguous, i.e. line stride bc_mod
    a.bc_mod = bc_mod;
    a.bc_line = bc_mod;
    a.bc_outer = (int64_t)plan.n * bc_mod;
    a.grp = grp;
    a.grp_stride = grp_stride;
    a.in = in;
    a.in2 = in2;
    a.out = out;
    a.s2 = s2;
    a.scale = T(1);
    a.inverse = 0;
    a.out_outer = out_outer;
    a.out_line = out_line;
    const bool pack = (P % 2 == 0) && (in_outer % 2 == 0) && (in_line % 2 == 0) &&
                      (out_outer % 2 == 0) && (out_line % 2 == 0) && (grp % 2 == 0) &&
                      (grp_stride % 2 == 0) && (bc_mod % 2 == 0);
    if (pack) {
        a.bc_mod /= 2;
        a.bc_line /= 2;
        a.bc_outer /= 2;
        a.ncols = P / 2;
        a.in_outer = in_outer / 2;
        a.in_line = in_line / 2;
        launch_lines<T, MODE_R2C, true>(st, plan, a, n_outer);
    } else {
        a.ncols = P;
        a.in_outer = in_outer;
        a.in_line = in_line;
        launch_lines<T, MODE_R2C, false>(st, plan, a, n_outer);
    }
}

template <typename T>
void fft_c2r(hipStream_t st, const FftPlan &plan, const cx<T> *in, T *out, int64_t n_outer,
             int64_t P, int64_t in_outer, int64_t in_line, int64_t out_outer, int64_t out_line,
             T scale, int64_t grp, int64_t grp_stride) {
    LineArgs<T> a{};
    a.grp = grp;
    a.grp_stride = grp_stride;
    a.in = in;
    a.in2 = nullptr;
    a.out = out;
    a.s2 = T(0);
    a.scale = scale;
    a.inverse = 1;
    a.in_outer = in_outer;
    a.in_line = in_line;
    const bool pack = (P % 2 == 0) && (in_outer % 2 == 0) && (in_line % 2 == 0) &&
                      (out_outer % 2 == 0) && (out_line % 2 == 0) && (grp % 2 == 0) &&
                      (grp_stride % 2 == 0);
    if (pack) {
        a.ncols = P / 2;
        a.out_outer = out_outer / 2;
        a.out_line = out_line / 2;
        launch_lines<T, MODE_C2R, true>(st, plan, a, n_outer);
    } else {
        a.ncols = P;
        a.out_outer = out_outer;
        a.out_line = out_line;
        launch_lines<T, MODE_C2R, false>(st, plan, a, n_outer);
    }
}

template <typename T> bool fft_c2r_vpost_supported(int64_t P) { return P % 2 == 0; }

template <typename T>
int64_t fft_c2r_vpost_blocks(const FftPlan &plan, int64_t n_outer, int64_t P) {
    return ceil_div(P / 2, pick_cfg<T>(plan.n, P / 2).cols) * n_outer;
}

template <typename T>
int64_t fft_c2r_vpost(hipStream_t st, const FftPlan &plan, const cx<T> *in, int64_t n_outer, int64_t P,
                      int64_t in_outer, int64_t in_line, T scale, const PostParams<T> &post, cx<T> *emit_out,
                      int64_t emit_outer, int64_t emit_line, double *partials) {
    SA_REQUIRE(fft_c2r_vpost_supported<T>(P) && in_outer % 2 == 0 && in_line % 2 == 0,
               "the fused row pass needs an even number of columns");
    SA_REQUIRE(post.v_in && post.v_out && !(post.flags & (F_JOINT | F_NOBNDRY)) && !post.wl1.ptr && !post.ams.ptr,
               "the fused row pass serves the single-array state with a plain l1 term");
    LineArgs<T> a{};
    a.in = in;
    a.in2 = nullptr;
    a.out = nullptr;
    a.s2 = T(0);
    a.scale = scale;
    a.inverse = 1;
    a.in_outer = in_outer;
    a.in_line = in_line;
    a.post = post;
    a.partials = partials;
    a.postP = P;
    a.emit_out = emit_out;
    a.emit_outer = emit_outer;
    a.emit_line = emit_line;
    a.ncols = P / 2;
    return emit_out ? launch_lines<T, MODE_C2R, true, 2>(st, plan, a, n_outer)
                    : launch_lines<T, MODE_C2R, true, 1>(st, plan, a, n_outer);
}

template <typename T>
void rfft2(hipStream_t st, const FftPlan &planW, const FftPlan &planH, const T *in, const T *in2,
           T s2, cx<T> *out, int H, int W, int64_t P) {
    const int64_t Wf = W / 2 + 1;
    fft_r2c<T>(st, planW, in, in2, s2, out, H, P, (int64_t)W * P, P, Wf * P, P, 0, 0, 0);
    // columns: (wf, p) is one contiguous run of Wf*P complex columns per row h
    fft_c2c<T>(st, planH, false, out, out, 1, Wf * P, 0, Wf * P, 0, Wf * P, T(1));
}

template <typename T>
void irfft2(hipStream_t st, const FftPlan &planW, const FftPlan &planH, const cx<T> *in,
            cx<T> *tmp, T *out, int H, int W, int64_t P) {
    const int64_t Wf = W / 2 + 1;
    fft_c2c<T>(st, planH, true, in, tmp, 1, Wf * P, 0, Wf * P, 0, Wf * P, T(1));
    fft_c2r<T>(st, planW, tmp, out, H, P, Wf * P, P, (int64_t)W * P, P,
               T(1.0 / ((double)H * (double)W)), 0, 0);
}

template <typename T> static size_t cols_sm_lds(int n, int K) {
    return sizeof(cx<T>) * ((size_t)n * K + n) + sizeof(double) * 16 + sizeof(int) * n;
}

template <typename T> static size_t cols_sm_slab_lds(int n, int Ks) {
    return sizeof(cx<T>) * ((size_t)n * Ks + 2 * n) + sizeof(double) * 16 + sizeof(int) * n;
}
// filters per slab when the tile does not fit: the largest power of two that does (0: none >= 8)
// (force: the test switch of fft.h -- slabs of that width)
template <typename T> static int cols_sm_slab_width(int n, int K, int force) {
    if (force > 0) {
        const int ks = force;
        return (ks >= 2 && ks < K && !(ks & (ks - 1)) && cols_sm_slab_lds<T>(n, ks) <= kLdsBudget) ? ks : 0;
    }
    for (int ks = 64; ks >= 8; ks >>= 1)
        if (ks < K && cols_sm_slab_lds<T>(n, ks) <= kLdsBudget) return ks;
    return 0;
}

template <typename T> bool fft_cols_sm_supported(const FftPlan &plan, int K, int force_slab) {
    if (K < 2 || plan.n < 2) return false;
    for (int p = 0; p < plan.nrad_ip; ++p) {
        const int r = plan.radix_ip[p];
        if (r > 12 || r == 9 || r == 11) return false;
    }
    if (K <= 64 && cols_sm_lds<T>(plan.n, K) <= kLdsBudget && force_slab <= 0) return true;
    return K <= 256 && cols_sm_slab_width<T>(plan.n, K, force_slab) > 0;
}

template <typename T>
int64_t fft_cols_sm(hipStream_t st, const FftPlan &plan, cx<T> *xf, const cx<T> *df, const cx<T> *sf,
                    const T *gram, T rho, int Wf, int CN, int K, int W, bool want_obj, double *partials,
                    int force_slab) {
    SA_REQUIRE(fft_cols_sm_supported<T>(plan, K, force_slab), "fft_cols_sm: unsupported length / filter count");
    ColsSmArgs<T> a{};
    a.xf = xf;
    a.df = df;
    a.sf = sf;
    a.gram = gram;
    a.tw = plan.tw<T>();
    a.drev = plan.drev;
    a.rho = rho;
    a.n = plan.n;
    a.Wf = Wf;
    a.CN = CN;
    a.K = K;
    a.Kp = 2;
    while (a.Kp < K) a.Kp <<= 1;
    a.W = W;
    a.nrad = plan.nrad_ip;
    for (int i = 0; i < plan.nrad_ip; ++i) a.radix[i] = plan.radix_ip[i];
    a.want_obj = want_obj ? 1 : 0;
    a.partials = partials;
    bool big = false;       // (a 10-, 12-, 14- or 15-point pass: the instantiations that carry them)
    for (int i = 0; i < plan.nrad_ip; ++i) big = big || plan.radix_ip[i] == 6 || plan.radix_ip[i] >= 10;
    static PerDeviceOnce attr_set;
    if (!(K <= 64 && cols_sm_lds<T>(plan.n, K) <= kLdsBudget) || force_slab > 0) {
        // the tile goes through in slabs of filters
        a.Ks = cols_sm_slab_width<T>(plan.n, K, force_slab);
        static PerDeviceOnce slab_attr;
        if (slab_attr.first()) {
            SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cols_sm_slab_kernel<T, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
            SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cols_sm_slab_kernel<T, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        }
        const int64_t tiles = (int64_t)Wf * CN;
        const dim3 sgrid((unsigned)(8 * ((tiles + 7) / 8)));
        if (big)
            hipLaunchKernelGGL((cols_sm_slab_kernel<T, true>), sgrid, dim3(1024), cols_sm_slab_lds<T>(plan.n, a.Ks),
                               st, a);
        else
            hipLaunchKernelGGL((cols_sm_slab_kernel<T, false>), sgrid, dim3(1024), cols_sm_slab_lds<T>(plan.n, a.Ks),
                               st, a);
        SA_HIP(hipGetLastError());
        return tiles;
    }
    const size_t lds = cols_sm_lds<T>(plan.n, K);
    const int threads = (int64_t)plan.n * K >= 4096 ? 1024 : 256;
    if (attr_set.first()) {
        constexpr int UM = sizeof(T) == 8 ? 6 : 12;
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cols_sm_kernel<T, UM / 3>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cols_sm_kernel<T, 2 * UM / 3>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cols_sm_kernel<T, UM>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
        SA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&cols_sm_kernel<T, UM / 3, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBudget));
    }
    const int64_t tiles = (int64_t)Wf * CN;
    const unsigned grid = (unsigned)(8 * ((tiles + 7) / 8));
    // rows per thread: all operands in one batch while the registers allow it
    const int nit = (int)ceil_div(plan.n, threads / a.Kp);
    constexpr int UMAX = sizeof(T) == 8 ? 6 : 12;
    if (big)       // (the wide butterflies leave no registers for a long operand batch)
        hipLaunchKernelGGL((cols_sm_kernel<T, UMAX / 3, true>), dim3(grid), dim3(threads), lds, st, a);
    else if (nit <= UMAX / 3)
        hipLaunchKernelGGL((cols_sm_kernel<T, UMAX / 3>), dim3(grid), dim3(threads), lds, st, a);
    else if (nit <= 2 * UMAX / 3 || nit > UMAX)
        hipLaunchKernelGGL((cols_sm_kernel<T, 2 * UMAX / 3>), dim3(grid), dim3(threads), lds, st, a);
    else
        hipLaunchKernelGGL((cols_sm_kernel<T, UMAX>), dim3(grid), dim3(threads), lds, st, a);
    SA_HIP(hipGetLastError());
    return tiles;
}

#define SA_INSTANTIATE(T)                                                                        \
    template void fft_c2c<T>(hipStream_t, const FftPlan &, bool, const cx<T> *, cx<T> *, int64_t, \
                             int64_t, int64_t, int64_t, int64_t, int64_t, T);                    \
    template void fft_r2c<T>(hipStream_t, const FftPlan &, const T *, const T *, T, cx<T> *,     \
                             int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,      \
                             int64_t, int64_t, const VformIn<T> *);                              \
    template void fft_c2r<T>(hipStream_t, const FftPlan &, const cx<T> *, T *, int64_t, int64_t,  \
                             int64_t, int64_t, int64_t, int64_t, T, int64_t, int64_t);           \
    template bool fft_c2r_vpost_supported<T>(int64_t);                                           \
    template int64_t fft_c2r_vpost_blocks<T>(const FftPlan &, int64_t, int64_t);                 \
    template int64_t fft_c2r_vpost<T>(hipStream_t, const FftPlan &, const cx<T> *, int64_t, int64_t, int64_t, \
                                      int64_t, T, const PostParams<T> &, cx<T> *, int64_t, int64_t, double *); \
    template bool fft_cols_sm_supported<T>(const FftPlan &, int, int);                          \
    template int64_t fft_cols_sm<T>(hipStream_t, const FftPlan &, cx<T> *, const cx<T> *,        \
                                    const cx<T> *, const T *, T, int, int, int, int, bool, double *, int); \
    template void rfft2<T>(hipStream_t, const FftPlan &, const FftPlan &, const T *, const T *,  \
                           T, cx<T> *, int, int, int64_t);                                       \
    template void irfft2<T>(hipStream_t, const FftPlan &, const FftPlan &, const cx<T> *,        \
                            cx<T> *, T *, int, int, int64_t);
SA_INSTANTIATE(float)
SA_INSTANTIATE(double)

}  // namespace sporco_amd
