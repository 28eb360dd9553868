// csc_fused_mc.hip -- the register-resident column pass for multi-channel dictionaries.
//
// Same structure as fused_cols_kernel (csc_fused.hip): one workgroup per (wf, n) tile of the
// coefficient spectrum, column FFT -> per-frequency solve -> column IFFT in registers, lane =
// filter.  The system of sporco/admm/cbpdn.py:277-279 is (rho I + sum_c a_c a_c^H) x = b with
// a_c = conj(Df[c, :]), b = sum_c conj(Df_c) Sf_c + rho yuf.  The reference solves it by
// iterated Sherman-Morrison (linalg.solvemdbi_ism); written with the C x C matrix
//     B = (rho I_C + Df Df^H)^-1            (Df as a C x K matrix; B depends on Df, rho only)
// the same solution is (Woodbury)
//     x = yuf + Df^H g,      g = B (Sf - Df yuf),      Df x - Sf = -rho g,
// i.e. per frequency C inner products over the filters (one transposing wave reduction of
// 2C <= 8 values), a C x C matrix-vector product on wave-uniform values, and C rank-one
// updates.  B comes from launch_mc_binv (recomputed when rho or the dictionary changes).
#include "csc_fused.h"

#include "regfft.h"

#include <type_traits>
#include <utility>

namespace sporco_amd {

namespace {

using namespace regfft;

constexpr size_t mc_lds_bytes(int NW, int LP) {
    return sizeof(f2) * LP * NW * NW * 64 + sizeof(double) * 16;
}

template <int N1, int NW, int LPARAM, int KC, int CC>
__global__ void __launch_bounds__(NW * 64) fused_cols_mc_kernel(const FusedMcArgs<float> a) {
    constexpr int H = N1 * NW;
    constexpr int J = N1 / NW;
    constexpr int LB1 = ilog2(N1), LBW = ilog2(NW);
    constexpr int LP = LPARAM;
    constexpr int FP = LP * NW, Q = J / LP;
    static_assert(CC >= 2 && CC <= 4, "2 to 4 dictionary channels");
    const int tid = threadIdx.x;
    const int k = tid & 63;
    const int w = sa_readfirstlane(tid >> 6);
    const int K = KC ? KC : a.K;
    const bool kv = KC == 64 ? true : k < K;
    const int Wf = a.W / 2 + 1;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wf = (slot / a.N) * 8 + xcd;
    if (wf >= Wf) return;
    const int n = slot % a.N;
    const int tile = wf * a.N + n;
    const BufRsrc Tb = make_rsrc(a.t + (int64_t)tile * H * K, (uint32_t)(H * K * sizeof(cf)));
    const BufRsrc Db = make_rsrc(a.dft + (int64_t)wf * H * CC * K, (uint32_t)(H * CC * K * sizeof(cf)));
    const int ko = (w * K + k) * (int)sizeof(cf);              // row h = w of the tile
    const int dko = (w * CC * K + k) * (int)sizeof(cf);        // row f = w, channel 0 of Df
    const cf *S = a.sft + ((int64_t)wf * H + w) * CC * a.N + n;        // [f][c][n]
    const float *B = a.bt + ((int64_t)wf * H + w) * 2 * CC * CC;      // [f][c][c'] complex
    const cf *twA = a.twA + w * N1;
    const cf *twB = a.twB + w * N1;
    f2 *LA = dyn_lds<f2>();
    double *scratch = reinterpret_cast<double *>(LA + FP * NW * 64);
    const cf zero = mk<float>(0.f, 0.f);
    int token = 0;

    cf v[N1];
#pragma unroll
    for (int h1 = 0; h1 < N1; ++h1)
        v[h1] = kv ? buf_load_cf(Tb, ko, NW * h1 * K * (int)sizeof(cf)) : zero;
    dif<N1, false>(v, 0);
    reg_fence<N1>(v, 0, token);
#pragma unroll
    for (int i = 1; i < N1; ++i) v[i] = cmul(v[i], twA[i]);
    reg_fence<N1>(v, 0, token);

    float obj = 0.f;
    static_for<Q>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
#pragma unroll
        for (int fl = 0; fl < FP; ++fl) {
            const cf x = v[brev(q * FP + fl, LB1)];
            f2 t;
            t.x = x.re;
            t.y = x.im;
            LA[(fl * NW + w) * 64 + k] = t;
        }
        // operands of one frequency, requested one frequency ahead of their use
        cf dn[CC], sn[CC];
        float bn[2 * CC * CC];
        auto prefetch = [&](auto tc) {
            constexpr int t = decltype(tc)::value;       // frequency slot of this group
            constexpr int jl = t / NW, i = t % NW, j = q * LP + jl;
            const int fo = NW * j + N1 * brev(i, LBW);   // f - w
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                dn[c] = kv ? buf_load_cf_cached(Db, dko, (fo * CC + c) * K * (int)sizeof(cf)) : zero;
                sa_uload2(reinterpret_cast<const float *>(S + (fo * CC + c) * a.N), sn[c].re, sn[c].im);
            }
#pragma unroll
            for (int m = 0; m < 2 * CC * CC; ++m) bn[m] = sa_uload(B + fo * 2 * CC * CC + m);
        };
        prefetch(std::integral_constant<int, 0>{});
        __syncthreads();
        cf u[FP];
#pragma unroll
        for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
            for (int h2 = 0; h2 < NW; ++h2) {
                const f2 t = LA[((w + NW * jl) * NW + h2) * 64 + k];
                u[NW * jl + h2] = mk<float>(t.x, t.y);
            }
        }
        static_for<FP>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int jl = t / NW, i = t % NW;
            if constexpr (i == 0) dif<NW, false>(u, NW * jl);
            cf d[CC], s[CC];
            float b[2 * CC * CC];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                d[c] = dn[c];
                s[c] = sn[c];
            }
#pragma unroll
            for (int m = 0; m < 2 * CC * CC; ++m) b[m] = bn[m];
            if constexpr (t + 1 < FP) prefetch(std::integral_constant<int, t + 1>{});
            float red[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) red[m] = 0.f;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const cf p = cmul(d[c], u[t]);
                red[2 * c] = p.re;
                red[2 * c + 1] = p.im;
            }
            const float tot = reduce8_across_lanes(red, k);
            cf r[CC];
#pragma unroll
            for (int c = 0; c < CC; ++c)
                r[c] = s[c] - mk<float>(sa_readlane(tot, 16 * c), sa_readlane(tot, 16 * c + 8));
            cf x = u[t];
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                cf g = zero;
#pragma unroll
                for (int c2 = 0; c2 < CC; ++c2)
                    g = g + cmul(mk<float>(b[2 * (c * CC + c2)], b[2 * (c * CC + c2) + 1]), r[c2]);
                obj += cabs2(g);                   // Df.xf - Sf = -rho g
                x = x + cmulc(d[c], g);
            }
            u[t] = x;
            if constexpr (i == NW - 1) {
                constexpr int j = q * LP + jl;
                dit<NW, true>(u, NW * jl);
#pragma unroll
                for (int h2 = 1; h2 < NW; ++h2) {
                    cf tw;
                    sa_uload2(reinterpret_cast<const float *>(twB + NW * j + h2), tw.re, tw.im);
                    u[NW * jl + h2] = cmulc(tw, u[NW * jl + h2]);
                }
            }
        });
        SA_VGPR_FENCE3(obj, token, token);
#pragma unroll
        for (int jl = 0; jl < LP; ++jl) {
#pragma unroll
            for (int h2 = 0; h2 < NW; ++h2) {
                f2 t;
                t.x = u[NW * jl + h2].re;
                t.y = u[NW * jl + h2].im;
                LA[((w + NW * jl) * NW + h2) * 64 + k] = t;
            }
        }
        __syncthreads();
#pragma unroll
        for (int fl = 0; fl < FP; ++fl) {
            const f2 t = LA[(fl * NW + w) * 64 + k];
            v[brev(q * FP + fl, LB1)] = mk<float>(t.x, t.y);
        }
    });
    reg_fence<N1>(v, 0, token);

    dit<N1, true>(v, 0);
    if (kv) {
#pragma unroll
        for (int h1 = 0; h1 < N1; ++h1)
            buf_store_cf(Tb, ko, NW * h1 * K * (int)sizeof(cf), v[h1]);
    }
    const double pw = (wf == 0 || ((a.W & 1) == 0 && wf == Wf - 1)) ? 1.0 : 2.0;
    double acc[1] = {k == 0 ? (double)obj * pw * (double)a.rho * (double)a.rho : 0.0};
    block_sum_store<1>(acc, scratch, a.partials + tile);
}

// bt[row] = (rho I + Df Df^H)^-1 for every frequency row of the tile-major Df [rows][CC][K]:
// one wave per row (lane = filter, so the CC rows of K filters are read coalesced), the
// CC x CC Gram matrix by a butterfly reduction, then Gauss-Jordan on the Hermitian positive
// definite matrix in every lane; lane 0 stores.
template <int CC>
__global__ void __launch_bounds__(256) mc_binv_kernel(const cf *__restrict__ dft,
                                                      float *__restrict__ bt, int64_t nrows, int K,
                                                      float rho) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += nwaves) {
        const cf *d = dft + row * CC * K;
        cf dk[CC];
#pragma unroll
        for (int c = 0; c < CC; ++c) dk[c] = lane < K ? d[c * K + lane] : mk<float>(0.f, 0.f);
        cf m[CC][2 * CC];
        // (Hermitian: the upper triangle is reduced, the lower one mirrored)
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int c2 = c; c2 < CC; ++c2) {
                cf g = cmulc(dk[c2], dk[c]);        // d_c conj(d_c2)
                for (int s = 32; s > 0; s >>= 1) {
                    g.re += __shfl_xor(g.re, s, 64);
                    if (c2 != c) g.im += __shfl_xor(g.im, s, 64);
                }
                if (c == c2) g = mk<float>(g.re + rho, 0.f);
                m[c][c2] = g;
                m[c2][c] = cconj(g);
            }
#pragma unroll
        for (int c = 0; c < CC; ++c)
#pragma unroll
            for (int c2 = 0; c2 < CC; ++c2) m[c][CC + c2] = mk<float>(c == c2 ? 1.f : 0.f, 0.f);
#pragma unroll
        for (int p = 0; p < CC; ++p) {
            const float ip = 1.f / cabs2(m[p][p]);
            const cf inv = cscale(cconj(m[p][p]), ip);
#pragma unroll
            for (int c2 = 0; c2 < 2 * CC; ++c2) m[p][c2] = cmul(m[p][c2], inv);
#pragma unroll
            for (int c = 0; c < CC; ++c)
                if (c != p) {
                    const cf f = m[c][p];
#pragma unroll
                    for (int c2 = 0; c2 < 2 * CC; ++c2) m[c][c2] = m[c][c2] - cmul(f, m[p][c2]);
                }
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < CC; ++c)
#pragma unroll
                for (int c2 = 0; c2 < CC; ++c2) {
                    bt[(row * CC * CC + c * CC + c2) * 2] = m[c][CC + c2].re;
                    bt[(row * CC * CC + c * CC + c2) * 2 + 1] = m[c][CC + c2].im;
                }
        }
    }
}

template <int N1, int NW, int LP, int KC, int CC>
void launch_mc_inst(hipStream_t st, const FusedMcArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        SA_HIP(hipFuncSetAttribute(
            reinterpret_cast<const void *>(&fused_cols_mc_kernel<N1, NW, LP, KC, CC>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)mc_lds_bytes(NW, LP)));
    }
    const int64_t wf_groups = ceil_div(a.W / 2 + 1, 8);
    hipLaunchKernelGGL((fused_cols_mc_kernel<N1, NW, LP, KC, CC>),
                       dim3((unsigned)(wf_groups * 8 * a.N)), dim3(NW * 64), mc_lds_bytes(NW, LP),
                       st, a);
}

template <int CC> void launch_mc_cc(hipStream_t st, const FusedMcArgs<float> &a) {
    if (a.H == 128) {
        if (a.K == 64) launch_mc_inst<32, 4, 4, 64, CC>(st, a);
        else launch_mc_inst<32, 4, 4, 0, CC>(st, a);
    } else if (a.H == 256) {
        if (a.K == 64) launch_mc_inst<32, 8, 2, 64, CC>(st, a);
        else launch_mc_inst<32, 8, 2, 0, CC>(st, a);
    } else {
        if (a.K == 64) launch_mc_inst<32, 16, 1, 64, CC>(st, a);
        else launch_mc_inst<32, 16, 1, 0, CC>(st, a);
    }
}

}  // namespace

template <> bool fused_mc_supported<float>(int H, int K, int Cd) {
    return (H == 128 || H == 256 || H == 512) && K >= 1 && K <= 64 && Cd >= 2 && Cd <= 4;
}
template <> bool fused_mc_supported<double>(int, int, int) { return false; }

template <> int64_t launch_fused_cols_mc<float>(hipStream_t st, const FusedMcArgs<float> &a) {
    SA_REQUIRE(fused_mc_supported<float>(a.H, a.K, a.Cd), "shape not handled by the multi-channel column kernel");
    switch (a.Cd) {
    case 2: launch_mc_cc<2>(st, a); break;
    case 3: launch_mc_cc<3>(st, a); break;
    default: launch_mc_cc<4>(st, a); break;
    }
    SA_HIP(hipGetLastError());
    return (int64_t)(a.W / 2 + 1) * a.N;
}
template <> int64_t launch_fused_cols_mc<double>(hipStream_t, const FusedMcArgs<double> &) {
    throw Error(-1, "the fused column kernels are float32 only");
}

template <>
void launch_mc_binv<float>(hipStream_t st, const cx<float> *dft, float *bt, int64_t nrows, int Cd,
                           int K, float rho) {
    const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(nrows, 4), 65535 * 16);
    switch (Cd) {
    case 2: hipLaunchKernelGGL((mc_binv_kernel<2>), dim3(grid), dim3(256), 0, st, dft, bt, nrows, K, rho); break;
    case 3: hipLaunchKernelGGL((mc_binv_kernel<3>), dim3(grid), dim3(256), 0, st, dft, bt, nrows, K, rho); break;
    case 4: hipLaunchKernelGGL((mc_binv_kernel<4>), dim3(grid), dim3(256), 0, st, dft, bt, nrows, K, rho); break;
    default: throw Error(-1, "2 to 4 dictionary channels");
    }
    SA_HIP(hipGetLastError());
}
template <>
void launch_mc_binv<double>(hipStream_t, const cx<double> *, double *, int64_t, int, int, double) {
    throw Error(-1, "the fused column kernels are float32 only");
}

}  // namespace sporco_amd
