// common.h -- shared host/device helpers for libsporco_amd (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <gfx950_intrin.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

namespace sporco_amd {

constexpr int kWave = 64;  // CDNA4 wavefront width

// ---------------------------------------------------------------------------
// error plumbing: C++ exceptions inside, codes + thread-local message at the ABI
// ---------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define SA_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t sa_e_ = (expr);                                                     \
        if (sa_e_ != hipSuccess)                                                       \
            throw ::sporco_amd::Error(-2, std::string(#expr) + ": " +                  \
                                              hipGetErrorString(sa_e_));               \
    } while (0)

#define SA_REQUIRE(cond, msg)                                                          \
    do {                                                                               \
        if (!(cond)) throw ::sporco_amd::Error(-1, std::string(msg));                  \
    } while (0)

// ---------------------------------------------------------------------------
// complex numbers as plain 2-vectors (layout-compatible with numpy complex64/128)
// ---------------------------------------------------------------------------
template <typename T> struct cx {
    T re, im;
};

template <typename T> __host__ __device__ __forceinline__ cx<T> mk(T a, T b) {
    cx<T> r;
    r.re = a;
    r.im = b;
    return r;
}
template <typename T> __host__ __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) {
    return mk<T>(a.re + b.re, a.im + b.im);
}
template <typename T> __host__ __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) {
    return mk<T>(a.re - b.re, a.im - b.im);
}
// occupancy floor of a kernel (waves per SIMD; caps its registers at 512 / n)
#ifdef SPORCO_AMD_HOSTSIM
#define SA_MIN_WAVES_PER_SIMD(n)
#else
#define SA_MIN_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif
// a * b + c with ONE rounding, spelled out.  The complex products below name their fused
// multiply-adds instead of leaving `x * y + z * w` to the compiler's contraction, which picks the
// product to fuse by the code AROUND the expression: two instantiations of the same transform
// (the (Y, U) and the V form of an epilogue, the one-launch solve and the launch-per-pass loop)
// then round differently.  The register-kernel translation units are compiled with
// -ffp-contract=off on top of this (Makefile), so what is not spelled out is not fused.
__host__ __device__ __forceinline__ float fma1(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__host__ __device__ __forceinline__ double fma1(double a, double b, double c) { return __builtin_fma(a, b, c); }
// a * b
template <typename T> __host__ __device__ __forceinline__ cx<T> cmul(cx<T> a, cx<T> b) {
    return mk<T>(fma1(a.re, b.re, -(a.im * b.im)), fma1(a.re, b.im, a.im * b.re));
}
// conj(a) * b
template <typename T> __host__ __device__ __forceinline__ cx<T> cmulc(cx<T> a, cx<T> b) {
    return mk<T>(fma1(a.re, b.re, a.im * b.im), fma1(a.re, b.im, -(a.im * b.re)));
}
// u + conj(a) * b and acc + |a|^2, every product fused into the sum it feeds (four / two
// instructions)
template <typename T> __host__ __device__ __forceinline__ cx<T> cmulc_add(cx<T> u, cx<T> a, cx<T> b) {
    return mk<T>(fma1(a.re, b.re, fma1(a.im, b.im, u.re)), fma1(a.re, b.im, fma1(-a.im, b.re, u.im)));
}
template <typename T> __host__ __device__ __forceinline__ T cabs2_add(T acc, cx<T> a) {
    return fma1(a.re, a.re, fma1(a.im, a.im, acc));
}
template <typename T> __host__ __device__ __forceinline__ cx<T> cscale(cx<T> a, T s) {
    return mk<T>(a.re * s, a.im * s);
}
template <typename T> __host__ __device__ __forceinline__ cx<T> cconj(cx<T> a) {
    return mk<T>(a.re, -a.im);
}
template <typename T> __host__ __device__ __forceinline__ T cabs2(cx<T> a) {
    return fma1(a.re, a.re, a.im * a.im);
}
// multiply by -i (forward quarter turn) / +i
template <typename T> __host__ __device__ __forceinline__ cx<T> mul_mi(cx<T> a) {
    return mk<T>(a.im, -a.re);
}
template <typename T> __host__ __device__ __forceinline__ cx<T> mul_pi(cx<T> a) {
    return mk<T>(-a.im, a.re);
}

// ---------------------------------------------------------------------------
// LDS: every kernel carves its scratch from the dynamic region (keeps the
// base 16-byte aligned and leaves no static __shared__ objects around).
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T *dyn_lds() {
#ifdef SPORCO_AMD_HOSTSIM
    return reinterpret_cast<T *>(hostsim::lds_base());   // (the CPU test simulator)
#else
    extern __shared__ __attribute__((aligned(16))) unsigned char sporco_amd_lds_raw[];
    return reinterpret_cast<T *>(sporco_amd_lds_raw);
#endif
}

// ---------------------------------------------------------------------------
// deterministic reductions of double accumulators
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#ifdef SPORCO_AMD_HOSTSIM
    return hostsim::wave_allreduce(v);      // (the CPU test simulator: same tree, one exchange)
#else
#pragma unroll
    for (int m = kWave / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
#endif
}

// Sum `NV` per-thread doubles over the block; thread 0 writes them to
// dst[0..NV).  `scratch` must hold NV * (blockDim.x / 64) doubles of LDS.
// AGENT: the sums are read by other workgroups of the same launch (agent-scope stores).
template <int NV, bool AGENT = false>
__device__ __forceinline__ void block_sum_store(const double (&acc)[NV], double *scratch,
                                                double *dst) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const int nwave = (blockDim.x + kWave - 1) / kWave;
    double w[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) w[i] = wave_sum(acc[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = w[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
            for (int j = 0; j < nwave; ++j) s += scratch[j * NV + i];
            if constexpr (AGENT) sa_store_agent(dst + i, s);
            else dst[i] = s;
        }
    }
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// "once per device" for the launchers' function attributes (hipFuncSetAttribute applies to the
// function as loaded on the CURRENT device; a process may drive several):
//     static PerDeviceOnce attr;  if (attr.first()) { hipFuncSetAttribute(...); }
struct PerDeviceOnce {
    bool done[64] = {};
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

// Compute units of the CURRENT device (the C ABI selects the handle's device before every call),
// cached per device id: the persistent launches size their grids by it, and a process may drive
// devices of different sizes or partition modes.
inline int current_device_cus() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!cache[dev]) {
        hipDeviceProp_t pr;
        SA_HIP(hipGetDeviceProperties(&pr, dev));
        cache[dev] = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
    }
    return cache[dev];
}

}  // namespace sporco_amd
