// tests/hostsim/gfx950_intrin.h -- TEST INFRASTRUCTURE ONLY.
//
// CPU emulation of sporco_amd/csrc/gfx950_intrin.h for the fiber simulator: the
// kernels include <gfx950_intrin.h> and this directory comes first on the
// simulator's include path.  Never seen by the hipcc build.
#pragma once
#include <algorithm>
#include <cmath>
#include <chrono>

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

namespace sporco_amd {

inline int sa_readfirstlane(int v) {
    // every caller passes a value that is uniform across the wave by construction
    return v;
}

inline float sa_readlane(float v, int src) {
    const int tid = (int)threadIdx.x;
    const int base = tid - (tid & 63);
    std::memcpy(hostsim::shuffle_slot(tid), &v, sizeof(float));
    hostsim::wave_sync();
    float r;
    std::memcpy(&r, hostsim::shuffle_slot(base + src), sizeof(float));
    hostsim::wave_sync();
    return r;
}

inline float sa_rcp(float x) { return 1.0f / x; }

struct SaBuf {
    char *base;
    uint32_t bytes;
};
inline SaBuf sa_make_buf(const void *base, uint32_t bytes) {
    SaBuf b;
    b.base = const_cast<char *>(static_cast<const char *>(base));
    b.bytes = bytes;
    return b;
}
inline void sa_buf_load2(SaBuf r, int voff, int soff, float &a, float &b) {
    const uint32_t off = (uint32_t)voff + (uint32_t)soff;
    a = b = 0.f;  // out-of-range reads return zero, as the hardware bounds check does
    if ((uint64_t)off + 8 <= r.bytes) {
        std::memcpy(&a, r.base + off, 4);
        std::memcpy(&b, r.base + off + 4, 4);
    }
}
inline float sa_buf_load1(SaBuf r, int voff, int soff) {
    const uint32_t off = (uint32_t)voff + (uint32_t)soff;
    float a = 0.f;
    if ((uint64_t)off + 4 <= r.bytes) std::memcpy(&a, r.base + off, 4);
    return a;
}
inline void sa_buf_load2_cached(SaBuf r, int voff, int soff, float &a, float &b) {
    sa_buf_load2(r, voff, soff, a, b);
}
inline void sa_stream_load4(const float *p, float (&v)[4]) {
    for (int i = 0; i < 4; ++i) v[i] = p[i];
}
inline void sa_stream_store4(float *p, const float (&v)[4]) {
    for (int i = 0; i < 4; ++i) p[i] = v[i];
}
inline void sa_buf_store2(SaBuf r, int voff, int soff, float a, float b) {
    const uint32_t off = (uint32_t)voff + (uint32_t)soff;
    if ((uint64_t)off + 8 <= r.bytes) {
        std::memcpy(r.base + off, &a, 4);
        std::memcpy(r.base + off + 4, &b, 4);
    }
}

inline float sa_uload(const float *p) { return *p; }
inline void sa_uload2(const float *p, float &a, float &b) {
    a = p[0];
    b = p[1];
}

// value of `v` held by lane `src` of the calling thread's wave (src may differ per lane)
inline float hostsim_gather(float v, int src) {
    const int tid = (int)threadIdx.x;
    const int base = tid - (tid & 63);
    std::memcpy(hostsim::shuffle_slot(tid), &v, sizeof(float));
    hostsim::wave_sync();
    float r;
    std::memcpy(&r, hostsim::shuffle_slot(base + src), sizeof(float));
    hostsim::wave_sync();
    return r;
}
inline void sa_swap32(float &a, float &b) {
    const int lane = (int)threadIdx.x & 63;
    const float b_lo = hostsim_gather(b, lane & 31);         // b[lane - 32] for the upper half
    const float a_hi = hostsim_gather(a, (lane & 31) + 32);  // a[lane + 32] for the lower half
    if (lane < 32)
        b = a_hi;
    else
        a = b_lo;
}
inline void sa_swap16(float &a, float &b) {
    const int lane = (int)threadIdx.x & 63;
    const bool odd_row = (lane >> 4) & 1;
    const float b_even = hostsim_gather(b, lane & ~16);  // b[lane - 16] for odd rows
    const float a_odd = hostsim_gather(a, lane | 16);    // a[lane + 16] for even rows
    if (odd_row)
        a = b_even;
    else
        b = a_odd;
}
// all-reduce of (a, b) over the 64 lanes in the association of the xor-butterfly
// (m = 32, 16, ..., 1): ONE exchange on the simulator instead of six shuffle rounds per value
template <typename T> inline void sa_wave_allreduce2(T &a, T &b) {
    T v[2] = {a, b};
    hostsim::wave_allreduce_n<T, 2>(v);
    a = v[0];
    b = v[1];
}
inline void sa_store_agent(double *p, double v) { *p = v; }
inline double sa_load_agent(const double *p) { return *p; }
inline void sa_store_agent(float *p, float v) { *p = v; }
inline void sa_store_agent(unsigned *p, unsigned v) { *p = v; }
inline unsigned sa_load_agent(const unsigned *p) { return *(const volatile unsigned *)p; }
inline void sa_load_agent2(const float *p, float &a, float &b) {
    a = p[0];
    b = p[1];
}
inline void sa_buf_load2_coh(SaBuf r, int voff, int soff, float &a, float &b) { sa_buf_load2(r, voff, soff, a, b); }
inline void sa_buf_store2_coh(SaBuf r, int voff, int soff, float a, float b) { sa_buf_store2(r, voff, soff, a, b); }
inline void sa_coh_load4(const float *p, float (&v)[4]) { for (int i = 0; i < 4; ++i) v[i] = p[i]; }
inline void sa_coh_store4(float *p, const float (&v)[4]) { for (int i = 0; i < 4; ++i) p[i] = v[i]; }
inline void sa_wait_stores() {}
inline unsigned sa_atomic_inc_agent(unsigned *p) { return (*p)++; }
inline void sa_fence_release_agent() {}
inline void sa_fence_acquire_agent() {}
inline void sa_scalar_cache_inv() {}
inline void sa_spin_pause() { hostsim::spin_pause(); }
inline float sa_fma(float a, float b, float c) { return std::fma(a, b, c); }
inline float sa_med3(float a, float b, float c) {
    return std::max(std::min(a, b), std::min(std::max(a, b), c));
}
inline float sa_rsq(float x) { return 1.0f / std::sqrt(x); }
inline float sa_sqrt(float x) { return std::sqrt(x); }
inline float sa_lane_xor1(float v) { return hostsim_gather(v, ((int)threadIdx.x & 63) ^ 1); }
inline float sa_lane_xor2(float v) { return hostsim_gather(v, ((int)threadIdx.x & 63) ^ 2); }
inline float sa_lane_xor7(float v) { return hostsim_gather(v, ((int)threadIdx.x & 63) ^ 7); }
inline float sa_lane_xor15(float v) { return hostsim_gather(v, ((int)threadIdx.x & 63) ^ 15); }

#define SA_VGPR_FENCE3(a, b, c) ((void)0)

inline unsigned long long sa_wall_clock() {
    return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(
                                    std::chrono::steady_clock::now().time_since_epoch()).count() / 10);
}
inline void sa_fence_system() {}
template <typename P> inline P *sa_opaque_sptr(P *p) { return p; }
inline void __builtin_amdgcn_s_sleep(int) {}
#define SA_ARGS_PTR_T(A) const A *
template <bool OPAQUE = true, int OFF = 0, typename A> inline const A *sa_args_reload(const A &a) { return &a; }

}  // namespace sporco_amd
