// tests/hostsim/gfx950_intrin.h -- TEST INFRASTRUCTURE ONLY.
//
// CPU emulation of sporco_amd/csrc/gfx950_intrin.h for the fiber simulator: the
// kernels include <gfx950_intrin.h> and this directory comes first on the
// simulator's include path.  Never seen by the hipcc build.
#pragma once

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
inline void sa_buf_store2(SaBuf r, int voff, int soff, float a, float b) {
    const uint32_t off = (uint32_t)voff + (uint32_t)soff;
    if ((uint64_t)off + 8 <= r.bytes) {
        std::memcpy(r.base + off, &a, 4);
        std::memcpy(r.base + off + 4, &b, 4);
    }
}

#define SA_VGPR_FENCE3(a, b, c) ((void)0)

}  // namespace sporco_amd
