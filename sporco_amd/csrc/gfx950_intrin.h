// gfx950_intrin.h -- the handful of CDNA4 builtins the kernels use directly,
// behind plain function names.  Included as <gfx950_intrin.h>: the product build
// finds this file; the CPU fiber simulator of tests/hostsim (test infrastructure
// only) supplies its own file of the same name that emulates the operations.
#pragma once

#include <cstdint>

namespace sporco_amd {

// value of lane 0's `v`, as a wave-uniform scalar (v_readfirstlane_b32)
__device__ __forceinline__ int sa_readfirstlane(int v) { return __builtin_amdgcn_readfirstlane(v); }

// value of `v` in lane `src` (compile-time constant), wave-uniform (v_readlane_b32)
__device__ __forceinline__ float sa_readlane(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}

// 1 / x to 1 ulp (v_rcp_f32)
__device__ __forceinline__ float sa_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Raw buffer access: a 128-bit descriptor in SGPRs, byte offset = voff (per
// lane) + soff (wave-uniform).  Accesses beyond `bytes` read zero / are dropped.
typedef __amdgpu_buffer_rsrc_t SaBuf;
__device__ __forceinline__ SaBuf sa_make_buf(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
typedef float sa_floatx2 __attribute__((ext_vector_type(2)));
typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(SaBuf(), 0, 0, 0)) sa_b64;
__device__ __forceinline__ void sa_buf_load2(SaBuf r, int voff, int soff, float &a, float &b) {
    const sa_floatx2 t =
        __builtin_bit_cast(sa_floatx2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
    a = t.x;
    b = t.y;
}
__device__ __forceinline__ void sa_buf_store2(SaBuf r, int voff, int soff, float a, float b) {
    sa_floatx2 t;
    t.x = a;
    t.y = b;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sa_b64, t), r, voff, soff, 0);
}

// Empty volatile asm that ties three values to vector registers at this point of
// the instruction stream (see reg_fence in csc_fused.hip): no instruction is
// emitted, it only orders the scheduler.
#define SA_VGPR_FENCE3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))

}  // namespace sporco_amd
