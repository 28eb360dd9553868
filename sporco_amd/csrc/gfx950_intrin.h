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

// all-reduce of (a, b) over the 64 lanes of the wave by an xor butterfly (every lane ends with
// the same bits: the tree is the same up to the order of the operands of each add)
template <typename T> __device__ __forceinline__ void sa_wave_allreduce2(T &a, T &b) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        a += __shfl_xor(a, m, 64);
        b += __shfl_xor(b, m, 64);
    }
}

// 1 / sqrt(x) and sqrt(x) to 1 ulp, no denormal / IEEE fix-up sequences (v_rsq_f32, v_sqrt_f32)
__device__ __forceinline__ float sa_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sa_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// a * b + c with one rounding, as one instruction (v_fma_f32).  The row kernels spell their
// fused multiply-adds out (and forbid the compiler's own contraction there): the two state
// forms of an ADMM iteration (csc_rows.h) must round alike, and which product of a sum the
// compiler fuses depends on the code around it.
__device__ __forceinline__ float sa_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// median of three (v_med3_f32): med3(v, -t, t) clamps v to [-t, t] for t >= 0
__device__ __forceinline__ float sa_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

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
// Cache policy of the streaming accesses (iterates and spectra that are touched once per
// kernel and are far larger than L2 + MALL): nt = 2 measured +4.5 % over the default
// policy on the whole iteration.  Re-read operands (Df) use the default policy.
#ifndef SA_STREAM_AUX
#define SA_STREAM_AUX 2
#endif
template <int AUX>
__device__ __forceinline__ void sa_buf_load2_aux(SaBuf r, int voff, int soff, float &a, float &b) {
    const sa_floatx2 t =
        __builtin_bit_cast(sa_floatx2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
    a = t.x;
    b = t.y;
}
__device__ __forceinline__ void sa_buf_load2(SaBuf r, int voff, int soff, float &a, float &b) {
    sa_buf_load2_aux<SA_STREAM_AUX>(r, voff, soff, a, b);
}
__device__ __forceinline__ void sa_buf_load2_cached(SaBuf r, int voff, int soff, float &a, float &b) {
    sa_buf_load2_aux<0>(r, voff, soff, a, b);
}
// one float, default cache policy (small re-read operands such as masks)
__device__ __forceinline__ float sa_buf_load1(SaBuf r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void sa_buf_store2(SaBuf r, int voff, int soff, float a, float b) {
    sa_floatx2 t;
    t.x = a;
    t.y = b;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sa_b64, t), r, voff, soff, SA_STREAM_AUX);
}
// 16-byte streaming accesses through plain pointers (the tile-major spectra in the row kernels)
typedef float sa_floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sa_stream_load4(const float *p, float (&v)[4]) {
    const sa_floatx4 t = SA_STREAM_AUX ? __builtin_nontemporal_load(reinterpret_cast<const sa_floatx4 *>(p))
                                       : *reinterpret_cast<const sa_floatx4 *>(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
}
__device__ __forceinline__ void sa_stream_store4(float *p, const float (&v)[4]) {
    sa_floatx4 t;
    t.x = v[0];
    t.y = v[1];
    t.z = v[2];
    t.w = v[3];
    if (SA_STREAM_AUX)
        __builtin_nontemporal_store(t, reinterpret_cast<sa_floatx4 *>(p));
    else
        *reinterpret_cast<sa_floatx4 *>(p) = t;
}

// Wave-uniform loads of read-only data through the scalar cache (s_load_dword[x2]):
// the pointer is re-typed to the constant address space, which is what lets the
// compiler keep them scalar after barriers/fences (a plain global load of a uniform
// address turns into a vector load as soon as anything before it may write memory).
__device__ __forceinline__ float sa_uload(const float *p) {
    return *(const float __attribute__((address_space(4))) *)p;
}
__device__ __forceinline__ void sa_uload2(const float *p, float &a, float &b) {
    const sa_floatx2 t = *(const sa_floatx2 __attribute__((address_space(4))) *)p;
    a = t.x;
    b = t.y;
}

// Cross-lane moves on the VALU (no LDS traffic, unlike ds_bpermute).
//   sa_swap32(a, b): lanes 32..63 of a <-> lanes 0..31 of b      (v_permlane32_swap)
//   sa_swap16(a, b): odd 16-lane rows of a <-> even rows of b     (v_permlane16_swap)
// (element access through named unsigned temporaries: subscripting the builtin's
// result directly as `const auto r; r[1]` is folded to r[0] by this clang)
typedef unsigned sa_uintx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sa_swap32(float &a, float &b) {
    sa_uintx2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a),
                                                   __builtin_bit_cast(unsigned, b), false, false);
    const unsigned x = r.x, y = r.y;
    a = __builtin_bit_cast(float, x);
    b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ void sa_swap16(float &a, float &b) {
    sa_uintx2 r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a),
                                                   __builtin_bit_cast(unsigned, b), false, false);
    const unsigned x = r.x, y = r.y;
    a = __builtin_bit_cast(float, x);
    b = __builtin_bit_cast(float, y);
}
// DPP reads: the value of `v` in lane l^1 / l^2 (quad_perm), l^7 (row_half_mirror),
// l^15 (row_mirror).
template <int CTRL> __device__ __forceinline__ float sa_dpp(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sa_lane_xor1(float v) { return sa_dpp<0xB1>(v); }
__device__ __forceinline__ float sa_lane_xor2(float v) { return sa_dpp<0x4E>(v); }
__device__ __forceinline__ float sa_lane_xor7(float v) { return sa_dpp<0x141>(v); }
__device__ __forceinline__ float sa_lane_xor15(float v) { return sa_dpp<0x140>(v); }

// Empty volatile asm that ties three values to vector registers at this point of
// the instruction stream (see reg_fence in csc_fused.hip): no instruction is
// emitted, it only orders the scheduler.
#define SA_VGPR_FENCE3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))

// constant-rate device clock (100 MHz) and a system-scope fence (host-visible records)
__device__ __forceinline__ unsigned long long sa_wall_clock() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void sa_fence_system() { __threadfence_system(); }

// Exchange between workgroups of ONE launch (the cooperating slab workgroups of csc_fused.hip):
// agent-scope stores are written through to where every XCD sees them, agent-scope loads bypass
// the non-coherent cache levels; sa_wait_stores holds the wave until its stores are acknowledged.
// (No acquire / release fences: at agent scope they write back and invalidate the whole L2.)
__device__ __forceinline__ void sa_store_agent(float *p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sa_store_agent(unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned sa_load_agent(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sa_load_agent2(const float *p, float &a, float &b) {   // 8-byte aligned
    const unsigned long long t = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __builtin_bit_cast(float, (unsigned)(t & 0xffffffffull));
    b = __builtin_bit_cast(float, (unsigned)(t >> 32));
}
__device__ __forceinline__ void sa_store_agent(double *p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double sa_load_agent(const double *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ... and whole arrays that change hands between the workgroups of one launch
// (admm_persist_kernel: the tile-major spectra): the same agent-scope accesses, 8 bytes at a
// time through plain pointers, or with the sc1 bit through a buffer descriptor.
__device__ __forceinline__ void sa_coh_load4(const float *p, float (&v)[4]) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[0] = __builtin_bit_cast(float, (unsigned)(a & 0xffffffffull));
    v[1] = __builtin_bit_cast(float, (unsigned)(a >> 32));
    v[2] = __builtin_bit_cast(float, (unsigned)(b & 0xffffffffull));
    v[3] = __builtin_bit_cast(float, (unsigned)(b >> 32));
}
__device__ __forceinline__ void sa_coh_store4(float *p, const float (&v)[4]) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    const unsigned long long a = (unsigned long long)__builtin_bit_cast(unsigned, v[0]) |
                                 ((unsigned long long)__builtin_bit_cast(unsigned, v[1]) << 32);
    const unsigned long long b = (unsigned long long)__builtin_bit_cast(unsigned, v[2]) |
                                 ((unsigned long long)__builtin_bit_cast(unsigned, v[3]) << 32);
    __hip_atomic_store(q, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kSaAuxAgent = 16;   // sc1: agent-scope coherence of a raw buffer access (gfx940+)
__device__ __forceinline__ void sa_buf_load2_coh(SaBuf r, int voff, int soff, float &a, float &b) {
    sa_buf_load2_aux<kSaAuxAgent>(r, voff, soff, a, b);
}
__device__ __forceinline__ void sa_buf_store2_coh(SaBuf r, int voff, int soff, float a, float b) {
    sa_floatx2 t;
    t.x = a;
    t.y = b;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(sa_b64, t), r, voff, soff, kSaAuxAgent);
}
__device__ __forceinline__ void sa_wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Barrier across ALL workgroups of one launch (admm_persist_kernel): whole arrays change hands,
// so here the fences are the real ones -- release writes this XCD's L2 back, acquire drops what
// the caches hold of other XCDs' data; the scalar data cache is not covered by either.
__device__ __forceinline__ unsigned sa_atomic_inc_agent(unsigned *p) {
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sa_fence_release_agent() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); }
__device__ __forceinline__ void sa_fence_acquire_agent() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
__device__ __forceinline__ void sa_scalar_cache_inv() { __builtin_amdgcn_s_dcache_inv(); }
__device__ __forceinline__ void sa_spin_pause() { __builtin_amdgcn_s_sleep(8); }

// A wave-uniform pointer made opaque to the optimiser (no instruction emitted): inside a
// persistent tile loop this keeps loop-invariant operand loads (twiddle tables, ...) where
// they are used instead of being hoisted in front of the loop, where they would occupy
// scalar registers for the whole kernel.
template <typename P> __device__ __forceinline__ P *sa_opaque_sptr(P *p) {
    asm volatile("" : "+s"(p));
    return p;
}
// The kernel's (single, by-value) argument struct re-read from the kernarg segment through
// an opaque pointer: fields used late in a long tile loop are then loaded (s_load) where
// they are used, every iteration, instead of living in scalar registers throughout.
#define SA_ARGS_PTR_T(A) const A __attribute__((address_space(4))) *
// OFF: byte offset of that struct inside the kernel's argument block (a kernel whose argument
// is a struct of several such structs: admm_persist_kernel).
template <bool OPAQUE = true, int OFF = 0, typename A>
__device__ __forceinline__ SA_ARGS_PTR_T(A) sa_args_reload(const A &) {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (OPAQUE) asm volatile("" : "+s"(p));
    return (SA_ARGS_PTR_T(A))((const char __attribute__((address_space(4))) *)p + OFF);
}

}  // namespace sporco_amd
