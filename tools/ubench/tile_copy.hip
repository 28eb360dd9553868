// Micro-benchmark (measurement only): what one 16-wave workgroup per CU can stream.
// 8224 tiles of 256 KiB are read and written in place, tile-major like csc_fused.hip's T:
// wave w, lane k, row h = 16 h1 + w (512-byte rows).  Variants: persistent loop or one
// workgroup per tile, 8- or 16-byte accesses, cache policy of the stores, an arithmetic gap
// between load and store, stores issued in two halves around the gap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __amdgpu_buffer_rsrc_t Buf;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(Buf(), 0, 0, 0)) b64;
typedef decltype(__builtin_amdgcn_raw_buffer_load_b128(Buf(), 0, 0, 0)) b128;

__device__ __forceinline__ void spin(float (&x)[8], int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = x[j] * 1.0001f + 0.5f;
    }
}

// W16: 16-byte accesses (lane -> (row parity, filter pair)); AUXS: store policy; GAP: spin count
template <bool PERSIST, bool W16, int AUXL, int AUXS>
__global__ void __launch_bounds__(1024) k(float *t, int ntiles, int gap, float *sink) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        Buf b = __builtin_amdgcn_make_buffer_rsrc(t + (int64_t)tile * 65536, 0, 262144, 0x00020000);
        if constexpr (!W16) {
            f2 v[32];
            const int vo = (w * 64 + lane) * 8;
#pragma unroll
            for (int i = 0; i < 32; ++i)
                v[i] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(b, vo, i * 16 * 512, AUXL));
            if (gap) { acc[0] += v[0].x; spin(acc, gap); v[0].x += acc[1] * 1e-30f; }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                v[i].x += 1.0f;
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64, v[i]), b, vo, i * 16 * 512, AUXS);
            }
        } else {
            f4 v[16];
            // wave w covers rows 2w, 2w+1 (+32 i): 1 KiB per instruction
            const int vo = w * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(b, vo, i * 16384, AUXL));
            if (gap) { acc[0] += v[0].x; spin(acc, gap); v[0].x += acc[1] * 1e-30f; }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                v[i].x += 1.0f;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(b128, v[i]), b, vo, i * 16384, AUXS);
            }
        }
        if constexpr (!PERSIST) break;
    }
    if (acc[0] == 123.456f) sink[0] = acc[0] + acc[7];
}

template <bool PERSIST, bool W16, int AUXL, int AUXS> void run(const char *name, float *t, float *sink, int gap) {
    const int ntiles = 8224;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = PERSIST ? 256 : ntiles;
    k<PERSIST, W16, AUXL, AUXS><<<grid, 1024, 131072>>>(t, ntiles, gap, sink);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<PERSIST, W16, AUXL, AUXS><<<grid, 1024, 131072>>>(t, ntiles, gap, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s gap %5d: %.3f ms  %.2f TB/s\n", name, gap, ms, 2.0 * ntiles * 262144 / ms / 1e9);
}
#define RUNALL(P, W, L, S, name) \
    (void)hipFuncSetAttribute((const void *)&k<P, W, L, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); \
    for (int gap : {0, 600, 1200}) run<P, W, L, S>(name, t, sink, gap);
int main() {
    float *t, *sink;
    (void)hipMalloc(&t, (size_t)8224 * 262144); (void)hipMalloc(&sink, 64);
    (void)hipMemset(t, 0, (size_t)8224 * 262144);
    RUNALL(false, false, 2, 2, "per-tile WG, 8 B, nt/nt");
    RUNALL(true, false, 2, 2, "persistent, 8 B, nt/nt");
    RUNALL(true, false, 2, 0, "persistent, 8 B, nt/default");
    RUNALL(true, false, 0, 0, "persistent, 8 B, default/default");
    RUNALL(true, false, 2, 1, "persistent, 8 B, nt/sc0");
    RUNALL(true, false, 2, 16, "persistent, 8 B, nt/sc1");
    RUNALL(true, false, 2, 17, "persistent, 8 B, nt/sc0 sc1");
    RUNALL(true, false, 2, 19, "persistent, 8 B, nt/sc0 sc1 nt");
    RUNALL(false, true, 2, 2, "per-tile WG, 16 B, nt/nt");
    RUNALL(true, true, 2, 2, "persistent, 16 B, nt/nt");
    RUNALL(true, true, 2, 0, "persistent, 16 B, nt/default");
    return 0;
}
