// Micro-benchmark (measurement only): how much HBM bandwidth ONE CU can pull when only a
// fraction of the CUs are in a memory phase -- the question behind "would de-synchronising the
// column kernel's phases across CUs pay" (profiles/r04_fused_cols_notes.md).
// Persistent 16-wave workgroups (128 KiB LDS each: one per CU) copy 256 KiB tiles in place with
// the column kernel's access pattern (wave w, lane k, rows h = 16 h1 + w, 512-byte rows).
//   part 1: G = 256 ... 16 workgroups, no arithmetic: B/clk per active CU as G shrinks.
//   part 2: G = 256, a barrier-locked arithmetic phase of `gap` spins between the loads and
//           the stores (every wave of the workgroup in the same phase, like fused_cols), with
//           the workgroups started in `ph` staggered phases.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef __amdgpu_buffer_rsrc_t Buf;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(Buf(), 0, 0, 0)) b64;

__device__ __forceinline__ void spin(float (&x)[8], int n) {
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = x[j] * 1.0001f + 0.5f;
    }
}

__global__ void __launch_bounds__(1024) k(float *t, int ntiles, int gap, int phases, int sleeps, float *sink,
                                          unsigned long long *cyc) {
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    if (phases > 1) {
        const int ph = (int)(blockIdx.x >> 3) % phases;
        for (int i = 0; i < ph * sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        Buf b = __builtin_amdgcn_make_buffer_rsrc(t + (int64_t)tile * 65536, 0, 262144, 0x00020000);
        f2 v[32];
        const int vo = (w * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            v[i] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(b, vo, i * 16 * 512, 2));
        if (gap) {
            acc[0] += v[0].x + v[31].y;     // wait for the loads
            __syncthreads();
            spin(acc, gap);
            __syncthreads();
            v[0].x += acc[1] * 1e-30f;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            v[i].x += 1.0f;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64, v[i]), b, vo, i * 16 * 512, 2);
        }
    }
    if (tid == 0 && cyc) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (acc[0] == 123.456f) sink[0] = acc[0] + acc[7];
}

int main() {
    const int ntiles = 8224;
    float *t, *sink;
    unsigned long long *cyc;
    (void)hipMalloc(&t, (size_t)ntiles * 262144); (void)hipMalloc(&sink, 64);
    (void)hipMalloc(&cyc, 256 * 8);
    (void)hipMemset(t, 0, (size_t)ntiles * 262144);
    (void)hipFuncSetAttribute((const void *)&k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](int G, int nt, int gap, int phases, int sleeps) {
        k<<<G, 1024, 131072>>>(t, nt, gap, phases, sleeps, sink, cyc);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) k<<<G, 1024, 131072>>>(t, nt, gap, phases, sleeps, sink, cyc);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
        unsigned long long h[256];
        (void)hipMemcpy(h, cyc, G * 8, hipMemcpyDeviceToHost);
        double mc = 0; for (int i = 0; i < G; ++i) mc += (double)h[i]; mc /= G;
        const double bytes = 2.0 * nt * 262144;
        printf("{\"G\": %d, \"tiles\": %d, \"gap\": %d, \"phases\": %d, \"sleeps\": %d, \"ms\": %.4f, \"TBps\": %.3f, "
               "\"GBps_per_CU\": %.2f, \"cycles_per_tile_per_wg\": %.0f, \"B_per_clk_per_CU\": %.2f}\n",
               G, nt, gap, phases, sleeps, ms, bytes / ms / 1e9, bytes / ms / 1e6 / G,
               mc / ((double)nt / G), bytes / G / mc);
    };
    // part 1: fewer active CUs, same tiles per CU (32)
    for (int G : {256, 192, 128, 96, 64, 32, 16, 8}) run(G, 32 * G, 0, 1, 0);
    // part 2: phase-locked arithmetic between loads and stores
    for (int gap : {0, 300, 600, 900})
        for (int ph : {1, 2, 4, 8}) run(256, ntiles, gap, ph, gap ? (ph > 1 ? 16 : 0) : 0);
    return 0;
}
