// Micro-benchmark (measurement only): issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add_f32 on
// gfx950 at 1, 2 and 4 waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE> __global__ void __launch_bounds__(1024) k(float *out, int iters, long long *cyc) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = 1.0001f, c = 0.5f;
    f2 pb = {b, b}, pc = {c, c};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                             "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
            }
        } else {
            // packed fma with op_sel / neg modifiers (the complex-multiply forms)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %1, %1, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %2, %2, %8, %9 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %3, %3, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %4, %4, %8, %9 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %5, %5, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n"
                             "v_pk_fma_f32 %6, %6, %8, %9 op_sel:[0,1,0] op_sel_hi:[0,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %7, %7, %8, %9 op_sel:[1,1,0] op_sel_hi:[1,0,1]\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, int threads) {
    float *out; long long *cyc, h;
    hipMalloc(&out, sizeof(float) * 1024 * 256); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(out, 10, cyc);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double ninst = (double)iters * 64;          // per wave
    const int waves_per_simd = threads / 64 / 4;
    printf("%-28s waves/SIMD %d: %.2f ms, s_memtime ticks/instr/wave %.2f, wall ns per instr per SIMD %.3f\n", name,
           waves_per_simd ? waves_per_simd : 1, ms, (double)h / ninst, ms * 1e6 / (ninst * (waves_per_simd ? waves_per_simd : 1)));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {256, 512, 1024}) {
        run<0>("v_fma_f32", th); run<1>("v_pk_fma_f32", th); run<2>("v_pk_add_f32", th); run<3>("v_add_f32", th); run<4>("v_pk_fma_f32 op_sel/neg", th);
    }
    return 0;
}
