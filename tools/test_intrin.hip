// Standalone probe of the cross-lane intrinsics in gfx950_intrin.h (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#define __forceinline__ inline __attribute__((always_inline))
#include "../sporco_amd/csrc/gfx950_intrin.h"
using namespace sporco_amd;
__global__ void probe(float *out) {
    const int l = threadIdx.x;
    float a = (float)l, b = 100.f + l;
    sa_swap32(a, b);
    out[0 * 64 + l] = a; out[1 * 64 + l] = b;
    a = (float)l; b = 100.f + l;
    sa_swap16(a, b);
    out[2 * 64 + l] = a; out[3 * 64 + l] = b;
    out[4 * 64 + l] = sa_lane_xor1((float)l);
    out[5 * 64 + l] = sa_lane_xor2((float)l);
    out[6 * 64 + l] = sa_lane_xor7((float)l);
    out[7 * 64 + l] = sa_lane_xor15((float)l);
    out[8 * 64 + l] = sa_readlane((float)l * 2.f, 17);
}
int main() {
    float *d; hipMalloc(&d, 9 * 64 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    float h[9 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[9] = {"swap32.a", "swap32.b", "swap16.a", "swap16.b", "xor1", "xor2", "xor7", "xor15", "readlane17x2"};
    for (int r = 0; r < 9; ++r) { printf("%-12s", names[r]); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); }
    return 0;
}
