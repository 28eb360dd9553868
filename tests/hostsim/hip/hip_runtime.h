// tests/hostsim/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal stand-in for <hip/hip_runtime.h> that lets the UNMODIFIED kernel
// sources of sporco_amd/csrc be compiled with g++ and executed on the CPU, one
// workgroup at a time, with every GPU thread modelled as a cooperative fiber
// (so __syncthreads and wave64 shuffles have their real semantics).  It exists
// because the authoring container has no GPU: index arithmetic, LDS carving and
// barrier placement of the kernels are debugged here, and the real hipcc build
// is then checked on an MI355X through `pytest -m gpu`.
//
// Nothing in sporco_amd/ includes, links or loads this.  The product library
// (libsporco_amd.so) is always the hipcc/gfx950 build and fails loudly without
// a GPU; the simulator library has a different name and is only ever loaded by
// tests/ (tests/hostsim/README.md).
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>

#define SPORCO_AMD_HOSTSIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ /* only ever used as `extern __shared__ ... name[]` */

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef void *hipStream_t;
struct hostsim_event;
typedef hostsim_event *hipEvent_t;
enum hipMemcpyKind {
    hipMemcpyHostToHost = 0,
    hipMemcpyHostToDevice = 1,
    hipMemcpyDeviceToHost = 2,
    hipMemcpyDeviceToDevice = 3,
    hipMemcpyDefault = 4
};
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
    char name[256];
    int multiProcessorCount;
    size_t totalGlobalMem;
};

hipError_t hipMalloc(void **p, size_t n);
hipError_t hipFree(void *p);
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width,
                            size_t height, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t *st);
hipError_t hipStreamDestroy(hipStream_t st);
hipError_t hipStreamSynchronize(hipStream_t st);
hipError_t hipDeviceSynchronize();
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int *d);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipGetLastError();
const char *hipGetErrorString(hipError_t e);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t st);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);

namespace hostsim {

void syncthreads();
void wave_sync();
void *shuffle_slot(int tid);
void *lds_base();          // the running workgroup's dynamic LDS
void set_coop(int n);      // the next launch runs n consecutive workgroups side by side
void spin_pause();         // inside a poll of memory another workgroup writes
int block_threads();
void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);

template <typename T> inline T shfl_xor(T v, int mask, int width) {
    static_assert(sizeof(T) <= 16, "shuffle payload too large");
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, base = tid - lane;
    std::memcpy(shuffle_slot(tid), &v, sizeof(T));
    wave_sync();
    T r = v;
    const int src = lane ^ mask;
    // sources stay inside the lane's own `width`-aligned segment and inside the block
    if (src / width == lane / width && base + src < block_threads())
        std::memcpy(&r, shuffle_slot(base + src), sizeof(T));
    wave_sync();
    return r;
}

// Sum of NV values per lane over the 64 lanes in the association of the xor butterfly
// (m = 32, ..., 1), delivered to every lane: the first lane of the wave computes the tree once
// between the two syncs (a shuffle-per-level emulation costs six exchange rounds, and a
// butterfly per lane costs 64 times the arithmetic).
template <typename T, int NV> inline void wave_allreduce_n(T (&v)[NV]) {
    static_assert(sizeof(T) * NV <= 16, "payload too large");
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, base = tid - lane;
    std::memcpy(shuffle_slot(tid), v, sizeof(T) * NV);
    wave_sync();
    if (lane == 0) {
        const int nl = block_threads() - base < 64 ? block_threads() - base : 64;
        T a[64][NV], n[64][NV];
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < NV; ++j) a[l][j] = T(0);
        for (int l = 0; l < nl; ++l) std::memcpy(a[l], shuffle_slot(base + l), sizeof(T) * NV);
        for (int m = 32; m > 0; m >>= 1) {
            // (a lane beyond the block reads as the caller's own value in __shfl_xor)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < NV; ++j) n[l][j] = a[l][j] + (((l ^ m) < nl) ? a[l ^ m][j] : a[l][j]);
            std::memcpy(a, n, sizeof(a));
        }
        for (int l = 0; l < nl; ++l) std::memcpy(shuffle_slot(base + l), a[l], sizeof(T) * NV);
    }
    wave_sync();
    std::memcpy(v, shuffle_slot(tid), sizeof(T) * NV);
}
template <typename T> inline T wave_allreduce(T v) {
    T a[1] = {v};
    wave_allreduce_n<T, 1>(a);
    return a[0];
}

template <typename... KA, typename... A>
inline void launch(void (*kernel)(KA...), dim3 grid, dim3 block, size_t shmem, hipStream_t,
                   A &&...args) {
    std::tuple<std::decay_t<KA>...> pack(std::forward<A>(args)...);
    run_grid(grid, block, shmem, [&]() { std::apply(kernel, pack); });
}

}  // namespace hostsim

inline void __syncthreads() { hostsim::syncthreads(); }
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
    return hostsim::shfl_xor<T>(v, mask, width);
}

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::hostsim::launch(kernel, grid, block, shmem, stream, __VA_ARGS__)
