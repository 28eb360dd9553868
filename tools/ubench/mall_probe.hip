// mall_probe.hip -- does the 256 MiB Infinity Cache (MALL) keep what a kernel has just written
// or read, and how fast is a read that hits it?  (image-blocked launch
// order so that the spectrum T of a group of images stays on the die between the column
// kernel and the row epilogue.)
//
//   hipcc -O3 --offload-arch=gfx950 mall_probe.hip -o mall_probe && ./mall_probe
//
// For a buffer of X MB: [producer pass over the buffer] [optional polluter: Q MB of unrelated
// streaming traffic] [timed consumer read of the buffer].  Producer = write or read, each with
// the default or the non-temporal (nt) cache policy; consumer = read, default or nt.  Output:
// one JSON line per configuration with the consumer's GB/s (median of 7).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT> __global__ void __launch_bounds__(256) k_write(f4 *p, size_t n, float v) {
    const f4 val = {v, v + 1.f, v + 2.f, v + 3.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(val, p + i);
        else p[i] = val;
    }
}
template <bool NT> __global__ void __launch_bounds__(256) k_read(const f4 *p, size_t n, float *sink) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f4 t = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc += t;
    }
    if (acc.x + acc.y + acc.z + acc.w == 1234.5f) sink[0] = acc.x;
}
template <bool NT> __global__ void __launch_bounds__(256) k_copy(const f4 *a, f4 *b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const f4 t = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(t, b + i);
        else b[i] = t;
    }
}

int main() {
    const size_t MB = 1 << 20;
    const size_t maxX = 1024 * MB, polb = 1024 * MB;
    f4 *A, *P1, *P2;
    float *sink;
    CK(hipMalloc((void **)&A, maxX));
    CK(hipMalloc((void **)&P1, polb));
    CK(hipMalloc((void **)&P2, polb));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(A, 0, maxX));
    CK(hipMemset(P1, 0, polb));
    CK(hipMemset(P2, 0, polb));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 256 * 8;
    const int sizes[] = {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024};
    const int pollute[] = {0, 128, 512};
    // producer: 0 write default, 1 write nt, 2 read default, 3 read nt; consumer: 0 default, 1 nt
    for (int x : sizes)
        for (int q : pollute)
            for (int prod = 0; prod < 4; ++prod)
                for (int cons = 0; cons < 2; ++cons) {
                    if (q == 128 && !(prod == 0 || prod == 1)) continue;
                    const size_t n = (size_t)x * MB / sizeof(f4), nq = (size_t)q * MB / sizeof(f4);
                    std::vector<float> ms;
                    for (int rep = 0; rep < 7; ++rep) {
                        // flush: a 2 GB default-policy copy between two other buffers
                        hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, P1, P2, polb / sizeof(f4));
                        switch (prod) {
                        case 0: hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, A, n, (float)rep); break;
                        case 1: hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, A, n, (float)rep); break;
                        case 2: hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, A, n, sink); break;
                        default: hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, A, n, sink); break;
                        }
                        // polluter: nt copy of q/2 MB -> q MB of streaming traffic
                        if (nq) hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, P1, P2, nq / 2);
                        CK(hipEventRecord(e0, 0));
                        if (cons == 0) hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, A, n, sink);
                        else hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, A, n, sink);
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        float t;
                        CK(hipEventElapsedTime(&t, e0, e1));
                        ms.push_back(t);
                    }
                    std::sort(ms.begin(), ms.end());
                    const double gbps = (double)x * MB / (ms[3] * 1e-3) / 1e9;
                    static const char *pn[] = {"write", "write_nt", "read", "read_nt"};
                    printf("{\"buffer_MB\": %d, \"producer\": \"%s\", \"polluter_nt_MB\": %d, \"consumer\": \"%s\", "
                           "\"consumer_ms\": %.4f, \"consumer_GBps\": %.0f}\n",
                           x, pn[prod], q, cons ? "read_nt" : "read", ms[3], gbps);
                }
    // A mixed consumer (what the row epilogue is): copy the buffer to another one.  Source just
    // written / read (possibly on the die) against a source that has been flushed out by 2 GB of
    // other traffic; destination written with the nt policy.
    for (int x : {32, 64, 128, 192, 256, 512})
        for (int prod = 0; prod < 5; ++prod) {       // 4 = flushed source (no producer)
            const size_t n = (size_t)x * MB / sizeof(f4);
            std::vector<float> ms;
            for (int rep = 0; rep < 7; ++rep) {
                if (prod != 4) hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, P1, P2, polb / sizeof(f4));
                switch (prod) {
                case 0: hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, A, n, (float)rep); break;
                case 1: hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, A, n, (float)rep); break;
                case 2: hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, A, n, sink); break;
                case 3: hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, A, n, sink); break;
                default: hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, P1, P2, polb / sizeof(f4)); break;
                }
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, A, P2, n);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            static const char *pn[] = {"write", "write_nt", "read", "read_nt", "flushed"};
            printf("{\"copy_of_buffer_MB\": %d, \"source_state\": \"%s\", \"ms\": %.4f, "
                   "\"GBps_read_plus_write\": %.0f}\n",
                   x, pn[prod], ms[3], 2.0 * x * MB / (ms[3] * 1e-3) / 1e9);
        }
    // reference: plain streaming rates at 1 GB
    for (int nt = 0; nt < 2; ++nt) {
        std::vector<float> ms;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, 0));
            if (nt) hipLaunchKernelGGL(k_copy<true>, dim3(grid), dim3(256), 0, 0, P1, P2, polb / sizeof(f4));
            else hipLaunchKernelGGL(k_copy<false>, dim3(grid), dim3(256), 0, 0, P1, P2, polb / sizeof(f4));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("{\"copy_1GB\": \"%s\", \"ms\": %.4f, \"GBps_read_plus_write\": %.0f}\n", nt ? "nt" : "default", ms[2],
               2.0 * polb / (ms[2] * 1e-3) / 1e9);
    }
    return 0;
}
