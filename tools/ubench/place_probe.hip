// Placement probe (measurement only; round 5, review item 1): what the emitting row epilogue
// (csc_rows.hip rows_inv_post<16, ..., EMIT, SF = 2>, the `roofline` kernel of bench.py) makes of
// WHERE its three arrays lie.  It is the library's own kernel (this program links
// sporco_amd/csrc/csc_rows.o) at the headline shape 512 x 512, K = 64, N = 32: T (tile-major
// spectrum, read and written in place), V (read), V' (written).
//
//   build:  tools/ubench/build_place_probe.sh          run:  tools/ubench/place_probe [part ...]
//
// Every line of output is one JSON object:
//   part "sep"     the three arrays from separate hipMalloc calls (what the library did up to
//                  round 4), every ordered (V, V') pair out of M buffers;
//   part "dist"    one arena: T at 0, V at a fixed offset, V' at V + span + d for a sweep of d
//                  (4 KiB ... 1 GiB in powers of two, then multiples of 2 MiB, then of 64 MiB);
//   part "tdist"   the same with T moved instead;
//   part "inplace" V' = V;
//   part "vmm"     the arena as ONE physical allocation (hipMemCreate) mapped at a reserved
//                  address, at the driver's recommended granularity.
// ms = average of `reps` launches after one warm-up, both directions of the ping-pong
// (V -> V' and V' -> V) reported separately.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../sporco_amd/csrc/csc_rows.h"

using namespace sporco_amd;

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            std::printf("{\"error\": \"%s\", \"at\": \"%s:%d\"}\n", hipGetErrorString(e_), __FILE__, \
                        __LINE__);                                                                 \
            std::exit(2);                                                                          \
        }                                                                                          \
    } while (0)

static const int H = 512, W = 512, C = 1, N = 32, K = 64;
static const int64_t P = (int64_t)C * N * K, E = (int64_t)H * W * P, EF = (int64_t)H * (W / 2 + 1) * P;
static const size_t VB = sizeof(float) * E, TB = sizeof(cx<float>) * EF;
static cx<float> *twA_d, *twW_d;
static double *part_d;
static hipStream_t st;
static hipEvent_t ev0, ev1;

static float run(cx<float> *t, const float *vin, float *vout, int reps) {
    RowsPostArgs<float> a;
    a.t = t;
    a.t_next = t;
    a.twW = twW_d;
    a.twA = twA_d;
    a.y = a.u = nullptr;
    a.y_out = a.u_out = nullptr;
    a.x = nullptr;
    a.v_in = vin;
    a.v_out = vout;
    a.thr_prev = 0.01f;
    a.scale = 1.f / ((float)H * W);
    a.rlx = 1.8f;
    a.thr = 0.01f;
    a.u_scale = 1.f;
    a.flags = 0;
    a.H = H;
    a.W = W;
    a.C = C;
    a.N = N;
    a.K = K;
    a.dH = a.dW = 1;
    a.P = P;
    a.partials = part_d;
    launch_rows_inv_post<float>(st, a);
    CK(hipEventRecord(ev0, st));
    for (int i = 0; i < reps; ++i) launch_rows_inv_post<float>(st, a);
    CK(hipEventRecord(ev1, st));
    CK(hipEventSynchronize(ev1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, ev0, ev1));
    return ms / reps;
}

static void report(const char *part, const char *label, long long oT, long long oA, long long oB, cx<float> *t,
                   float *va, float *vb, int reps) {
    const float ab = run(t, va, vb, reps), ba = run(t, vb, va, reps);
    std::printf("{\"part\": \"%s\", \"case\": \"%s\", \"off_T\": %lld, \"off_V\": %lld, \"off_Vp\": %lld, "
                "\"ms_ab\": %.4f, \"ms_ba\": %.4f, \"ms\": %.4f}\n",
                part, label, oT, oA, oB, ab, ba, 0.5f * (ab + ba));
    std::fflush(stdout);
}

// plain streaming stores (16 bytes per lane, grid-stride): is a buffer slow to WRITE by itself?
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) fill_kernel(float4 *p_, size_t n4, float v) {
    v4f *p = reinterpret_cast<v4f *>(p_);
    const v4f x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(x, p + i);
}
__global__ void __launch_bounds__(256) fill2_kernel(float4 *p_, float4 *q_, size_t n4, float v) {
    v4f *p = reinterpret_cast<v4f *>(p_), *q = reinterpret_cast<v4f *>(q_);
    const v4f x = {v, v, v, v};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        __builtin_nontemporal_store(x, p + i);
        __builtin_nontemporal_store(x, q + i);
    }
}
__global__ void __launch_bounds__(256) read_kernel(const float4 *p_, size_t n4, float *sink) {
    const v4f *p = reinterpret_cast<const v4f *>(p_);
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const v4f x = __builtin_nontemporal_load(p + i);
        acc += x.x + x.y + x.z + x.w;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void __launch_bounds__(256) copy_kernel(const float4 *p_, float4 *q_, size_t n4) {
    const v4f *p = reinterpret_cast<const v4f *>(p_);
    v4f *q = reinterpret_cast<v4f *>(q_);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(p + i) + 1.0f, q + i);
}
// read two, write two (the epilogue's stream count without its arithmetic)
__global__ void __launch_bounds__(256) copy22_kernel(const float4 *a_, const float4 *b_, float4 *c_, float4 *d_, size_t n4) {
    const v4f *a = reinterpret_cast<const v4f *>(a_), *b = reinterpret_cast<const v4f *>(b_);
    v4f *c = reinterpret_cast<v4f *>(c_), *d = reinterpret_cast<v4f *>(d_);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const v4f x = __builtin_nontemporal_load(a + i), y = __builtin_nontemporal_load(b + i);
        __builtin_nontemporal_store(x + y, c + i);
        __builtin_nontemporal_store(x - y, d + i);
    }
}
// Persistent 16-wave workgroups that load a 256 KiB tile (the column kernel's access pattern: wave w,
// lane k, rows h = 16 h1 + w, 512-byte rows), spend `gap` spins of barrier-locked arithmetic, and
// store it -- to the same place (out0 == out1 == in), or to out0 / out1 by the parity of the tile.
typedef __amdgpu_buffer_rsrc_t PBuf;
typedef float pf2 __attribute__((ext_vector_type(2)));
typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(PBuf(), 0, 0, 0)) pb64;
__global__ void __launch_bounds__(1024) tile_copy_kernel(float *in, float *out0, float *out1, int ntiles, int gap,
                                                         float *sink) {
    extern __shared__ float lds_dummy[];     // 128 KiB requested: one workgroup per CU, as the real kernel
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        PBuf bi = __builtin_amdgcn_make_buffer_rsrc(in + (int64_t)tile * 65536, 0, 262144, 0x00020000);
        // striped output: even tiles -> out0[tile / 2], odd -> out1[tile / 2]; in place when out0 == in
        float *ob = out0 == in ? in + (int64_t)tile * 65536
                               : ((tile & 1) ? out1 : out0) + (int64_t)(out0 == out1 ? tile : tile >> 1) * 65536;
        PBuf bo = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 262144, 0x00020000);
        pf2 v[32];
        const int vo = (w * 64 + lane) * 8;
#pragma unroll
        for (int i = 0; i < 32; ++i)
            v[i] = __builtin_bit_cast(pf2, __builtin_amdgcn_raw_buffer_load_b64(bi, vo, i * 16 * 512, 2));
        if (gap) {
            acc[0] += v[0].x + v[31].y;
            __syncthreads();
            for (int i = 0; i < gap; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = acc[j] * 1.0001f + 0.5f;
            }
            __syncthreads();
            v[0].x += acc[1] * 1e-30f;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            v[i].x += 1.0f;
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pb64, v[i]), bo, vo, i * 16 * 512, 2);
        }
    }
    if (acc[0] == 123.456f) sink[0] = acc[0] + acc[7];
}
template <typename F> static float time_launch(F &&f, int reps) {
    f();
    CK(hipEventRecord(ev0, st));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(ev1, st));
    CK(hipEventSynchronize(ev1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, ev0, ev1));
    return ms / reps;
}

static size_t up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static void sweep(const char *part, char *arena, size_t arena_bytes, int reps) {
    const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
    const size_t span = up(TB > VB ? TB : VB, 64 * MiB);   // 2112 MiB
    std::vector<size_t> ds;
    for (size_t d = 4096; d <= GiB; d *= 2) ds.push_back(d);
    for (size_t j = 1; j <= 40; ++j)
        if ((j & (j - 1)) != 0) ds.push_back(j * 2 * MiB);
    for (size_t j = 3; j <= 15; ++j)
        if ((j & (j - 1)) != 0) ds.push_back(j * 64 * MiB);
    ds.insert(ds.begin(), 0);
    // layout: [T: span + 1 GiB slack][V: span][V': span + 1 GiB slack]
    const size_t oT0 = 0, oA = span + GiB, oB0 = oA + span;
    if (oB0 + span + GiB > arena_bytes) {
        std::printf("{\"error\": \"arena too small\"}\n");
        return;
    }
    if (!std::strcmp(part, "dist") || !std::strcmp(part, "vmm")) {
        for (size_t d : ds)
            report(part, "Vp_moves", (long long)oT0, (long long)oA, (long long)(oB0 + d),
                   reinterpret_cast<cx<float> *>(arena + oT0), reinterpret_cast<float *>(arena + oA),
                   reinterpret_cast<float *>(arena + oB0 + d), reps);
    }
    if (!std::strcmp(part, "tdist") || !std::strcmp(part, "vmm")) {
        for (size_t d : ds)
            report(part, "T_moves", (long long)(oT0 + d), (long long)oA, (long long)oB0,
                   reinterpret_cast<cx<float> *>(arena + oT0 + d), reinterpret_cast<float *>(arena + oA),
                   reinterpret_cast<float *>(arena + oB0), reps);
    }
    if (!std::strcmp(part, "inplace"))
        report(part, "Vp_is_V", (long long)oT0, (long long)oA, (long long)oA,
               reinterpret_cast<cx<float> *>(arena + oT0), reinterpret_cast<float *>(arena + oA),
               reinterpret_cast<float *>(arena + oA), reps);
}

int main(int argc, char **argv) {
    const int reps = std::getenv("PROBE_REPS") ? std::atoi(std::getenv("PROBE_REPS")) : 3;
    std::vector<std::string> parts;
    for (int i = 1; i < argc; ++i) parts.push_back(argv[i]);
    if (parts.empty()) parts = {"sep", "dist", "tdist", "inplace", "vmm"};
    CK(hipSetDevice(0));
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    {
        std::vector<cx<float>> ta(W), tw(W);
        rows_twiddles<float>(W, ta.data());
        for (int t = 0; t < W; ++t) {
            const double ang = -2.0 * M_PI * t / W;
            tw[t] = mk<float>((float)std::cos(ang), (float)std::sin(ang));
        }
        CK(hipMalloc((void **)&twA_d, sizeof(cx<float>) * W));
        CK(hipMalloc((void **)&twW_d, sizeof(cx<float>) * W));
        CK(hipMemcpy(twA_d, ta.data(), sizeof(cx<float>) * W, hipMemcpyHostToDevice));
        CK(hipMemcpy(twW_d, tw.data(), sizeof(cx<float>) * W, hipMemcpyHostToDevice));
        CK(hipMalloc((void **)&part_d, sizeof(double) * 8 * H * ((P + 127) / 128)));
    }
    const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
    std::printf("{\"part\": \"info\", \"V_bytes\": %zu, \"T_bytes\": %zu, \"reps\": %d}\n", VB, TB, reps);
    for (const std::string &part : parts) {
        if (part == "sep") {
            // M separately allocated, 64 MiB-aligned buffers (big_alloc of round 4); T is buffer 0
            const int M = std::getenv("PROBE_M") ? std::atoi(std::getenv("PROBE_M")) : 7;
            std::vector<char *> base(M), al(M);
            for (int i = 0; i < M; ++i) {
                CK(hipMalloc((void **)&base[i], TB + 64 * MiB));
                al[i] = reinterpret_cast<char *>(up(reinterpret_cast<size_t>(base[i]), 64 * MiB));
                CK(hipMemset(al[i], 0, TB));
            }
            for (int i = 0; i < M; ++i)
                std::printf("{\"part\": \"sep\", \"buffer\": %d, \"va\": \"0x%zx\"}\n", i, reinterpret_cast<size_t>(al[i]));
            for (int ti = 0; ti < 2; ++ti)
                for (int a = 0; a < M; ++a)
                    for (int b = a + 1; b < M; ++b) {
                        if (a == ti || b == ti) continue;
                        char lab[64];
                        std::snprintf(lab, sizeof lab, "T%d_V%d_Vp%d", ti, a, b);
                        report("sep", lab, (long long)(al[ti] - al[0]), (long long)(al[a] - al[0]),
                               (long long)(al[b] - al[0]), reinterpret_cast<cx<float> *>(al[ti]),
                               reinterpret_cast<float *>(al[a]), reinterpret_cast<float *>(al[b]), reps);
                    }
            for (int i = 0; i < M; ++i) CK(hipFree(base[i]));
        } else if (part == "matrix") {
            // M separately allocated buffers: fill / read time of each alone, then the epilogue for
            // every (T, V') with V read from a third buffer -- is "fast" a property of the buffer
            // written, or of how it lies relative to T?
            const int M = std::getenv("PROBE_M") ? std::atoi(std::getenv("PROBE_M")) : 16;
            std::vector<char *> base(M), al(M);
            for (int i = 0; i < M; ++i) {
                CK(hipMalloc((void **)&base[i], TB + 64 * MiB));
                al[i] = reinterpret_cast<char *>(up(reinterpret_cast<size_t>(base[i]), 64 * MiB));
                CK(hipMemset(al[i], 0, TB));
            }
            float *sink;
            CK(hipMalloc((void **)&sink, 64));
            for (int i = 0; i < M; ++i) {
                const size_t n4 = VB / 16;
                const float wms = time_launch([&] { hipLaunchKernelGGL(fill_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<float4 *>(al[i]), n4, 0.f); }, reps);
                const float rms = time_launch([&] { hipLaunchKernelGGL(read_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const float4 *>(al[i]), n4, sink); }, reps);
                std::printf("{\"part\": \"matrix\", \"buffer\": %d, \"base_va\": \"0x%zx\", \"va\": \"0x%zx\", \"fill_ms\": %.4f, \"fill_GBps\": %.0f, \"read_ms\": %.4f, \"read_GBps\": %.0f}\n",
                            i, reinterpret_cast<size_t>(base[i]), reinterpret_cast<size_t>(al[i]), wms, VB / wms * 1e-6, rms, VB / rms * 1e-6);
            }
            for (int i = 0; i + 1 < M; i += 1) {
                const int j = (i + M / 2) % M;
                const size_t n4 = VB / 16;
                const float wms = time_launch([&] { hipLaunchKernelGGL(fill2_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<float4 *>(al[i]), reinterpret_cast<float4 *>(al[j]), n4, 0.f); }, reps);
                std::printf("{\"part\": \"matrix\", \"fill2\": [%d, %d], \"ms\": %.4f, \"GBps\": %.0f}\n", i, j, wms, 2.0 * VB / wms * 1e-6);
            }
            for (int ti = 0; ti < M; ++ti) {
                std::printf("{\"part\": \"matrix\", \"T\": %d, \"ms_by_Vp\": [", ti);
                for (int b = 0; b < M; ++b) {
                    int a = (b + 1) % M;
                    if (a == ti) a = (a + 1) % M;
                    float ms = -1.f;
                    if (b != ti)
                        ms = run(reinterpret_cast<cx<float> *>(al[ti]), reinterpret_cast<float *>(al[a]),
                                 reinterpret_cast<float *>(al[b]), reps);
                    std::printf("%s%.3f", b ? ", " : "", ms);
                }
                std::printf("]}\n");
                std::fflush(stdout);
            }
            for (int i = 0; i < M; ++i) CK(hipFree(base[i]));
        } else if (part == "calib") {
            // calibration of the library's placement probe (csc_kernels.h launch_place_probe) against
            // the kernel it stands for: T = buffer 0, every later buffer as V'
            const int M = std::getenv("PROBE_M") ? std::atoi(std::getenv("PROBE_M")) : 24;
            std::vector<char *> base(M), al(M);
            hipEvent_t ma, mb;
            CK(hipEventCreate(&ma));
            CK(hipEventCreate(&mb));
            for (int i = 0; i < M; ++i) {
                CK(hipEventRecord(ma, st));
                CK(hipMalloc((void **)&base[i], TB + 64 * MiB));
                al[i] = reinterpret_cast<char *>(up(reinterpret_cast<size_t>(base[i]), 64 * MiB));
            }
            auto rate = [&](void *a, void *b, int64_t n16, bool fa, bool fb) {
                const float ms = time_launch([&] { launch_place_probe(st, a, b, n16, n16, 0, 0, fa, fb); }, reps);
                return 2.0 * n16 * 16 / ms * 1e-6;      // GB/s of stores
            };
            const int64_t n16 = VB / 16;
            const double ref_rw = rate(al[0], al[0] + VB / 2, n16 / 2, false, false);
            std::printf("{\"part\": \"calib\", \"ref_rewrite_halves_of_T_GBps\": %.0f}\n", ref_rw);
            for (int j = 1; j < M; ++j) {
                const double rw = rate(al[0], al[j], n16, false, false), rf = rate(al[0], al[j], n16, false, true),
                             ff = rate(al[0] , al[j], n16, true, true);
                const float ms = run(reinterpret_cast<cx<float> *>(al[0]), reinterpret_cast<float *>(al[j == 1 ? 2 : 1]),
                                     reinterpret_cast<float *>(al[j]), reps);
                std::printf("{\"part\": \"calib\", \"cand\": %d, \"epilogue_ms\": %.4f, \"rewrite_rewrite\": %.0f, "
                            "\"rewrite_fill\": %.0f, \"fill_fill\": %.0f, \"rr_over_ref\": %.3f}\n", j, ms, rw, rf, ff, rw / ref_rw);
            }
            // cost of allocating and freeing a buffer of this size
            {
                char *x = nullptr;
                const auto t0 = std::chrono::steady_clock::now();
                CK(hipMalloc((void **)&x, TB + 64 * MiB));
                const auto t1 = std::chrono::steady_clock::now();
                CK(hipFree(x));
                const auto t2 = std::chrono::steady_clock::now();
                std::printf("{\"part\": \"calib\", \"hipMalloc_ms\": %.3f, \"hipFree_ms\": %.3f}\n",
                            std::chrono::duration<double, std::milli>(t1 - t0).count(),
                            std::chrono::duration<double, std::milli>(t2 - t1).count());
            }
            for (int i = 0; i < M; ++i) CK(hipFree(base[i]));
        } else if (part == "lockstep") {
            // does the library's probe depend on the distance between its two arrays?  one buffer,
            // the second stream `off` behind the first
            char *base = nullptr;
            CK(hipMalloc((void **)&base, 6 * GiB + 64 * MiB));
            char *x = reinterpret_cast<char *>(up(reinterpret_cast<size_t>(base), 64 * MiB));
            const int64_t n16 = (int64_t)(GiB / 16);
            for (size_t off : {1024 * MiB, 1028 * MiB, 1032 * MiB, 1040 * MiB, 1056 * MiB, 1088 * MiB, 1152 * MiB,
                               1280 * MiB, 1536 * MiB, 2048 * MiB, 2052 * MiB, 3072 * MiB, 4096 * MiB, 1024 * MiB + 4096,
                               1024 * MiB + 65536, 1025 * MiB}) {
                const float ms = time_launch([&] { launch_place_probe(st, x, x + off, n16, n16, 0, 0, false, false); }, reps);
                std::printf("{\"part\": \"lockstep\", \"off_MiB\": %.3f, \"rewrite_rewrite_GBps\": %.0f}\n",
                            (double)off / MiB, 2.0 * n16 * 16 / ms * 1e-6);
            }
            CK(hipFree(base));
        } else if (part == "stripe") {
            // one write stream into one region, or spread over two?  (the column kernel rewrites the
            // spectrum in place: its store phases run at the single-region write rate)
            const int M = std::getenv("PROBE_M") ? std::atoi(std::getenv("PROBE_M")) : 14;
            std::vector<char *> base(M), al(M);
            for (int i = 0; i < M; ++i) {
                CK(hipMalloc((void **)&base[i], TB + 64 * MiB));
                al[i] = reinterpret_cast<char *>(up(reinterpret_cast<size_t>(base[i]), 64 * MiB));
            }
            const int64_t n16 = VB / 16;
            auto rate = [&](int a, int b) {
                const float ms = time_launch([&] { launch_place_probe(st, al[a], al[b], n16, n16, 0, 0, false, false); }, reps);
                return 2.0 * n16 * 16 / ms * 1e-6;
            };
            // classify against buffer 0
            int same = -1, other = -1;
            double rs = 1e30, ro = 0;
            for (int j = 1; j < M; ++j) {
                const double r = rate(0, j);
                if (r < rs) { rs = r; same = j; }
                if (r > ro) { ro = r; other = j; }
            }
            std::printf("{\"part\": \"stripe\", \"same_region_buffer\": %d, \"rate\": %.0f, \"other_region_buffer\": %d, \"rate_other\": %.0f}\n",
                        same, rs, other, ro);
            CK(hipFuncSetAttribute((const void *)&tile_copy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
            float *sink;
            CK(hipMalloc((void **)&sink, 64));
            const int ntiles = 8224;
            auto tc = [&](float *in, float *o0, float *o1, int gap) {
                const float ms = time_launch([&] { hipLaunchKernelGGL(tile_copy_kernel, dim3(256), dim3(1024), 131072, st, in, o0, o1, ntiles, gap, sink); }, reps);
                return ms;
            };
            float *A = reinterpret_cast<float *>(al[0]), *A2 = reinterpret_cast<float *>(al[same]), *B = reinterpret_cast<float *>(al[other]);
            for (int gap : {0, 300, 600, 900, 1500}) {
                std::printf("{\"part\": \"stripe\", \"gap\": %d, \"in_place_ms\": %.4f, \"out_same_region_ms\": %.4f, "
                            "\"out_other_region_ms\": %.4f, \"out_striped_same_plus_other_ms\": %.4f, \"out_striped_same_same_ms\": %.4f}\n",
                            gap, tc(A, A, A, gap), tc(A, A2, A2, gap), tc(A, B, B, gap), tc(A, A2, B, gap),
                            tc(A, A2, A2 + (size_t)4112 * 65536, gap));
            }
            for (int i = 0; i < M; ++i) CK(hipFree(base[i]));
        } else if (part == "map") {
            // G chunks of 1 GiB in allocation order, each classified by the two-stream fill against
            // chunk 0 (and against the first chunk found to differ from chunk 0): the region map
            const int G = std::getenv("PROBE_G") ? std::atoi(std::getenv("PROBE_G")) : 200;
            std::vector<char *> ch(G);
            for (int i = 0; i < G; ++i) CK(hipMalloc((void **)&ch[i], GiB));
            const size_t n4 = GiB / 16;
            std::vector<float> t0(G, 0.f), t1(G, 0.f);
            int other = -1;
            for (int i = 1; i < G; ++i) {
                t0[i] = time_launch([&] { hipLaunchKernelGGL(fill2_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<float4 *>(ch[0]), reinterpret_cast<float4 *>(ch[i]), n4, 0.f); }, reps);
                if (other < 0 && i > 1 && t0[i] < 0.85f * t0[1]) other = i;
            }
            if (other > 0)
                for (int i = 0; i < G; ++i)
                    if (i != other)
                        t1[i] = time_launch([&] { hipLaunchKernelGGL(fill2_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<float4 *>(ch[other]), reinterpret_cast<float4 *>(ch[i]), n4, 0.f); }, reps);
            std::printf("{\"part\": \"map\", \"chunks\": %d, \"other\": %d, \"first_va\": \"0x%zx\", \"last_va\": \"0x%zx\", \"fill2_vs_chunk0_us\": [", G, other,
                        reinterpret_cast<size_t>(ch[0]), reinterpret_cast<size_t>(ch[G - 1]));
            for (int i = 0; i < G; ++i) std::printf("%s%.0f", i ? ", " : "", 1000.f * t0[i]);
            std::printf("], \"fill2_vs_other_us\": [");
            for (int i = 0; i < G; ++i) std::printf("%s%.0f", i ? ", " : "", 1000.f * t1[i]);
            std::printf("]}\n");
            // copies: within the region of chunk 0, across regions, in place
            if (other > 0) {
                int same = -1;
                for (int i = 1; i < G; ++i)
                    if (i != other && t0[i] > 0.92f * t0[1]) same = i;
                const float cs = time_launch([&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const float4 *>(ch[0]), reinterpret_cast<float4 *>(ch[same]), n4); }, reps);
                const float cx_ = time_launch([&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const float4 *>(ch[0]), reinterpret_cast<float4 *>(ch[other]), n4); }, reps);
                const float ci = time_launch([&] { hipLaunchKernelGGL(copy_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const float4 *>(ch[0]), reinterpret_cast<float4 *>(ch[0]), n4); }, reps);
                std::printf("{\"part\": \"map\", \"copy_GBps\": {\"same_region\": %.0f, \"cross_region\": %.0f, \"in_place\": %.0f}}\n",
                            2.0 * GiB / cs * 1e-6, 2.0 * GiB / cx_ * 1e-6, 2.0 * GiB / ci * 1e-6);
                // two reads + two writes: A = chunk 0's region (chunks 0, same, and a third), B = other's
                int same2 = -1, other2 = -1;
                for (int i = 1; i < G; ++i) {
                    if (i != other && i != same && t0[i] > 0.92f * t0[1] && same2 < 0) same2 = i;
                    if (i != other && t0[i] < 0.85f * t0[1] && other2 < 0) other2 = i;
                }
                auto c22 = [&](int a, int b, int c, int d) {
                    return time_launch([&] { hipLaunchKernelGGL(copy22_kernel, dim3(256 * 8), dim3(256), 0, st, reinterpret_cast<const float4 *>(ch[a]), reinterpret_cast<const float4 *>(ch[b]), reinterpret_cast<float4 *>(ch[c]), reinterpret_cast<float4 *>(ch[d]), n4); }, reps);
                };
                if (same2 > 0 && other2 > 0) {
                    const float w_same = c22(0, same, 0, same), w_cross = c22(0, other, 0, other),
                                w_far = c22(0, same, other, other2), w_mixed = c22(0, other, same, other2);
                    std::printf("{\"part\": \"map\", \"rw22_GBps\": {\"all_in_A_inplace\": %.0f, \"A_and_B_inplace\": %.0f, "
                                "\"read_AA_write_BB\": %.0f, \"read_AB_write_AB\": %.0f}}\n",
                                4.0 * GiB / w_same * 1e-6, 4.0 * GiB / w_cross * 1e-6, 4.0 * GiB / w_far * 1e-6, 4.0 * GiB / w_mixed * 1e-6);
                }
            }
            for (int i = 0; i < G; ++i) CK(hipFree(ch[i]));
        } else if (part == "dist" || part == "tdist" || part == "inplace") {
            const size_t bytes = 3 * up(TB, 64 * MiB) + 2 * GiB + 64 * MiB;
            char *base = nullptr;
            CK(hipMalloc((void **)&base, bytes + 64 * MiB));
            char *arena = reinterpret_cast<char *>(up(reinterpret_cast<size_t>(base), 64 * MiB));
            CK(hipMemset(arena, 0, bytes));
            std::printf("{\"part\": \"%s\", \"arena_va\": \"0x%zx\", \"bytes\": %zu}\n", part.c_str(),
                        reinterpret_cast<size_t>(arena), bytes);
            sweep(part.c_str(), arena, bytes, reps);
            CK(hipFree(base));
        } else if (part == "vmm") {
            hipMemAllocationProp prop;
            std::memset(&prop, 0, sizeof prop);
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            size_t gmin = 0, grec = 0;
            CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
            CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
            std::printf("{\"part\": \"vmm\", \"granularity_min\": %zu, \"granularity_recommended\": %zu}\n", gmin, grec);
            const size_t g = grec > gmin ? grec : gmin;
            const size_t bytes = up(3 * up(TB, 64 * MiB) + 2 * GiB + 64 * MiB, g > GiB ? g : GiB);
            hipMemGenericAllocationHandle_t hnd;
            CK(hipMemCreate(&hnd, bytes, &prop, 0));
            void *va = nullptr;
            CK(hipMemAddressReserve(&va, bytes, GiB, nullptr, 0));
            CK(hipMemMap(va, bytes, 0, hnd, 0));
            hipMemAccessDesc acc;
            std::memset(&acc, 0, sizeof acc);
            acc.location.type = hipMemLocationTypeDevice;
            acc.location.id = 0;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, bytes, &acc, 1));
            CK(hipMemset(va, 0, bytes));
            std::printf("{\"part\": \"vmm\", \"arena_va\": \"0x%zx\", \"bytes\": %zu}\n", reinterpret_cast<size_t>(va), bytes);
            sweep("vmm", static_cast<char *>(va), bytes, reps);
            CK(hipMemUnmap(va, bytes));
            CK(hipMemRelease(hnd));
            CK(hipMemAddressFree(va, bytes));
        }
    }
    return 0;
}
