// tests/hostsim/hostsim_runtime.cpp -- TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h).
//
// Fiber-based execution of one workgroup at a time: every GPU thread is a
// ucontext fiber; __syncthreads / wave shuffles yield to a round-robin
// scheduler and are released by arrival counters, so barrier semantics (and
// barrier bugs: divergent barriers deadlock and are reported) match the GPU.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace {
constexpr int kMaxCoop = 8;               // workgroups that may run side by side (set_coop)
constexpr size_t kLdsBytes = 160 * 1024;
alignas(16) unsigned char lds_store[kMaxCoop][kLdsBytes];
}  // namespace

struct hostsim_event {
    std::chrono::steady_clock::time_point t;
};

namespace hostsim {

namespace {

constexpr size_t kStack = 128 * 1024;

// Minimal x86-64 System V context switch (callee-saved registers + stack
// pointer); ucontext's swapcontext costs two sigprocmask syscalls per switch.
extern "C" void hostsim_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl hostsim_switch
.type hostsim_switch,@function
hostsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hostsim_switch,.-hostsim_switch
)");

struct Fiber {
    void *sp = nullptr;
    char *stack = nullptr;
    bool done = false;
};

std::vector<Fiber> fibers;
void *sched_sp = nullptr;
int cur = -1;          // fiber index = slot * nthreads + thread
int cur_slot = 0;      // which of the side-by-side workgroups the running fiber belongs to
int nthreads = 0;
int coop = 1;          // workgroups of consecutive index run side by side (next launch only)
const std::function<void()> *body_fn = nullptr;

int bar_arrived[kMaxCoop], bar_gen[kMaxCoop];
int wave_arrived[kMaxCoop][32], wave_gen[kMaxCoop][32];
unsigned long progress = 0;
alignas(16) unsigned char slots[kMaxCoop][2048][16];

void yield() { hostsim_switch(&fibers[cur].sp, sched_sp); }

void trampoline() {
    (*body_fn)();
    fibers[cur].done = true;
    ++progress;
    hostsim_switch(&fibers[cur].sp, sched_sp);
    std::abort();  // a finished fiber is never resumed
}

void prepare(Fiber &f) {
    // stack image consumed by hostsim_switch: r15 r14 r13 r12 rbx rbp, return
    // address (trampoline), one pad word so that trampoline starts with the ABI's
    // call-site alignment (rsp % 16 == 8)
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void **sp = (void **)top - 8;
    for (int i = 0; i < 6; ++i) sp[i] = nullptr;
    sp[6] = (void *)&trampoline;
    sp[7] = nullptr;
    f.sp = sp;
    f.done = false;
}

}  // namespace

int block_threads() { return nthreads; }
void *shuffle_slot(int tid) { return slots[cur_slot][tid]; }
void *lds_base() { return lds_store[cur_slot]; }
void set_coop(int n) {
    if (n < 1 || n > kMaxCoop) {
        std::fprintf(stderr, "hostsim: at most %d workgroups side by side\n", kMaxCoop);
        std::abort();
    }
    coop = n;
}
// a thread that polls memory written by another workgroup gives the others a turn
void spin_pause() {
    ++progress;      // (a poll is not a deadlock: the writer may need many rounds to get there)
    yield();
}

void syncthreads() {
    const int s = cur_slot;
    const int g = bar_gen[s];
    if (++bar_arrived[s] == nthreads) {
        bar_arrived[s] = 0;
        ++bar_gen[s];
        ++progress;
    } else {
        while (bar_gen[s] == g) yield();
    }
}

void wave_sync() {
    const int s = cur_slot;
    const int w = (int)threadIdx.x / 64;
    const int wsize = (nthreads - w * 64) < 64 ? (nthreads - w * 64) : 64;
    const int g = wave_gen[s][w];
    if (++wave_arrived[s][w] == wsize) {
        wave_arrived[s][w] = 0;
        ++wave_gen[s][w];
        ++progress;
    } else {
        while (wave_gen[s][w] == g) yield();
    }
}

void run_grid(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body) {
    if (shmem > kLdsBytes) {
        std::fprintf(stderr, "hostsim: dynamic LDS request %zu exceeds 160 KiB\n", shmem);
        std::abort();
    }
    nthreads = (int)(block.x * block.y * block.z);
    if (nthreads > 1024 || block.y != 1 || block.z != 1) {
        std::fprintf(stderr, "hostsim: unsupported block shape\n");
        std::abort();
    }
    const int group = coop;
    coop = 1;
    if ((int)fibers.size() < nthreads * group) {
        const size_t old = fibers.size();
        fibers.resize((size_t)nthreads * group);
        for (size_t i = old; i < fibers.size(); ++i) fibers[i].stack = (char *)std::malloc(kStack);
    }
    body_fn = &body;
    blockDim = block;
    gridDim = grid;
    // workgroups in x-fastest order, `group` of them side by side (1 unless set_coop was called)
    const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
    dim3 idx[kMaxCoop];
    for (unsigned long b0 = 0; b0 < nblocks; b0 += group) {
        const int nb = (int)((nblocks - b0) < (unsigned long)group ? (nblocks - b0) : group);
        for (int s = 0; s < nb; ++s) {
            const unsigned long b = b0 + s;
            idx[s] = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y),
                          (unsigned)(b / ((unsigned long)grid.x * grid.y)));
            bar_arrived[s] = 0;
            for (int w = 0; w < 32; ++w) wave_arrived[s][w] = 0;
            for (int t = 0; t < nthreads; ++t) prepare(fibers[s * nthreads + t]);
        }
        int live = nb * nthreads;
        while (live > 0) {
            const unsigned long before = progress;
            live = 0;
            for (int f = 0; f < nb * nthreads; ++f) {
                if (fibers[f].done) continue;
                cur = f;
                cur_slot = f / nthreads;
                blockIdx = idx[cur_slot];
                threadIdx = dim3((unsigned)(f % nthreads), 0, 0);
                hostsim_switch(&sched_sp, fibers[f].sp);
                if (!fibers[f].done) ++live;
            }
            if (live > 0 && progress == before) {
                std::fprintf(stderr,
                             "hostsim: deadlock -- %d threads wait at a barrier that the "
                             "others never reach (divergent __syncthreads/shuffle)\n",
                             live);
                std::abort();
            }
        }
    }
    cur = -1;
    cur_slot = 0;
}

}  // namespace hostsim

// ---------------------------------------------------------------------------
// runtime API: "device" memory is host memory, streams are synchronous
// ---------------------------------------------------------------------------
hipError_t hipMalloc(void **p, size_t n) {
    void *q = nullptr;
    if (posix_memalign(&q, 256, n ? n : 1) != 0) return hipErrorOutOfMemory;
    std::memset(q, 0xCD, n);  // poison: catches reads of never-written device memory
    *p = q;
    return hipSuccess;
}
hipError_t hipMemGetInfo(size_t *free_bytes, size_t *total_bytes) {
    *free_bytes = *total_bytes = (size_t)1 << 40;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void *p) { return hipFree(p); }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) {
    std::memmove(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) {
    return hipMemcpy(d, s, n, k);
}
hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width,
                            size_t height, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < height; ++r)
        std::memmove((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemset(void *d, int v, size_t n) {
    std::memset(d, v, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
hipError_t hipStreamCreate(hipStream_t *st) {
    *st = (hipStream_t)(uintptr_t)0x1;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) {
    *d = 0;
    return hipSuccess;
}
hipError_t hipGetDeviceCount(int *n) {
    *n = 1;
    return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->name, "hostsim (CPU fiber simulator, tests only)");
    p->multiProcessorCount = 1;
    p->totalGlobalMem = (size_t)1 << 34;
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hostsim error"; }
hipError_t hipEventCreate(hipEvent_t *e) {
    *e = new hostsim_event;
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e) {
    delete e;
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
    e->t = std::chrono::steady_clock::now();
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }


// ---------------------------------------------------------------------------------------------
// Stand-in for librccl (csc_comm.hip opens whatever SPORCO_AMD_RCCL_LIB names; tests/conftest.py
// points it at this library): single-rank communicators, identity collectives -- what runs is the
// sharded code path of the library around them.
// ---------------------------------------------------------------------------------------------
extern "C" {
struct HostsimNcclId {
    char internal[128];
};
int ncclGetUniqueId(HostsimNcclId *id) {
    std::memset(id->internal, 0, sizeof id->internal);
    return 0;
}
int ncclCommInitRank(void **comm, int world, HostsimNcclId, int rank) {
    if (world != 1 || rank != 0) return 5;      // (ncclInvalidArgument)
    *comm = reinterpret_cast<void *>(0x1);
    return 0;
}
int ncclCommDestroy(void *) { return 0; }
int ncclAllReduce(const void *in, void *out, size_t count, int dtype, int, void *, hipStream_t) {
    if (in != out) std::memcpy(out, in, count * (dtype == 7 ? 4 : 8));
    return 0;
}
const char *ncclGetErrorString(int rc) {
    return rc == 5 ? "the CPU simulator build has single-rank communicators only" : "hostsim nccl stand-in";
}
}
