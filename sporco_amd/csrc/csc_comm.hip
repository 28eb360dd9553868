// csc_comm.hip -- RCCL inside the C ABI (include/sporco_amd.h: sporco_amd_comm_*): the image
// shards of one node sum their 16 per-iteration scalars -- and, for dictionary learning, a
// dictionary-sized gradient -- with an all-reduce the library enqueues itself on the solver's
// stream, instead of calling back into the host program every iteration.
//
// librccl is opened at run time: libsporco_amd.so keeps libamdhip64 as its only link-time
// dependency, and a box without RCCL still runs everything single-GPU.  The few declarations
// needed are restated here (rccl.h: ncclUniqueId is 128 opaque bytes, ncclComm_t an opaque
// pointer, ncclSum = 0, ncclMax = 2, ncclFloat32 = 7, ncclFloat64 = 8).
//
// (The CPU simulator of the test-suite brings its own stand-in for librccl -- single-rank,
// identity collectives, tests/hostsim/hostsim_runtime.cpp -- and points SPORCO_AMD_RCCL_LIB at
// it: this file has one code path.)
#include "csc_impl.h"

#include <dlfcn.h>

using namespace sporco_amd;

namespace {

struct NcclId {
    char internal[SPORCO_AMD_COMM_ID_BYTES];
};
typedef void *NcclComm;
typedef int (*GetUniqueIdFn)(NcclId *);
typedef int (*CommInitRankFn)(NcclComm *, int, NcclId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, NcclComm, hipStream_t);
typedef const char *(*GetErrorStringFn)(int);

struct Rccl {
    void *lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
    std::string tried;
};

Rccl &rccl() {
    static Rccl r;
    if (r.lib) return r;
    std::vector<std::string> names;
    if (const char *e = std::getenv("SPORCO_AMD_RCCL_LIB")) names.push_back(e);
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    names.push_back("/opt/rocm/lib/librccl.so");
    for (const auto &n : names) {
        r.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
        r.tried += (r.tried.empty() ? "" : ", ") + n;
    }
    if (!r.lib)
        throw Error(SPORCO_AMD_EUNSUPPORTED, "RCCL is not available (tried " + r.tried + ")");
    r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
    r.all_reduce = (AllReduceFn)dlsym(r.lib, "ncclAllReduce");
    r.error_string = (GetErrorStringFn)dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce) {
        r.lib = nullptr;
        throw Error(SPORCO_AMD_EUNSUPPORTED, "librccl lacks the expected entry points");
    }
    return r;
}
void nccl_check(int rc, const char *what) {
    if (rc == 0) return;
    Rccl &r = rccl();
    throw Error(SPORCO_AMD_EHIP, std::string(what) + ": " +
                                     (r.error_string ? r.error_string(rc) : "RCCL error " + std::to_string(rc)));
}

}  // namespace

struct sporco_amd_comm {
    void *comm = nullptr;
    int rank = 0, world = 1, device = 0;
    double *stage = nullptr;     // 64 doubles of device memory for allreduce_host
};

namespace sporco_amd {
// the thunk sporco_amd_csc_admm_run calls between the local sums and the control update
void comm_reduce_sums(void *user, double *sums_dev, hipStream_t st) {
    sporco_amd_comm *c = static_cast<sporco_amd_comm *>(user);
    if (!c) return;
    // (a single rank still goes through RCCL: the one-rank test exercises the real call)
    nccl_check(rccl().all_reduce(sums_dev, sums_dev, 16, 8 /* ncclFloat64 */, 0 /* ncclSum */, c->comm, st),
               "ncclAllReduce");
}
}  // namespace sporco_amd

extern "C" {

int sporco_amd_comm_unique_id(void *id128) {
    SA_API_BEGIN
    SA_REQUIRE(id128 != nullptr, "id128 is null");
    NcclId id;
    nccl_check(rccl().get_unique_id(&id), "ncclGetUniqueId");
    std::memcpy(id128, id.internal, SPORCO_AMD_COMM_ID_BYTES);
    SA_API_END
}

int sporco_amd_comm_create(const void *id128, int32_t rank, int32_t world, int32_t device,
                           sporco_amd_comm_t *out) {
    SA_API_BEGIN
    SA_REQUIRE(id128 && out, "null argument");
    SA_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world out of range");
    std::unique_ptr<sporco_amd_comm> c(new sporco_amd_comm);
    c->rank = rank;
    c->world = world;
    c->device = device;
    SA_HIP(hipSetDevice(device));
    NcclId id;
    std::memcpy(id.internal, id128, SPORCO_AMD_COMM_ID_BYTES);
    nccl_check(rccl().comm_init_rank(&c->comm, world, id, rank), "ncclCommInitRank");
    SA_HIP(hipMalloc((void **)&c->stage, sizeof(double) * 64));
    *out = c.release();
    SA_API_END
}

int sporco_amd_comm_destroy(sporco_amd_comm_t c) {
    SA_API_BEGIN
    if (c) {
        (void)hipSetDevice(c->device);
        if (c->comm) (void)rccl().comm_destroy(c->comm);
        if (c->stage) (void)hipFree(c->stage);
        delete c;
    }
    SA_API_END
}

int sporco_amd_comm_info(sporco_amd_comm_t c, int32_t *rank, int32_t *world) {
    SA_API_BEGIN
    SA_REQUIRE(c != nullptr, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    SA_API_END
}

int sporco_amd_comm_allreduce(sporco_amd_comm_t c, void *buf_dev, int64_t count, int dtype, int op,
                              void *stream) {
    SA_API_BEGIN
    SA_REQUIRE(c != nullptr && buf_dev != nullptr && count >= 0, "bad argument");
    SA_REQUIRE(dtype == SPORCO_AMD_F32 || dtype == SPORCO_AMD_F64, "dtype must be SPORCO_AMD_F32 or _F64");
    SA_REQUIRE(op == SPORCO_AMD_COMM_SUM || op == SPORCO_AMD_COMM_MAX, "op must be SUM or MAX");
    if (count > 0) {
        SA_HIP(hipSetDevice(c->device));
        nccl_check(rccl().all_reduce(buf_dev, buf_dev, (size_t)count, dtype == SPORCO_AMD_F32 ? 7 : 8, op,
                                     c->comm, (hipStream_t)stream),
                   "ncclAllReduce");
    }
    SA_API_END
}

int sporco_amd_comm_allreduce_host(sporco_amd_comm_t c, double *vals, int32_t n, int op) {
    SA_API_BEGIN
    SA_REQUIRE(c != nullptr && vals != nullptr && n >= 0 && n <= 64, "bad argument (n <= 64)");
    SA_REQUIRE(op == SPORCO_AMD_COMM_SUM || op == SPORCO_AMD_COMM_MAX, "op must be SUM or MAX");
    if (n > 0) {
        SA_HIP(hipSetDevice(c->device));
        SA_HIP(hipMemcpy(c->stage, vals, sizeof(double) * n, hipMemcpyHostToDevice));
        nccl_check(rccl().all_reduce(c->stage, c->stage, (size_t)n, 8, op, c->comm, (hipStream_t) nullptr),
                   "ncclAllReduce");
        SA_HIP(hipStreamSynchronize(nullptr));
        SA_HIP(hipMemcpy(vals, c->stage, sizeof(double) * n, hipMemcpyDeviceToHost));
    }
    SA_API_END
}

int sporco_amd_csc_set_comm(sporco_amd_csc_t h, sporco_amd_comm_t c) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->comm_user = c;
    SA_API_END
}

}  // extern "C"
