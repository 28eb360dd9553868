// csc_impl.h -- what the translation units behind the C ABI share: transfer counting, the
// per-kernel event profiler, the type-erased solver interface (CscBase) that the extern "C"
// glue (csc_abi.hip) talks to, and the error-to-return-code macros.  The solver itself
// (template Csc<T>, csc_api.hip + api_*.inc) is only visible through make_csc().
#pragma once
#include "../../include/sporco_amd.h"

#include <cmath>
#include <cstdlib>
#include <memory>
#include <string>
#include <utility>
#include <cstring>
#include <vector>

#include "common.h"
#include "csc_fused.h"
#include "csc_kernels.h"
#include "csc_pgm.h"
#include "csc_rows.h"
#include "fft.h"

#include <atomic>

namespace sporco_amd {

constexpr long kAllocAlignMb = 64;    // alignment of the large device buffers, MiB (csc_api.hip big_alloc)

extern thread_local std::string g_last_error;

// Every host <-> device copy of this file is counted (sporco_amd_transfer_stats): the claim
// "the pipeline makes one upload and one download" is then something a test can check.
extern std::atomic<int64_t> g_xfer[4];
static inline void count_xfer(hipMemcpyKind k, size_t bytes) {
    if (k == hipMemcpyHostToDevice) {
        g_xfer[0] += (int64_t)bytes;
        g_xfer[1] += 1;
    } else if (k == hipMemcpyDeviceToHost) {
        g_xfer[2] += (int64_t)bytes;
        g_xfer[3] += 1;
    }
}
static inline hipError_t sa_memcpy(void *d, const void *s, size_t n, hipMemcpyKind k) {
    count_xfer(k, n);
    return hipMemcpy(d, s, n, k);
}
static inline hipError_t sa_memcpy_async(void *d, const void *s, size_t n, hipMemcpyKind k,
                                         hipStream_t st) {
    count_xfer(k, n);
    return hipMemcpyAsync(d, s, n, k, st);
}
static inline hipError_t sa_memcpy2d_async(void *d, size_t dp, const void *s, size_t sp, size_t w,
                                           size_t h, hipMemcpyKind k, hipStream_t st) {
    count_xfer(k, w * h);
    return hipMemcpy2DAsync(d, dp, s, sp, w, h, k, st);
}
#define hipMemcpy sa_memcpy
#define hipMemcpyAsync sa_memcpy_async
#define hipMemcpy2DAsync sa_memcpy2d_async

enum ProfSlot {
    PS_FFT_R2C = 0,
    PS_FFT_C2C_FWD,
    PS_SM_SOLVE,
    PS_FFT_C2C_INV,
    PS_FFT_C2R,
    PS_ADMM_POST,
    PS_FUSED_COLS,
    PS_ROWS_FWD,
    PS_ROWS_INV_POST,
    PS_ROWS_INV_POST_EMIT,
    PS_ROWS_FWD_V,              // the same three in the single-array state (csc_rows.h):
    PS_ROWS_INV_POST_V,         // V in and out (the (Y, U) -> V transition counts with the
    PS_ROWS_INV_POST_V_EMIT,    // (Y, U) slots: it reads both arrays)
    PS_PGM_GRAD_IFFT,
    PS_PGM_ROWS_PROX,
    PS_PGM_FFT_MOM,
    PS_FINALIZE,
    PS_PGM,
    PS_OTHER,
    PS_PERSIST,                 // a run of iterations in one launch (csc_rows.h admm_persist)
    PS_SETCOEF_ROWS,            // dictionary update: row / column transform of the coefficient maps
    PS_SETCOEF_COLS,            // into the tile-major spectrum Zf (ccmod_setcoef)
    PS_CCMOD_GRAD,              // ... and the gradient over tiles (ccmod_grad_tiled + group sum)
    PS_C2R_VPOST,               // generic chain: c2r row pass + epilogue of the single-array state in one
    PS_C2R_VPOST_EMIT,          // kernel (fft.h fft_c2r_vpost), ... + the next iteration's row spectrum
    PS_COUNT
};
extern const char *kProfNames[PS_COUNT];

struct Profiler {
    bool on = false;
    hipStream_t st = nullptr;
    struct Rec {
        int slot;
        hipEvent_t a, b;
    };
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[PS_COUNT] = {0};
    int64_t count[PS_COUNT] = {0};

    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        SA_HIP(hipEventCreate(&e));
        return e;
    }
    void drain() {
        for (auto &r : pending) {
            SA_HIP(hipEventSynchronize(r.b));
            float ms = 0.f;
            SA_HIP(hipEventElapsedTime(&ms, r.a, r.b));
            total_ms[r.slot] += ms;
            count[r.slot] += 1;
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        pending.clear();
    }
    ~Profiler() {
        for (auto &r : pending) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        for (auto e : pool) (void)hipEventDestroy(e);
    }
};

// RAII timing scope around one kernel (group) on the handle's stream.
struct ProfScope {
    Profiler &p;
    int slot;
    hipEvent_t a = nullptr;
    ProfScope(Profiler &p_, int slot_) : p(p_), slot(slot_) {
        if (p.on) {
            a = p.get();
            SA_HIP(hipEventRecord(a, p.st));
        }
    }
    ~ProfScope() {
        if (p.on && a) {
            hipEvent_t b = p.get();
            (void)hipEventRecord(b, p.st);
            p.pending.push_back({slot, a, b});
            if (p.pending.size() >= 200) p.drain();
        }
    }
};

struct CscBase {
    virtual ~CscBase() {}
    virtual void sync() = 0;
    virtual void *stream_handle() = 0;
    virtual int query(int what) = 0;
    virtual std::string placement() = 0;
    virtual void set_hint(int what, int value) = 0;
    virtual void set_signal(const void *S) = 0;
    virtual void set_signal_dev(const void *S_dev) = 0;
    virtual void reconstruct_dev(int var, void *dst_dev) = 0;
    virtual void set_dict(const void *D, int dH, int dW) = 0;
    virtual void set_dict_imag(const void *D, int dH, int dW) = 0;
    virtual void set_weight(int which, const void *w, const int64_t shape[5]) = 0;
    virtual void set_grad_weight(const void *w) = 0;
    virtual void set_filter_sizes(const int32_t *fh, const int32_t *fw) = 0;
    virtual void upload(int var, const void *src) = 0;
    virtual void download(int var, void *dst) = 0;
    virtual void *device_ptr(int var) = 0;
    virtual void admm_iter(const sporco_amd_admm_params &p, double *out_dev) = 0;
    virtual int admm_run(const sporco_amd_admm_params &p, const sporco_amd_admm_ctrl &c,
                         sporco_amd_admm_record *records, double *rho_out, double *u_scale_out,
                         sporco_amd_reduce_fn reduce, void *user) = 0;
    virtual void admm_xstep(const sporco_amd_admm_params &p, double *out_dev) = 0;
    virtual void admm_relax(double rlx) = 0;
    virtual void admm_ystep(const sporco_amd_admm_params &p) = 0;
    virtual void admm_ustep(const sporco_amd_admm_params &p) = 0;
    virtual void admm_stats(const sporco_amd_admm_params &p, double *out_dev) = 0;
    virtual void scale_u(double s) = 0;
    virtual void reconstruct(int var, void *dst) = 0;
    virtual void dhs_absmax(double *out_host) = 0;
    virtual void pgm_grad(int var, double *out_dev) = 0;
    virtual void pgm_eval(int var, double *out_dev) = 0;
    virtual void pgm_iter(const sporco_amd_pgm_params &p, double *out_dev) = 0;
    virtual void pgm_commit() = 0;
    virtual void pgm_prox_step(double L, double lmbda, uint32_t flags, int dH, int dW,
                               double *out_dev) = 0;
    virtual void lincomb(int dst, double a, int va, double b, int vb, double c, int vc) = 0;
    virtual void pair_stats(int va, int vb, int vg, double *out_dev) = 0;
    virtual void pgm_resid(int var, int slot) = 0;
    virtual void pgm_resid_stats(int a, int b, int c, int d, double *out_dev) = 0;
    virtual void copy(int dst, int src) = 0;
    virtual void ccmod_setcoef(int var) = 0;
    virtual void ccmod_grad(int var, bool write_grad, double *out_dev) = 0;
    virtual void ccmod_prox_step(double L, int dH, int dW, bool zm) = 0;
    virtual void ccmod_sgd_step(double eta, int dH, int dW, bool zm, double *out_dev) = 0;
    virtual void ccmod_cnstr(int dH, int dW, bool zm, double *out_dev) = 0;
    virtual void ccmod_getdict(int dH, int dW, void *dst) = 0;
    virtual void setdict_from_dstep(int dH, int dW) = 0;
    virtual void asum(int var, double *out_dev) = 0;
    virtual void masked_grad(int var, bool dstep, int mode, double *out_dev) = 0;
    virtual void cns_init(const void *Y0, double rho) = 0;
    virtual void cns_iter(const sporco_amd_cns_params &p, double *out_dev) = 0;
    virtual void cns_md_init(const void *S) = 0;
    virtual void *cns_mean_ptr(int64_t *count) = 0;
    virtual void mdcpl_init(const void *S) = 0;
    virtual void mdcpl_iter(const sporco_amd_admm_params &p, double *out_dev) = 0;
    virtual void dstep_init(const void *Y0) = 0;
    virtual void dstep_md_init(const void *Y0, const void *S) = 0;
    virtual void dstep_iter(const sporco_amd_dstep_params &p, double *out_dev) = 0;
    virtual void fft_var(int rvar, int cvar, bool inverse) = 0;
    virtual void read_out(const double *out_dev, double *out_host) = 0;
    double *out_dev_default = nullptr;
    Profiler prof;
    void *comm_user = nullptr;   // sporco_amd_csc_set_comm: admm_run sums its 16 doubles over the ranks
};
// (csc_comm.hip) all-reduce of the 16 per-iteration sums on `st`
void comm_reduce_sums(void *user, double *sums_dev, hipStream_t st);

static bool var_is_complex(int var) {
    switch (var) {
    case SPORCO_AMD_VAR_XF:
    case SPORCO_AMD_VAR_DF:
    case SPORCO_AMD_VAR_SF:
    case SPORCO_AMD_VAR_YF:
    case SPORCO_AMD_VAR_XFPRV:
    case SPORCO_AMD_VAR_YFPRV:
    case SPORCO_AMD_VAR_VF:
    case SPORCO_AMD_VAR_GF:
    case SPORCO_AMD_VAR_T0:
    case SPORCO_AMD_VAR_T1:
    case SPORCO_AMD_VAR_T2:
    case SPORCO_AMD_VAR_ZF:
    case SPORCO_AMD_VAR_DXF:
    case SPORCO_AMD_VAR_DYF:
    case SPORCO_AMD_VAR_DXFPRV:
    case SPORCO_AMD_VAR_DYFPRV:
    case SPORCO_AMD_VAR_DVF:
    case SPORCO_AMD_VAR_DGF:
    case SPORCO_AMD_VAR_DT0:
    case SPORCO_AMD_VAR_DT1:
    case SPORCO_AMD_VAR_DT2:
        return true;
    default:
        return false;
    }
}

static bool var_is_dict_sized(int var) {
    return var == SPORCO_AMD_VAR_DF || (var >= SPORCO_AMD_VAR_DX && var < SPORCO_AMD_VAR_COUNT);
}

static bool var_is_valid(int var) {
    return (var >= 0 && var <= SPORCO_AMD_VAR_DMU0) ||
           (var >= SPORCO_AMD_VAR_DX && var < SPORCO_AMD_VAR_COUNT);
}

// the one place that instantiates Csc<float> / Csc<double> (csc_api.hip)
CscBase *make_csc(const sporco_amd_dims &dims, int dict_channels, int device, void *stream,
                  int depth = 1);

}  // namespace sporco_amd


struct sporco_amd_csc {
    std::unique_ptr<sporco_amd::CscBase> impl;
    int device;
    int depth = 1;                // > 1: a volume handle (sporco_amd_csc_create_volume)
    double *stats_dev = nullptr;  // scratch for pgm_stats into a separate buffer
    ~sporco_amd_csc() {
        if (stats_dev) (void)hipFree(stats_dev);
    }
};

#define SA_API_BEGIN try {
#define SA_API_END                                                                     \
    }                                                                                  \
    catch (const sporco_amd::Error &e) {                                               \
        g_last_error = e.what();                                                       \
        return e.code;                                                                 \
    }                                                                                  \
    catch (const std::bad_alloc &) {                                                   \
        g_last_error = "host allocation failed";                                       \
        return SPORCO_AMD_ENOMEM;                                                      \
    }                                                                                  \
    catch (const std::exception &e) {                                                  \
        g_last_error = e.what();                                                       \
        return SPORCO_AMD_EINVAL;                                                      \
    }                                                                                  \
    return SPORCO_AMD_OK;

#define SA_HANDLE_ANY(h)                                                               \
    SA_REQUIRE((h) != nullptr && (h)->impl, "null solver handle");                     \
    SA_HIP(hipSetDevice((h)->device));
// (every entry point that has not been taught the third transform axis refuses a volume handle)
#define SA_HANDLE(h)                                                                   \
    SA_HANDLE_ANY(h)                                                                   \
    SA_REQUIRE((h)->depth == 1,                                                        \
               "a volume handle (dimN = 3) serves the ADMM sparse coding calls only");
