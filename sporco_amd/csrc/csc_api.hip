// csc_api.hip -- C ABI of libsporco_amd.so (see include/sporco_amd.h).
//
// The handle owns every device array of one ConvBPDN problem; the host side
// (sporco_amd/*.py, which mirrors the reference classes) keeps only scalars and
// the iteration loop.  Arrays stay resident in HBM for the life of the handle.
#include "../../include/sporco_amd.h"

#include <cmath>
#include <cstdlib>
#include <memory>
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

static thread_local std::string g_last_error;

// Every host <-> device copy of this file is counted (sporco_amd_transfer_stats): the claim
// "the pipeline makes one upload and one download" is then something a test can check.
static std::atomic<int64_t> g_xfer[4];
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
    PS_COUNT
};
static const char *kProfNames[PS_COUNT] = {"fft_r2c_rows",     "fft_c2c_cols_fwd", "sm_solve",
                                           "fft_c2c_cols_inv", "fft_c2r_rows",     "admm_post",
                                           "fused_cols_sm",    "rows_fwd",         "rows_inv_post",
                                           "rows_inv_post_emit",
                                           "rows_fwd_v",       "rows_inv_post_v",  "rows_inv_post_v_emit",
                                           "pgm_grad_ifft",    "pgm_rows_prox",    "pgm_fft_momentum",
                                           "finalize",         "pgm_elementwise",  "other",
                                           "admm_persist_run"};

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
    virtual void set_hint(int what, int value) = 0;
    virtual void set_signal(const void *S) = 0;
    virtual void set_signal_dev(const void *S_dev) = 0;
    virtual void reconstruct_dev(int var, void *dst_dev) = 0;
    virtual void set_dict(const void *D, int dH, int dW) = 0;
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
};

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

template <typename T> struct Csc : CscBase {
    sporco_amd_dims dm;
    int device;
    hipStream_t st = nullptr;
    bool own_stream = false;
    int H, W, Wf, C, N, K, CN;
    // Ku: the caller's filter count.  K (what every kernel sees) is Ku, or Ku + 1 when Ku is
    // odd and one all-zero filter buys the register-resident kernels (which pair filters):
    // a zero filter's coefficient map stays zero through every iteration of every solver here
    // (x_k = yuf_k, prox(0) = 0, zero gradient) up to the rounding-level cross-talk of the
    // row FFTs, which transform two filters as one complex line; the shrinkage removes that
    // again, and the dictionary projection (the one place that would normalise noise up to
    // unit norm) keeps padding filters at zero.  Host arrays always have Ku filters; the
    // copies in and out are strided.
    int Ku;
    // Multi-channel dictionary (Cd > 1, cnvrep.py:186-194): D and S have Cd = Cs channels,
    // the coefficient maps one (C = 1); X-step by iterated Sherman-Morrison (ism_*).
    int Cd = 1, Cs, CNs;
    cx<T> *ism_gam = nullptr, *ism_del = nullptr, *ism_mm = nullptr;
    // ... and on the fast-path shapes the register-resident column kernel of csc_fused_mc.hip
    bool fused_mc = false;
    cx<T> *dft_mc = nullptr, *sft_mc = nullptr;
    T *bt_mc = nullptr;
    bool binv_valid = false;
    double binv_rho = 0.0;
    // consensus D-step scratch: spectrum of the per-image copies, per-(pixel, image) gram of
    // the coefficient spectra, mean / previous-Y buffers (dictionary sized)
    cx<T> *cns_f = nullptr;
    // 64 < K <= 64 + kTailMax filters: single column kernel on the first 64, the tail through
    // the generic column FFT (run_fused_cols)
    static constexpr int kTailMax = 8;
    // In that mode the rows of the tile-major spectrum (and of dft) are Ks = 80 filters apart,
    // so that every 8 K-byte row starts on a 128-byte line (K * 8 = 528 .. 576 bytes does not
    // divide into lines: measured 1.65 ms against 1.2 ms for rows_fwd at K = 66).  Ks = K
    // everywhere else.
    int Ks = 0;
    int64_t EFt = 0;   // elements of a buffer that may hold the tile-major spectrum
    bool tail_mode = false;   // decided once, at construction
    cx<T> *sft_eff = nullptr, *coef_t = nullptr;
    uint32_t *ams_bits = nullptr;   // AddMaskSim mask, one bit per pixel (csc_rows.h)
    bool ams_bits_valid = false;
    T *cns_m = nullptr, *cns_yold = nullptr;
    bool cns_active = false;   // a consensus D-step lives on this handle
    // single-copy ADMM D-step (dstep_iter): Zf stays in the natural layout; ZSf cache and the
    // iterated Sherman-Morrison tables over the images
    T *md_s = nullptr;         // ConvBPDNMaskDcpl: the real signal
    bool eq_active = false;
    bool zsf_valid = false, dism_valid = false;
    double dism_rho = 0.0;
    cx<T> *dism_gam = nullptr, *dism_del = nullptr, *dism_mm = nullptr;
    T *gramz_t = nullptr;      // its fused path: sum_k |Zf|^2 per row of the tile-major Zf
    bool gramz_valid = false;
    bool cns_fused() const {
        return rows_ok && fused && cols256 && !std::getenv("SPORCO_AMD_CNS_GENERIC");
    }
    bool ism_valid = false;
    double ism_rho = 0.0, ism_mu = -1.0;   // (ism_mu: mu of the gradient diagonal, -1 = none)
    int64_t P, E, npix, EF;  // P = C*N*K, E = H*W*P, npix = H*Wf, EF = npix*P
    FftPlan planW, planH;
    void *vars[SPORCO_AMD_VAR_COUNT] = {nullptr};
    cx<T> *work = nullptr;    // column-pass scratch (EF complex)
    cx<T> *dwork = nullptr;   // column-pass scratch of the D-step (npix*K complex)
    T *pcn_stats = nullptr;   // per-filter mean and 1/norm of the constraint projection
    cx<T> *innerb = nullptr;  // (npix, CN) complex
    T *gram = nullptr;        // (npix)
    T *dpad = nullptr;        // (H, W, K) real
    T *sreal = nullptr;       // (H, W, CN) real staging / reconstruct output
    T *wl1_buf = nullptr, *wl21_buf = nullptr, *wams_buf = nullptr, *wdat_buf = nullptr;
    Weight<T> wl1, wl21, wams;   // wams: AddMaskSim mask (F_AMS)
    Weight<T> wdat;              // data-fidelity mask of the *Mask PGM classes
    bool have_wdat = false;
    double *part_a = nullptr, *part_b = nullptr;  // block partials
    double *out_dev_own = nullptr;
    double *out_pinned = nullptr;
    bool have_dict = false, have_signal = false;
    int dH_ = 0, dW_ = 0;
    // fused X-step (csc_fused.h): tile-major copies of Df, Sf, gram, its twiddles and
    // per-tile partials; xf_tiled marks VAR_XF as holding a tile-major intermediate
    bool fused = false, xf_tiled = false;
    bool cols256 = false;           // H in {128, 256, 512}: every column kernel family serves it
    bool fused_slabs = false;       // K = 64*NH: column pass as two slab kernels (csc_fused.h)
    cx<T> *qpart = nullptr;
    unsigned *coop_flags = nullptr;     // cooperating slab workgroups (csc_fused.h): per (tile, slab)
    unsigned coop_seq = 0;              // ... the launch counter their flags carry
    int *coop_err = nullptr;            // ... pinned: set by a workgroup whose partner never showed up
    cx<T> *dft = nullptr, *sft = nullptr, *twA = nullptr, *twB = nullptr;
    T *gramt = nullptr;
    double *part_f = nullptr;
    int part_f_rows = 0;   // rows of part_f the last column pass wrote (tiles, or tiles x slabs)
    // fused row passes (csc_rows.h)
    bool rows_ok = false;
    cx<T> *twRows = nullptr;
    double *part_rows = nullptr;
    // one-launch solve of small problems (csc_rows.h admm_persist): second set of partial-sum
    // buffers, per-workgroup argument copies and control blocks, the barrier words
    double *pst_part_rows = nullptr, *pst_part_f = nullptr;
    void *pst_blk = nullptr;
    AdmmCtl *pst_ctl = nullptr;
    unsigned *pst_bar = nullptr;
    int pst_grid = 0, pst_runs = 0;
    // The three-launch iteration writes the new (Y, U) into a second pair of
    // buffers and swaps: the previous iterate stays intact, so X (which that path
    // keeps in registers only) can be rebuilt exactly, on demand, by re-running the
    // X-step on it with the parameters of the iteration (`last_p`).
    T *y_alt = nullptr, *u_alt = nullptr;
    bool x_stale = false, x_invalid = false;
    // after a three-launch iteration the previous iterate sits in y_alt and AX was never
    // formed: a host read of VAR_YPREV / VAR_AX derives them (download)
    bool prev_in_alt = false;
    // Single-array state of the fused iteration (csc_rows.h, "V form").  While v_live, the
    // current iterate is V = AX + U of the iteration that produced it, in v_cur (one of y_alt /
    // u_alt, which ping-pong as V buffers); vars[Y] / vars[U] then hold the iterate the run
    // started from (v_prev_kind 1: it is the previous iterate) or stale data (v_prev_kind 2:
    // the previous iterate is the V in the other alt buffer, produced with v_prev_thr).
    // ensure_yu() returns the handle to the (Y, U) form every other code path expects;
    // ensure_prev_yu() also brings the previous iterate back into (y_alt, u_alt).
    bool v_live = false;
    T *v_cur = nullptr;
    T v_thr = T(0), v_prev_thr = T(0);
    T v_thr21 = T(0), v_prev_thr21 = T(0), vp_thr21 = T(0);   // ConvBPDNJoint: the l2,1 thresholds
    bool v_nonneg = false, v_joint = false;
    uint32_t v_opts = 0;     // F_NOBNDRY | F_AMS of the iterations that produced the V's
    int v_dH = 1, v_dW = 1;  // ... and their filter support (NoBndryCross)
    int v_prev_kind = 0;
    bool vp_pending = false;       // the previous iterate still waits, in V form, in vp_buf
    T *vp_buf = nullptr, *vp_free = nullptr;
    T vp_thr = T(0);
    bool vp_nonneg = false;
    uint64_t touch_epoch = 0, fused_epoch = ~(uint64_t)0;   // host accesses between fused iterations
    // t_ready: the Xf buffer already holds rows_fwd(Y, U, s = 1) of the current
    // iterate, emitted by the previous rows_inv_post on the bet that rho stays put
    bool t_ready = false;
    int stable_run = 0;   // consecutive fused iterations entered with an unchanged rho
    // device-driven solve (admm_run): control block in device memory, records in pinned memory
    AdmmCtl *ctl_dev = nullptr;
    AdmmRecord *rec_ring = nullptr;
    int rec_cap = 0;
    CgCtl *cg_dev = nullptr;        // CG dictionary update: scalars on the device
    CgPinned *cg_pin = nullptr;
    sporco_amd_admm_params last_p;
    // fused PGM iteration (csc_pgm.h): Xf, Yf, Xfprv, Yfprv tile-major; X of the last
    // iteration is prox(irfft_W(work)) and is rebuilt on demand with `last_pgm`
    bool zf_tiled = false;          // VAR_ZF holds the tile-major spectrum (fused D-step)
    cx<T> *gpart = nullptr;         // group partials of the tiled D-step gradient
    int ccmod_groups = 1;
    bool pgm_tiled = false, pgm_x_stale = false;
    sporco_amd_pgm_params last_pgm;
    double *part_pgm = nullptr, *part_pgm2 = nullptr;
    cx<T> *ccmod_r = nullptr;       // K > 64 tiled dictionary-update gradient: the residual per frequency
    cx<T> *pgm_ey = nullptr;        // e_y of a held (backtracking) pgm_iter, tile-major (Wf, CN, H)
    bool pgm_held = false;          // a trial's iterates wait in the spare buffers
    bool run_always_emit = false;   // device-driven solve of a small problem (admm_run)
    // ConvBPDNGradReg (F_GRADREG): separable gradient spectrum tables and filter weights
    T *ghh = nullptr, *ghw = nullptr, *wg = nullptr;
    bool have_wg = false;
    T *g1t = nullptr;               // fused path: 1 + sum_k |Df|^2 / diagonal, tile-major
    bool g1_valid = false;
    double g1_rho = 0.0, g1_mu = 0.0;

    static int padded_filters(int H_, int W_, int K_) {
        if (K_ % 2 == 0 || std::getenv("SPORCO_AMD_UNFUSED") || std::getenv("SPORCO_AMD_NO_PAD"))
            return K_;
        const bool cols = fused_cols_supported<T>(H_, K_ + 1) || fused_slabs_supported<T>(H_, K_ + 1);
        return (cols && rows_supported<T>(W_, K_ + 1)) ? K_ + 1 : K_;
    }

    Csc(const sporco_amd_dims &d, int dev, void *stream, int cd) : dm(d), device(dev) {
        SA_REQUIRE(d.H >= 1 && d.W >= 1 && d.C >= 1 && d.N >= 1 && d.K >= 1,
                   "all dimensions must be >= 1");
        SA_REQUIRE(cd == 1 || cd == d.C,
                   "a multi-channel dictionary needs as many channels as the signal");
        SA_HIP(hipSetDevice(device));
        H = d.H;
        W = d.W;
        Cd = cd;
        Cs = d.C;
        C = cd > 1 ? 1 : d.C;   // channels of the coefficient maps
        N = d.N;
        Ku = d.K;
        K = cd > 1 ? d.K : padded_filters(d.H, d.W, d.K);
        Wf = W / 2 + 1;
        CN = C * N;
        CNs = Cs * N;
        P = (int64_t)C * N * K;
        E = (int64_t)H * W * P;
        npix = (int64_t)H * Wf;
        EF = npix * P;
        if (stream) {
            st = (hipStream_t)stream;
        } else {
            SA_HIP(hipStreamCreate(&st));
            own_stream = true;
        }
        prof.st = st;
        planW.init(W);
        planH.init(H);
        SA_HIP(hipMalloc((void **)&part_a, sizeof(double) * kMaxPartialBlocks * 8));
        SA_HIP(hipMalloc((void **)&part_b, sizeof(double) * kMaxPartialBlocks * 8));
        SA_HIP(hipMalloc((void **)&out_dev_own, sizeof(double) * kOutSlots));
        SA_HIP(hipMemset(out_dev_own, 0, sizeof(double) * kOutSlots));
        SA_HIP(hipHostMalloc((void **)&out_pinned, sizeof(double) * kOutSlots, 0));
        out_dev_default = out_dev_own;
        SA_HIP(hipMalloc((void **)&gram, sizeof(T) * npix));
        SA_HIP(hipMalloc((void **)&innerb, sizeof(cx<T>) * npix * CNs));
        SA_HIP(hipMalloc((void **)&sreal, sizeof(T) * (int64_t)H * W * CNs));
        fused = Cd == 1 && fused_cols_supported<T>(H, K) && K % 2 == 0 &&
                !std::getenv("SPORCO_AMD_UNFUSED");
        cols256 = H == 128 || H == 256 || H == 512;   // (every column kernel family has the 32 x 4 split)
        fused_slabs = Cd == 1 && fused_slabs_supported<T>(H, K) && !std::getenv("SPORCO_AMD_UNFUSED");
        fused_mc = Cd > 1 && fused_mc_supported<T>(H, K, Cd) && K % 2 == 0 &&
                   !std::getenv("SPORCO_AMD_UNFUSED");
        if (fused_mc) {
            SA_HIP(hipMalloc((void **)&dft_mc, sizeof(cx<T>) * npix * Cd * K));
            SA_HIP(hipMalloc((void **)&sft_mc, sizeof(cx<T>) * npix * CNs));
            SA_HIP(hipMalloc((void **)&bt_mc, sizeof(T) * npix * 2 * Cd * Cd));
            SA_HIP(hipMalloc((void **)&part_f, sizeof(double) * 2 * (int64_t)Wf * CN));
            SA_HIP(hipMalloc((void **)&twA, sizeof(cx<T>) * H));
            SA_HIP(hipMalloc((void **)&twB, sizeof(cx<T>) * H));
            std::vector<cx<T>> ta(H), tb(H);
            fused_twiddles<T>(H, K, ta.data(), tb.data());
            SA_HIP(hipMemcpy(twA, ta.data(), sizeof(cx<T>) * H, hipMemcpyHostToDevice));
            SA_HIP(hipMemcpy(twB, tb.data(), sizeof(cx<T>) * H, hipMemcpyHostToDevice));
        }
        if (fused_slabs)
            SA_HIP(hipMalloc((void **)&qpart, sizeof(cx<T>) * (int64_t)Wf * CN * ((K + 63) / 64) * H));
        if (fused || fused_slabs) {
            SA_HIP(hipMalloc((void **)&dft, sizeof(cx<T>) * npix * K));
            SA_HIP(hipMalloc((void **)&sft, sizeof(cx<T>) * npix * CN));
            SA_HIP(hipMalloc((void **)&gramt, sizeof(T) * npix));
            // (two per tile, and per 64-filter slab for the gradient-regularised slab pass)
            SA_HIP(hipMalloc((void **)&part_f,
                             sizeof(double) * 2 * (int64_t)Wf * CN * ((K + 63) / 64)));
            SA_HIP(hipMalloc((void **)&twA, sizeof(cx<T>) * H));
            SA_HIP(hipMalloc((void **)&twB, sizeof(cx<T>) * H));
            std::vector<cx<T>> ta(H), tb(H);
            fused_twiddles<T>(H, K, ta.data(), tb.data());
            SA_HIP(hipMemcpy(twA, ta.data(), sizeof(cx<T>) * H, hipMemcpyHostToDevice));
            SA_HIP(hipMemcpy(twB, tb.data(), sizeof(cx<T>) * H, hipMemcpyHostToDevice));
        }
        rows_ok = (fused || fused_slabs || fused_mc) && rows_supported<T>(W, K) &&
                  !std::getenv("SPORCO_AMD_OLD_ROWS");
        tail_mode = fused_slabs && K - 64 <= kTailMax && !std::getenv("SPORCO_AMD_NO_TAIL");
        Ks = (rows_ok && tail_mode && !std::getenv("SPORCO_AMD_NO_ROW_PAD")) ? 80 : K;
        EFt = npix * CN * (int64_t)Ks;
        if (Ks != K) {   // dft was sized for K-filter rows above
            SA_HIP(hipFree(dft));
            SA_HIP(hipMalloc((void **)&dft, sizeof(cx<T>) * npix * Ks));
        }
        if (rows_ok) {
            SA_HIP(hipMalloc((void **)&twRows, sizeof(cx<T>) * W));
            std::vector<cx<T>> ta(W);
            rows_twiddles<T>(W, ta.data());
            SA_HIP(hipMemcpy(twRows, ta.data(), sizeof(cx<T>) * W, hipMemcpyHostToDevice));
            // (the joint epilogue tiles by (image, 32 filters): N K / 32 workgroups per row)
            SA_HIP(hipMalloc((void **)&part_rows,
                             sizeof(double) * 8 * (int64_t)H *
                                 std::max<int64_t>(ceil_div(P, 128), (int64_t)N * ceil_div(K, 32))));
        }
        // the three ADMM state arrays start at zero (yinit/uinit, admm.py:279-289)
        for (int v : {SPORCO_AMD_VAR_Y, SPORCO_AMD_VAR_U, SPORCO_AMD_VAR_X}) (void)var_ptr(v);
    }

    ~Csc() override {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(st);
        for (auto &v : vars)
            if (v) (void)hipFree(v);
        for (void *p : {(void *)pst_part_rows, (void *)pst_part_f, pst_blk, (void *)pst_ctl, (void *)pst_bar,
                        (void *)part_c2r, (void *)cns_w, (void *)cns_sft, (void *)flt_h, (void *)flt_w,
                        (void *)zf_ch})
            if (p) (void)hipFree(p);
        for (void *p : {(void *)dft, (void *)sft, (void *)gramt, (void *)part_f, (void *)twA, (void *)twB,
                        (void *)twRows, (void *)part_rows, (void *)y_alt, (void *)u_alt, (void *)part_pgm, (void *)part_pgm2, (void *)ccmod_r, (void *)pgm_ey, (void *)gpart,
                        (void *)qpart, (void *)coop_flags, (void *)ghh, (void *)ghw, (void *)wg, (void *)g1t, (void *)ism_gam, (void *)ism_del, (void *)ism_mm, (void *)dft_mc, (void *)sft_mc, (void *)bt_mc, (void *)cns_f, (void *)cns_m, (void *)sft_eff, (void *)coef_t, (void *)ams_bits, (void *)gramz_t,
                        (void *)cns_yold, (void *)md_s, (void *)dism_gam, (void *)dism_del, (void *)dism_mm,
                        (void *)dwork, (void *)pcn_stats, (void *)work, (void *)innerb, (void *)gram, (void *)dpad, (void *)sreal,
                        (void *)wl1_buf, (void *)wl21_buf, (void *)wams_buf, (void *)wdat_buf, (void *)part_a, (void *)part_b,
                        (void *)out_dev_own})
            if (p) (void)hipFree(p);
        if (out_pinned) (void)hipHostFree(out_pinned);
        if (coop_err) (void)hipHostFree(coop_err);
        if (rec_ring) (void)hipHostFree(rec_ring);
        if (cg_pin) (void)hipHostFree((void *)cg_pin);
        if (cg_dev) (void)hipFree(cg_dev);
        if (ctl_dev) (void)hipFree(ctl_dev);
        planW.destroy();
        planH.destroy();
        if (own_stream) (void)hipStreamDestroy(st);
    }

    int64_t KD() const { return (int64_t)Cd * K; }   // dictionary entries per pixel
    static bool var_is_signal_real(int var) {
        return var >= SPORCO_AMD_VAR_MY0 && var <= SPORCO_AMD_VAR_DMU0;
    }
    size_t var_bytes(int var) const {
        if (var == SPORCO_AMD_VAR_SF) return sizeof(cx<T>) * npix * CNs;
        if (var_is_signal_real(var)) return sizeof(T) * (int64_t)H * W * CNs;
        if (var_is_dict_sized(var))
            return var_is_complex(var) ? sizeof(cx<T>) * npix * KD()
                                       : sizeof(T) * (int64_t)H * W * KD();
        // (consensus copies of a multi-channel dictionary: one (Cd, K) block per image)
        if (var == SPORCO_AMD_VAR_CX || var == SPORCO_AMD_VAR_CU) return sizeof(T) * E * Cd;
        return var_is_complex(var) ? sizeof(cx<T>) * EF : sizeof(T) * E;
    }

    void *var_ptr(int var) {
        SA_REQUIRE(var_is_valid(var), "unknown state variable id");
        if (var == SPORCO_AMD_VAR_Y || var == SPORCO_AMD_VAR_U) {
            ++touch_epoch;
            if (v_live) ensure_yu();
        }
        if (!vars[var]) {
            // (the Xf buffer also holds the tile-major spectrum, whose rows may be padded)
            const size_t nb = var == SPORCO_AMD_VAR_XF ? sizeof(cx<T>) * (size_t)std::max(EF, EFt)
                                                       : var_bytes(var);
            SA_HIP(hipMalloc(&vars[var], nb));
            SA_HIP(hipMemsetAsync(vars[var], 0, nb, st));
        }
        return vars[var];
    }
    T *rv(int var) { return static_cast<T *>(var_ptr(var)); }
    cx<T> *cv(int var) { return static_cast<cx<T> *>(var_ptr(var)); }
    cx<T> *work_buf() {
        if (!work) SA_HIP(hipMalloc((void **)&work, sizeof(cx<T>) * std::max(EF, EFt)));
        return work;
    }
    cx<T> *dwork_buf() {
        if (!dwork) SA_HIP(hipMalloc((void **)&dwork, sizeof(cx<T>) * npix * KD()));
        return dwork;
    }
    T *pcn_stats_buf() {
        if (!pcn_stats) SA_HIP(hipMalloc((void **)&pcn_stats, sizeof(T) * 2 * KD()));
        return pcn_stats;
    }
    Dims5 d5() const { return Dims5{H, W, C, N, K}; }

    void *stream_handle() override { return (void *)st; }
    void sync() override {
        SA_HIP(hipStreamSynchronize(st));
        // (cooperating slab workgroups, csc_fused.h: a partner's partial sums never arrived)
        if (coop_err && *coop_err) {
            *coop_err = 0;
            throw Error(SPORCO_AMD_EHIP, "cooperating slab workgroups: a partner's partial sums never "
                                         "arrived; the iterates of this handle are invalid");
        }
    }
    // The fused FISTA iteration: the register-resident kernels of both directions, and -- for
    // K > 64 -- rows of exactly K filters (a handle in tail mode, 64 < K <= 72, pads the rows of
    // its Xf buffer to 80 for the ADMM tail kernels: the staged composition serves it).
    bool pgm_fused_ok() const { return rows_ok && cols256 && (fused || (fused_slabs && !tail_mode)); }
    bool hint_vform = false, hint_one_launch = false;
    void set_hint(int what, int value) override {
        if (what == SPORCO_AMD_HINT_KEEP_VFORM) hint_vform = value != 0;
        else if (what == SPORCO_AMD_HINT_ONE_LAUNCH) hint_one_launch = value != 0;
        else throw Error(SPORCO_AMD_EINVAL, "unknown hint");
    }
    int query(int what) override {
        if (what == SPORCO_AMD_QUERY_FUSED_COLS) return (fused || fused_slabs || fused_mc) ? 1 : 0;
        if (what == SPORCO_AMD_QUERY_FUSED_ROWS) return rows_ok ? 1 : 0;
        if (what == SPORCO_AMD_QUERY_FUSED_PGM) return pgm_fused_ok() ? 1 : 0;
        if (what == SPORCO_AMD_QUERY_DEVICE_FILTERS) return K;
        if (what == SPORCO_AMD_QUERY_VFORM_LIVE) return v_live ? 1 : 0;
        if (what == SPORCO_AMD_QUERY_PERSIST_RUNS) return pst_runs;
        throw Error(SPORCO_AMD_EINVAL, "unknown query");
    }

    // ---- 2-D transforms with per-kernel timing -------------------------------
    void fwd2(const T *in, const T *in2, T s2, cx<T> *out, int64_t cols) {
        {
            ProfScope ps(prof, PS_FFT_R2C);
            fft_r2c<T>(st, planW, in, in2, s2, out, H, cols, (int64_t)W * cols, cols,
                       (int64_t)Wf * cols, cols);
        }
        {
            ProfScope ps(prof, PS_FFT_C2C_FWD);
            fft_c2c<T>(st, planH, false, out, out, 1, (int64_t)Wf * cols, 0, (int64_t)Wf * cols, 0,
                       (int64_t)Wf * cols, T(1));
        }
    }
    // The generic ADMM iteration can fuse its epilogue into the half-spectrum -> real row pass
    // (fft.h fft_c2r_post): the x step then stops after the column pass (c2r_deferred), and
    // admm_iter runs the rest.  Opt-in (SPORCO_AMD_C2R_POST=1): measured, it saves the write and
    // re-read of X but moves Y, U, X in the row pass's 128-byte segments instead of the epilogue
    // kernel's long runs, and the two cancel (profiles/r03q_generic_chain.md: 392 against 387
    // it/s at 512 x 512, K = 64, N = 8; +3 % in float64 at 256 x 256).
    bool defer_c2r = false, c2r_deferred = false;
    double *part_c2r = nullptr;
    int64_t part_c2r_cap = 0;
    void inv2(const cx<T> *in, cx<T> *tmp, T *out, int64_t cols) {
        {
            ProfScope ps(prof, PS_FFT_C2C_INV);
            fft_c2c<T>(st, planH, true, in, tmp, 1, (int64_t)Wf * cols, 0, (int64_t)Wf * cols, 0,
                       (int64_t)Wf * cols, T(1));
        }
        if (defer_c2r && cols == P && out == rv(SPORCO_AMD_VAR_X)) {
            c2r_deferred = true;
            return;
        }
        {
            ProfScope ps(prof, PS_FFT_C2R);
            fft_c2r<T>(st, planW, tmp, out, H, cols, (int64_t)Wf * cols, cols, (int64_t)W * cols,
                       cols, T(1.0 / ((double)H * (double)W)));
        }
    }

    // The column pass for 64 < K <= 256: one launch of cooperating slab workgroups, or the two
    // slab kernels (SPORCO_AMD_SLAB_COOP=0).  Returns the number of tiles.
    int64_t run_slab_cols(FusedSlabArgs<T> &sa) {
        static const bool coop = !(std::getenv("SPORCO_AMD_SLAB_COOP") &&
                                   std::atoi(std::getenv("SPORCO_AMD_SLAB_COOP")) == 0);
        if (!coop) {
            launch_cols_fwd_partial<T>(st, sa);
            return launch_cols_sm_apply_inv<T>(st, sa);
        }
        coop_prepare(sa);
        return launch_cols_slab_coop<T>(st, sa);
    }
    // flags, launch counter and error word of a launch of cooperating slab workgroups
    void coop_prepare(FusedSlabArgs<T> &sa) {
        if (!coop_flags) {
            const size_t n = sizeof(unsigned) * (size_t)Wf * CN * ((K + 63) / 64);
            SA_HIP(hipMalloc((void **)&coop_flags, n));
            SA_HIP(hipMemsetAsync(coop_flags, 0, n, st));
            SA_HIP(hipHostMalloc((void **)&coop_err, sizeof(int), 0));
            *coop_err = 0;
        }
        sa.coop_flags = coop_flags;
        sa.coop_seq = ++coop_seq;
        sa.coop_err = coop_err;
    }

    void finalize(const double *part, int nblocks, int stride, int nvals, const int *slots,
                  const double *scales, double *out_dev, bool is_max = false) {
        ProfScope ps(prof, PS_FINALIZE);
        launch_finalize(st, part, nblocks, stride, nvals, slots, scales, is_max, out_dev);
    }

    // ---- tile-major operands of the fused X-step -----------------------------------
    void refresh_fused_dict() {
        if (fused_mc) {
            ProfScope ps(prof, PS_OTHER);
            launch_permute_ab<cx<T>>(st, cv(SPORCO_AMD_VAR_DF), dft_mc, H, Wf, (int64_t)Cd * K);
            binv_valid = false;
            return;
        }
        if (!fused && !fused_slabs) return;
        ProfScope ps(prof, PS_OTHER);
        launch_permute_ab<cx<T>>(st, cv(SPORCO_AMD_VAR_DF), dft, H, Wf, K, 0, Ks);
        launch_permute_ab<T>(st, gram, gramt, H, Wf, 1);
        g1_valid = false;
    }
    void refresh_fused_signal() {
        if (fused_mc) {
            ProfScope ps(prof, PS_OTHER);
            launch_permute_ab<cx<T>>(st, cv(SPORCO_AMD_VAR_SF), sft_mc, H, Wf, (int64_t)CNs);
            return;
        }
        if (!fused && !fused_slabs) return;
        ProfScope ps(prof, PS_OTHER);
        launch_permute_ab<cx<T>>(st, cv(SPORCO_AMD_VAR_SF), sft, H, (int64_t)Wf * CN, 1);
    }
    // ---- single-array state (csc_rows.h): back to the (Y, U) form ----------------------------
    bool vform_ok(const sporco_amd_admm_params &p) const {
        const bool off = std::getenv("SPORCO_AMD_NO_VFORM") != nullptr;   // (test switch)
        return !off && std::is_same<T, float>::value && rows_ok &&
               !(p.flags & (F_KEEP_X | F_FEVAL_Y | F_XRRS)) &&
               (!(p.flags & F_JOINT) || joint_rows_ok(p));
    }
    // the live V was produced under the options of p (otherwise: back to (Y, U) first)
    bool vform_same_opts(const sporco_amd_admm_params &p) const {
        const uint32_t o = p.flags & (F_NOBNDRY | F_AMS);
        return (bool)(p.flags & F_NONNEG) == v_nonneg && (bool)(p.flags & F_JOINT) == v_joint &&
               o == v_opts && (!(o & F_NOBNDRY) || (p.dH == v_dH && p.dW == v_dW));
    }
    // Y (and / or U) of an iterate held as V: y or u may be null, u may alias v
    void vform_split(const T *v, T *y, T *u, T thr, T thr21) {
        if (v_joint)
            launch_vform_split_joint<T>(st, v, y, u, thr, thr21, v_nonneg, C, (int64_t)N * K,
                                        (int64_t)H * W);
        else if (wl1.ptr || v_opts)
            launch_vform_split_general<T>(st, v, y, u, thr,
                                          (v_nonneg ? F_NONNEG : 0u) | (v_opts & F_NOBNDRY), d5(),
                                          v_dH, v_dW, wl1, (v_opts & F_AMS) ? wams : Weight<T>(),
                                          Ku - 1);
        else
            launch_vform_split<T>(st, v, y, u, thr, v_nonneg, E);
    }
    void ensure_yu() {
        if (!v_live) return;
        v_live = false;
        T *other = v_cur == y_alt ? u_alt : y_alt;
        ProfScope ps(prof, PS_OTHER);
        if (v_prev_kind == 1) {
            // vars hold the previous iterate as (Y, U) and the other alt buffer is free: the
            // new pair goes to (other, v_cur) and the buffers trade places -- exactly the state
            // an iteration of the (Y, U) form leaves behind
            vform_split(v_cur, other, v_cur, v_thr, v_thr21);
            T *oldY = static_cast<T *>(vars[SPORCO_AMD_VAR_Y]), *oldU = static_cast<T *>(vars[SPORCO_AMD_VAR_U]);
            vars[SPORCO_AMD_VAR_Y] = other;
            vars[SPORCO_AMD_VAR_U] = v_cur;
            y_alt = oldY;
            u_alt = oldU;
            prev_in_alt = true;
        } else {
            vform_split(v_cur, static_cast<T *>(vars[SPORCO_AMD_VAR_Y]),
                        static_cast<T *>(vars[SPORCO_AMD_VAR_U]), v_thr, v_thr21);
            // the previous iterate stays in V form until somebody asks for it
            vp_pending = v_prev_kind == 2;
            vp_buf = other;
            vp_free = v_cur;
            vp_thr = v_prev_thr;
            vp_thr21 = v_prev_thr21;
            vp_nonneg = v_nonneg;
            prev_in_alt = false;
        }
        v_cur = nullptr;
    }
    void ensure_prev_yu() {
        ensure_yu();
        if (!vp_pending) return;
        vp_pending = false;
        ProfScope ps(prof, PS_OTHER);
        {
            const bool nn = v_nonneg;
            v_nonneg = vp_nonneg;
            vform_split(vp_buf, vp_free, vp_buf, vp_thr, vp_thr21);
            v_nonneg = nn;
        }
        y_alt = vp_free;
        u_alt = vp_buf;
        prev_in_alt = true;
    }

    // X of the last three-launch iteration, rebuilt from the previous iterate.
    void materialize_x() {
        if (x_invalid)
            throw Error(SPORCO_AMD_ESTATE,
                        "X / Xf of an iteration run with SPORCO_AMD_FLAG_NO_X were requested");
        if (pgm_x_stale) {
            pgm_x_stale = false;
            pgm_rows_prox(last_pgm, work_buf(), nullptr, rv(SPORCO_AMD_VAR_X), nullptr);
        }
        if (!x_stale) return;
        ensure_prev_yu();
        x_stale = false;
        t_ready = false;   // the Xf buffer is about to be reused
        sporco_amd_admm_params q = last_p;
        q.flags = last_p.flags & F_GRADREG;   // (the system solved, not the sums wanted)
        launch_rows_fwd_on(y_alt, u_alt, (T)q.u_scale);
        run_fused_cols(q, nullptr);
        rows_inverse_to(rv(SPORCO_AMD_VAR_X));
    }
    // call before reading `var` / before changing anything X depends on
    void before_read(int var) {
        ++touch_epoch;
        if (var == SPORCO_AMD_VAR_X || var == SPORCO_AMD_VAR_XF) materialize_x();
        need_natural(var);
    }
    void before_state_change() {
        ++touch_epoch;
        if ((x_stale && !x_invalid) || pgm_x_stale) materialize_x();
        ensure_yu();
        vp_pending = false;
        t_ready = false;
        prev_in_alt = false;
    }
    // The dictionary changes: a pending X depends on the old one, but the speculatively emitted
    // row spectra of Y - U (t_ready) and the ping-pong parity do not -- a dictionary-learning
    // loop keeps skipping the forward row pass of its one-iteration X-steps.
    void before_dict_change() {
        ++touch_epoch;
        if ((x_stale && !x_invalid) || pgm_x_stale) materialize_x();
    }
    void x_written() {
        x_stale = false;
        x_invalid = false;
        pgm_x_stale = false;
    }
    static bool is_pgm_iterate(int var) {
        return var == SPORCO_AMD_VAR_XF || var == SPORCO_AMD_VAR_YF ||
               var == SPORCO_AMD_VAR_XFPRV || var == SPORCO_AMD_VAR_YFPRV;
    }
    // natural (H, Wf*CN, K) <-> tile-major (Wf*CN, H, K) of one X-sized spectrum, through
    // the column-pass scratch buffer (pointer swap, no second copy)
    void relayout(int var, bool to_tiled) {
        cx<T> *src = cv(var), *dst = work_buf();
        const int64_t ks = var == SPORCO_AMD_VAR_XF ? Ks : K;   // row stride of the tiled side
        {
            ProfScope ps(prof, PS_OTHER);
            if (to_tiled)
                launch_permute_ab<cx<T>>(st, src, dst, H, (int64_t)Wf * CN, K, K, ks);
            else
                launch_permute_ab<cx<T>>(st, src, dst, (int64_t)Wf * CN, H, K, ks, K);
        }
        vars[var] = dst;
        work = src;
    }
    void pgm_leave_tiled() {
        if (!pgm_tiled) return;
        if (pgm_x_stale) materialize_x();   // needs `work` before it is reused as scratch
        pgm_tiled = false;
        for (int v : {SPORCO_AMD_VAR_XF, SPORCO_AMD_VAR_YF, SPORCO_AMD_VAR_XFPRV,
                      SPORCO_AMD_VAR_YFPRV})
            relayout(v, false);
    }

    // VAR_XF as callers know it (natural layout): after a fused X-step the buffer
    // holds a tile-major intermediate, and Xf = rfftn(X) is rebuilt on demand.
    void need_natural(int var) {
        if (var == SPORCO_AMD_VAR_XF) t_ready = false;
        if (pgm_tiled && is_pgm_iterate(var)) pgm_leave_tiled();
        if (var == SPORCO_AMD_VAR_ZF && zf_tiled) {
            if (pgm_x_stale) materialize_x();   // `work` is about to be used as scratch
            zf_tiled = false;
            relayout(SPORCO_AMD_VAR_ZF, false);
        }
        if (var == SPORCO_AMD_VAR_XF && xf_tiled) {
            xf_tiled = false;
            fwd2(rv(SPORCO_AMD_VAR_X), nullptr, T(0), cv(SPORCO_AMD_VAR_XF), P);
        }
    }

    // ---- set-up ------------------------------------------------------------------
    void set_signal(const void *S) override {
        before_state_change();
        SA_HIP(hipMemcpyAsync(sreal, S, sizeof(T) * (int64_t)H * W * CNs, hipMemcpyHostToDevice, st));
        fwd2(sreal, nullptr, T(0), cv(SPORCO_AMD_VAR_SF), CNs);
        refresh_fused_signal();
        sync();  // the host buffer may be released after return
        have_signal = true;
    }

    void set_signal_dev(const void *S_dev) override {
        before_state_change();
        SA_HIP(hipMemcpyAsync(sreal, S_dev, sizeof(T) * (int64_t)H * W * CNs,
                              hipMemcpyDeviceToDevice, st));
        fwd2(sreal, nullptr, T(0), cv(SPORCO_AMD_VAR_SF), CNs);
        refresh_fused_signal();
        sync();
        have_signal = true;
    }

    void set_dict(const void *D, int dH, int dW) override {
        SA_REQUIRE(dH >= 1 && dW >= 1 && dH <= H && dW <= W,
                   "filter support must fit inside the signal");
        before_dict_change();
        if (!dpad) SA_HIP(hipMalloc((void **)&dpad, sizeof(T) * (int64_t)H * W * Cd * K));
        // stage the compact filters at the tail of dpad's own allocation? no: use `work`-free
        // dedicated staging so set_dict is safe while iterates are live.
        T *stage = nullptr;
        SA_HIP(hipMalloc((void **)&stage, sizeof(T) * (int64_t)dH * dW * Cd * Ku));
        SA_HIP(hipMemcpyAsync(stage, D, sizeof(T) * (int64_t)dH * dW * Cd * Ku, hipMemcpyHostToDevice,
                              st));
        {   // (Cd > 1: host layout (dH, dW, Cd, K), never padded)
            ProfScope ps(prof, PS_OTHER);
            launch_pad_dict<T>(st, stage, dpad, H, W, Cd * K, dH, dW, Cd * Ku);
        }
        fwd2(dpad, nullptr, T(0), cv(SPORCO_AMD_VAR_DF), (int64_t)Cd * K);
        ism_valid = false;
        if (Cd == 1) {
            ProfScope ps(prof, PS_OTHER);
            launch_gram<T>(st, cv(SPORCO_AMD_VAR_DF), gram, npix, K);
        }
        refresh_fused_dict();
        sync();
        SA_HIP(hipFree(stage));
        dH_ = dH;
        dW_ = dW;
        have_dict = true;
    }

    void set_weight(int which, const void *w, const int64_t shape[5]) override {
        before_state_change();   // a pending X of the fused PGM step depends on the weights
        if (which == 2) ams_bits_valid = false;
        if (which == 3) have_wdat = w != nullptr;
        Weight<T> &dst = which == 0 ? wl1 : (which == 1 ? wl21 : (which == 2 ? wams : wdat));
        T *&buf = which == 0 ? wl1_buf : (which == 1 ? wl21_buf : (which == 2 ? wams_buf : wdat_buf));
        if (buf) {
            sync();
            SA_HIP(hipFree(buf));
            buf = nullptr;
        }
        dst = Weight<T>();
        if (!w) return;
        // (the data-fidelity mask lives on the signal, which keeps its channels under a
        // multi-channel dictionary)
        const int64_t full[5] = {H, W, which == 3 ? Cs : C, N, Ku};
        int64_t n = 1;
        // (the AddMaskSim mask of a multi-channel dictionary has one slice per impulse filter
        // on its last axis: cbpdn.py:2358-2364 swaps the mask's channel axis there)
        const bool ams_mc = which == 2 && Cd > 1 && shape[4] == Cd;
        for (int i = 0; i < 5; ++i) {
            SA_REQUIRE(shape[i] == 1 || shape[i] == full[i] || (i == 4 && ams_mc),
                       "weight shape must be 1 or the full extent on every axis");
            n *= shape[i];
        }
        if (which == 1) SA_REQUIRE(shape[2] == 1, "L21Weight must not vary over the channel axis");
        if (which >= 2)
            SA_REQUIRE(shape[4] == 1 || ams_mc, "the mask must not vary over the filter axis");
        int64_t dshape[5] = {shape[0], shape[1], shape[2], shape[3], shape[4]};
        std::vector<T> padded;
        const void *srcp = w;
        if (K != Ku && shape[4] == Ku && !ams_mc) {
            // weight 1 on the padding filter (its coefficients are zero whatever the weight)
            dshape[4] = K;
            const int64_t rows = n / Ku;
            padded.assign((size_t)(rows * K), T(1));
            const T *wt = static_cast<const T *>(w);
            for (int64_t r = 0; r < rows; ++r)
                for (int k = 0; k < Ku; ++k) padded[(size_t)(r * K + k)] = wt[r * Ku + k];
            srcp = padded.data();
            n = rows * K;
        }
        SA_HIP(hipMalloc((void **)&buf, sizeof(T) * n));
        SA_HIP(hipMemcpyAsync(buf, srcp, sizeof(T) * n, hipMemcpyHostToDevice, st));
        sync();
        int64_t stride = 1;
        for (int i = 4; i >= 0; --i) {
            dst.stride[i] = dshape[i] == 1 ? 0 : stride;
            stride *= dshape[i];
        }
        dst.ptr = buf;
    }

    // GradWeight (cbpdn.py:1063-1071, :1134-1139): K per-filter weights, NULL => scalar 1
    void set_grad_weight(const void *w) override {
        before_state_change();
        have_wg = w != nullptr;
        g1_valid = false;
        ism_valid = false;
        if (!w) return;
        if (!wg) SA_HIP(hipMalloc((void **)&wg, sizeof(T) * K));
        std::vector<T> tmp((size_t)K, T(1));
        std::memcpy(tmp.data(), w, sizeof(T) * Ku);
        SA_HIP(hipMemcpyAsync(wg, tmp.data(), sizeof(T) * K, hipMemcpyHostToDevice, st));
        SA_HIP(hipStreamSynchronize(st));
    }

    // sum_i |G_i|^2 of the difference filters [1, -1] along each axis is separable:
    // |1 - e^{-i t}|^2 = 2 - 2 cos t  (signal.gradient_filters, signal.py:230-239)
    GradTerm<T> grad_term(double mu) {
        if (!ghh) {
            std::vector<T> th(H), tw(Wf);
            const double tau = 6.283185307179586476925286766559;
            for (int h = 0; h < H; ++h) th[h] = (T)(2.0 - 2.0 * std::cos(tau * h / H));
            for (int f = 0; f < Wf; ++f) tw[f] = (T)(2.0 - 2.0 * std::cos(tau * f / W));
            // a singleton axis crops the 2-tap filter to its first tap (numpy's fft `s` rule)
            if (H == 1) th[0] = T(1);
            if (W == 1) tw[0] = T(1);
            SA_HIP(hipMalloc((void **)&ghh, sizeof(T) * H));
            SA_HIP(hipMalloc((void **)&ghw, sizeof(T) * Wf));
            SA_HIP(hipMemcpy(ghh, th.data(), sizeof(T) * H, hipMemcpyHostToDevice));
            SA_HIP(hipMemcpy(ghw, tw.data(), sizeof(T) * Wf, hipMemcpyHostToDevice));
        }
        GradTerm<T> g;
        g.ghh = ghh;
        g.ghw = ghw;
        g.wg = have_wg ? wg : nullptr;
        g.mu = (T)mu;
        return g;
    }

    // Host <-> device copy of one state array; the host side has Ku filters on its last axis.
    void host_copy(int var, void *host, bool to_device) {
        void *dev = var_ptr(var);
        const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
        if (K == Ku || var == SPORCO_AMD_VAR_SF || var_is_signal_real(var)) {
            if (to_device) SA_HIP(hipMemcpyAsync(dev, host, var_bytes(var), kind, st));
            else SA_HIP(hipMemcpyAsync(host, dev, var_bytes(var), kind, st));
            return;
        }
        const size_t es = var_is_complex(var) ? sizeof(cx<T>) : sizeof(T);
        const size_t rows = var_bytes(var) / (es * K);
        if (to_device) {
            SA_HIP(hipMemsetAsync(dev, 0, var_bytes(var), st));
            SA_HIP(hipMemcpy2DAsync(dev, es * K, host, es * Ku, es * Ku, rows, kind, st));
        } else {
            SA_HIP(hipMemcpy2DAsync(host, es * Ku, dev, es * K, es * Ku, rows, kind, st));
        }
    }

    void upload(int var, const void *src) override {
        if (is_pgm_iterate(var)) pgm_leave_tiled();
        if (var == SPORCO_AMD_VAR_ZF) {
            zf_tiled = false;
            gramz_valid = false;
            zsf_valid = dism_valid = false;
        }
        if (var == SPORCO_AMD_VAR_X) {
            x_written();
        } else if (var == SPORCO_AMD_VAR_XF) {
            if (x_stale && !x_invalid) materialize_x();   // keep X; Xf is replaced below
        } else if (var == SPORCO_AMD_VAR_Y || var == SPORCO_AMD_VAR_U || var == SPORCO_AMD_VAR_DF ||
                   var == SPORCO_AMD_VAR_SF) {
            before_state_change();
        }
        host_copy(var, const_cast<void *>(src), true);
        if (var == SPORCO_AMD_VAR_XF) xf_tiled = false;
        if (var == SPORCO_AMD_VAR_DF) {
            ism_valid = false;
            if (Cd == 1) launch_gram<T>(st, cv(SPORCO_AMD_VAR_DF), gram, npix, K);
            refresh_fused_dict();
        }
        if (var == SPORCO_AMD_VAR_SF) refresh_fused_signal();
        sync();
    }
    void download(int var, void *dst) override {
        if (var == SPORCO_AMD_VAR_YPREV || var == SPORCO_AMD_VAR_AX) ensure_prev_yu();
        if (prev_in_alt && y_alt && (var == SPORCO_AMD_VAR_YPREV || var == SPORCO_AMD_VAR_AX)) {
            // Yprev = the other half of the (Y, U) ping-pong; AX = rlx X + (1 - rlx) Yprev
            // (admm.py:877-885) with X rebuilt from that same previous iterate
            if (var == SPORCO_AMD_VAR_YPREV) {
                SA_HIP(hipMemcpyAsync(rv(var), y_alt, sizeof(T) * E, hipMemcpyDeviceToDevice, st));
            } else {
                const bool keep = prev_in_alt;
                materialize_x();
                prev_in_alt = keep;
                ProfScope ps(prof, PS_OTHER);
                launch_relax<T>(st, rv(SPORCO_AMD_VAR_X), y_alt, rv(SPORCO_AMD_VAR_AX),
                                (T)last_p.rlx, E);
            }
        }
        before_read(var);
        host_copy(var, dst, false);
        sync();
    }
    void *device_ptr(int var) override {
        before_read(var);
        return var_ptr(var);
    }

    void read_out(const double *out_dev, double *out_host) override {
        SA_HIP(hipMemcpyAsync(out_pinned, out_dev, sizeof(double) * kOutSlots, hipMemcpyDeviceToHost,
                              st));
        sync();
        std::memcpy(out_host, out_pinned, sizeof(double) * kOutSlots);
    }

    void require_single_channel_dict() const {
        if (Cd > 1)
            throw Error(SPORCO_AMD_EINVAL,
                        "multi-channel dictionaries are handled by the ADMM ConvBPDN calls only");
    }
    void require_ready() const {
        if (!have_dict || !have_signal)
            throw Error(SPORCO_AMD_ESTATE, "set_signal and set_dict must be called first");
    }

    // column FFT + Sherman-Morrison + column IFFT on the tile-major spectrum in the
    // Xf buffer (csc_fused.h); the data-fidelity sum goes to out_dev when wanted
    void run_fused_cols(const sporco_amd_admm_params &p, double *out_dev) {
        if (fused_mc) {
            SA_REQUIRE(!(p.flags & (F_GRADREG | F_AMS | F_JOINT)),
                       "this solver variant needs a single-channel dictionary");
            if (!binv_valid || binv_rho != p.rho) {
                ProfScope ps(prof, PS_OTHER);
                launch_mc_binv<T>(st, dft_mc, bt_mc, npix, Cd, K, (T)p.rho);
                binv_valid = true;
                binv_rho = p.rho;
            }
            FusedMcArgs<T> ma;
            ma.t = cv(SPORCO_AMD_VAR_XF);
            ma.dft = dft_mc;
            ma.sft = sft_mc;
            ma.bt = bt_mc;
            ma.twA = twA;
            ma.twB = twB;
            ma.rho = (T)p.rho;
            ma.H = H;
            ma.W = W;
            ma.N = N;
            ma.K = K;
            ma.Cd = Cd;
            ma.partials = part_f;
            int64_t nt;
            {
                ProfScope ps(prof, PS_FUSED_COLS);
                nt = launch_fused_cols_mc<T>(st, ma);
            }
            part_f_rows = (int)nt;
            xf_tiled = true;
            if (out_dev && (p.flags & F_OBJ) && !(p.flags & F_FEVAL_Y)) {
                const int slots[1] = {SPORCO_AMD_OUT_DFID};
                const double scales[1] = {1.0 / ((double)H * W)};
                finalize(part_f, (int)nt, 1, 1, slots, scales, out_dev);
            }
            return;
        }
        FusedColsArgs<T> fa;
        fa.t = cv(SPORCO_AMD_VAR_XF);
        fa.dft = dft;
        fa.sft = sft;
        fa.gramt = gramt;
        fa.twA = twA;
        fa.twB = twB;
        fa.rho = (T)p.rho;
        fa.H = H;
        fa.W = W;
        fa.CN = CN;
        fa.K = K;
        fa.partials = part_f;
        fa.Ks = Ks;
        const bool gradreg = p.flags & F_GRADREG;
        const bool tail_ok = tail_mode;
        if (gradreg) {
            SA_REQUIRE(fused || fused_slabs, "no gradient-regularised column pass for this shape");
            const GradTerm<T> gt = grad_term(p.mu);
            if (!g1t) SA_HIP(hipMalloc((void **)&g1t, sizeof(T) * npix));
            fa.ghh = gt.ghh;
            fa.ghw = gt.ghw;
            fa.wg = gt.wg;
            fa.mu = gt.mu;
            fa.g1t_out = g1t;
            if (!g1_valid || g1_rho != p.rho || g1_mu != p.mu) {
                ProfScope ps(prof, PS_OTHER);
                launch_grad_g1<T>(st, fa);
                g1_valid = true;
                g1_rho = p.rho;
                g1_mu = p.mu;
            }
            fa.g1t = g1t;
        }
        int64_t ntiles;
        if (tail_ok) {
            // A handful of filters past 64 (the AddMaskSim impulse on a 64-filter dictionary):
            // they go through the generic column FFT, their inner products are folded into
            // Sf, and the register-resident kernel runs on the first 64 as if alone.
            if (!sft_eff) {
                SA_HIP(hipMalloc((void **)&sft_eff, sizeof(cx<T>) * npix * CN));
                SA_HIP(hipMalloc((void **)&coef_t, sizeof(cx<T>) * npix * CN));
            }
            const int Kt = K - 64;
            const int64_t nt = (int64_t)Wf * CN, tstride = (int64_t)H * Ks;
            cx<T> *tail = fa.t + 64;
            fa.Kv = 64;
            fa.Ks = Ks;
            fa.coef_out = coef_t;
            ProfScope ps(prof, PS_FUSED_COLS);
            fft_c2c<T>(st, planH, false, tail, tail, nt, Kt, tstride, Ks, tstride, Ks, T(1));
            launch_tail_inner<T>(st, fa, sft, sft_eff);
            fa.sft = sft_eff;
            ntiles = launch_fused_cols<T>(st, fa);
            launch_tail_update<T>(st, fa);
            fft_c2c<T>(st, planH, true, tail, tail, nt, Kt, tstride, Ks, tstride, Ks, T(1));
        } else if (fused_slabs) {
            FusedSlabArgs<T> sa;
            sa.c = fa;
            sa.qpart = qpart;
            ProfScope ps(prof, PS_FUSED_COLS);
            ntiles = run_slab_cols(sa);
            if (gradreg) ntiles *= (K + 63) / 64;   // (one row of partials per tile and slab)
        } else {
            ProfScope ps(prof, PS_FUSED_COLS);
            ntiles = launch_fused_cols<T>(st, fa);
        }
        part_f_rows = (int)ntiles;
        xf_tiled = true;
        if (out_dev && (p.flags & F_OBJ) && !(p.flags & F_FEVAL_Y)) {
            const int slots[2] = {SPORCO_AMD_OUT_DFID, SPORCO_AMD_OUT_RGR};
            const double scales[2] = {1.0 / ((double)H * W), 1.0 / ((double)H * W)};
            const int nv = gradreg ? 2 : 1;
            finalize(part_f, (int)ntiles, nv, nv, slots, scales, out_dev);
        }
    }

    // One whole ADMM iteration in three launches (csc_rows.h): rows_fwd, fused_cols,
    // rows_inv_post.  X is written only on request (F_KEEP_X).
    void admm_iter_fused(const sporco_amd_admm_params &p, double *out_dev) {
        require_ready();
        const bool keep_x = p.flags & F_KEEP_X;
        // Single-array state (csc_rows.h): from the second fused iteration in a row with no host
        // access to the iterates in between, the epilogue stores V' = AX + U alone and the next
        // iteration derives (Y, U) from it -- seven passes instead of ten (six instead of eight
        // with an emitted spectrum).  Anything else that wants Y or U gets them through
        // ensure_yu() (var_ptr).
        const bool nn = p.flags & F_NONNEG, jn = p.flags & F_JOINT;
        if (v_live && (!vform_ok(p) || !vform_same_opts(p))) ensure_yu();
        const bool vf = vform_ok(p) && (v_live || touch_epoch == fused_epoch);
        T *vin = vf && v_live ? v_cur : nullptr;
        T *Y = vin ? nullptr : rv(SPORCO_AMD_VAR_Y), *U = vin ? nullptr : rv(SPORCO_AMD_VAR_U);
        cx<T> *Xf = cv(SPORCO_AMD_VAR_XF);
        if (!keep_x && !y_alt) {
            SA_HIP(hipMalloc((void **)&y_alt, sizeof(T) * E));
            SA_HIP(hipMalloc((void **)&u_alt, sizeof(T) * E));
        }
        T *vout = vf ? (vin == y_alt ? u_alt : y_alt) : nullptr;
        // rows_fwd, unless the previous iteration already left its result behind
        if (!(t_ready && p.u_scale == 1.0)) {
            if (vin) launch_rows_fwd_on(nullptr, nullptr, (T)p.u_scale, vin, v_thr, p.flags, v_thr21, &p);
            else launch_rows_fwd_on(Y, U, (T)p.u_scale);
        }
        t_ready = false;
        run_fused_cols(p, nullptr);
        // Bet on an unchanged rho only once it has stayed put for two updates in a row:
        // while AutoRho is still moving it almost every iteration a lost bet costs one
        // extra pass, a won one saves three (rows_fwd of the next iteration).
        stable_run = p.u_scale == 1.0 ? stable_run + 1 : 0;
        // (ConvBPDNJoint speculates too since round 3: its emitting epilogue used to spill the
        // tile -- 22.4 ms against 13.0 + 6.6 ms for epilogue + rows_fwd at config 3,
        // profiles/r02i_config3_*.json -- until the two elements of a pixel were serialised;
        // SPORCO_AMD_JOINT_EMIT=0 switches it off)
        const bool emit = stable_run >= 2 && !std::getenv("SPORCO_AMD_NO_SPECULATION") &&
                          !((p.flags & F_JOINT) && joint_emit_off());
        RowsPostArgs<T> pa;
        pa.twA = twRows;
        pa.t_next = emit ? Xf : nullptr;
        pa.t = Xf;
        pa.twW = planW.tw<T>();
        pa.y = Y;
        pa.u = U;
        pa.y_out = keep_x ? Y : y_alt;
        pa.u_out = keep_x ? U : u_alt;
        pa.v_in = vin;
        pa.v_out = vout;
        pa.thr_prev = v_thr;
        pa.thr21_prev = v_thr21;
        pa.x = keep_x ? rv(SPORCO_AMD_VAR_X) : nullptr;
        pa.scale = T(1.0 / ((double)H * (double)W));
        pa.rlx = (T)p.rlx;
        pa.thr = (T)(p.lmbda / p.rho);
        pa.thr21 = (T)(p.mu / p.rho);
        pa.u_scale = (T)p.u_scale;
        pa.flags = p.flags;
        pa.H = H;
        pa.W = W;
        pa.C = C;
        pa.N = N;
        pa.K = K;
        pa.dH = p.dH;
        pa.dW = p.dW;
        pa.P = P;
        pa.wl1 = wl1;
        pa.Ks = Ks;
        pa.ams_bits = ams_bits_of(p);
        pa.ams_k = Ku - 1;
        pa.partials = part_rows;
        int64_t nt;
        {
            ProfScope ps(prof, vin ? (emit ? PS_ROWS_INV_POST_V_EMIT : PS_ROWS_INV_POST_V)
                                   : (emit ? PS_ROWS_INV_POST_EMIT : PS_ROWS_INV_POST));
            nt = launch_rows_inv_post<T>(st, pa);
        }
        if (p.flags & (F_RESID | F_OBJ)) {
            // one launch sums both partial arrays: the six (joint: seven) epilogue sums and the
            // data-fidelity term of the column kernel
            const int slots[7] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_S2, SPORCO_AMD_OUT_AX2,
                                  SPORCO_AMD_OUT_Y2, SPORCO_AMD_OUT_U2, SPORCO_AMD_OUT_L1,
                                  SPORCO_AMD_OUT_L21};
            const double scales[7] = {1, 1, 1, 1, 1, 1, 1};
            const int nrow = (p.flags & F_JOINT) ? 7 : 6;
            const int fslots[2] = {SPORCO_AMD_OUT_DFID, SPORCO_AMD_OUT_RGR};
            const double fscales[2] = {1.0 / ((double)H * W), 1.0 / ((double)H * W)};
            const bool dfid = (p.flags & F_OBJ) && !(p.flags & F_FEVAL_Y);
            const int fnv = (p.flags & F_GRADREG) ? 2 : 1;
            ProfScope ps(prof, PS_FINALIZE);
            launch_finalize2(st, part_rows, (int)nt, 8, nrow, slots, scales, part_f, part_f_rows, fnv,
                             dfid ? fnv : 0, fslots, fscales, out_dev);
        }
        if (keep_x) {
            x_written();
        } else if (vf) {
            // the new iterate is the V in vout; the previous one is (Y, U) in vars (first V
            // iteration) or the V this iteration read
            v_prev_kind = vin ? 2 : 1;
            v_prev_thr = v_thr;
            v_prev_thr21 = v_thr21;
            v_cur = vout;
            v_thr = pa.thr;
            v_thr21 = pa.thr21;
            v_nonneg = nn;
            v_joint = jn;
            v_opts = p.flags & (F_NOBNDRY | F_AMS);
            v_dH = p.dH;
            v_dW = p.dW;
            v_live = true;
            vp_pending = false;
            last_p = p;
            x_stale = true;
            x_invalid = p.flags & F_NO_X;
            prev_in_alt = false;
        } else {
            // the new iterate lives in the alternate buffers: swap roles
            std::swap(vars[SPORCO_AMD_VAR_Y], reinterpret_cast<void *&>(y_alt));
            std::swap(vars[SPORCO_AMD_VAR_U], reinterpret_cast<void *&>(u_alt));
            vp_pending = false;
            last_p = p;
            x_stale = true;
            x_invalid = p.flags & F_NO_X;
            prev_in_alt = true;
        }
        t_ready = emit;
        if ((p.flags & F_OBJ) && (p.flags & F_FEVAL_Y)) dfid_at(rv(SPORCO_AMD_VAR_Y), out_dev, &p);
        fused_epoch = touch_epoch;
    }

    // ---- device-driven solve (include/sporco_amd.h: sporco_amd_csc_admm_run) -----------------
    bool admm_run_supported(const sporco_amd_admm_params &p) const {
        return std::is_same<T, float>::value && rows_ok && (fused || fused_slabs) && !fused_mc &&
               !tail_mode && !(p.flags & (F_XRRS | F_GRADREG | F_KEEP_X | F_FEVAL_Y)) &&
               (!(p.flags & F_JOINT) || joint_rows_ok(p)) && p.lmbda >= 0.0 && p.rho > 0.0 &&
               !std::getenv("SPORCO_AMD_HOST_LOOP");
    }

    // The run as one launch (csc_rows.h admm_persist): small problems, plain options, the
    // single-array state, no collective between the sums and the control update.  Opt-in
    // (SPORCO_AMD_PERSIST=1 or SPORCO_AMD_HINT_ONE_LAUNCH): it needs the whole device to itself
    // (its workgroups wait for each other), it buys 9-12 % where it applies
    // (profiles/r03_persist.md), and on the GPU its iterates equal those of the launch-per-pass
    // loop to rounding, not bit for bit.
    bool persist_ok(const sporco_amd_admm_params &p, const sporco_amd_admm_ctrl &c, bool vf,
                    bool has_reduce) const {
        const char *e = std::getenv("SPORCO_AMD_PERSIST");
        const bool off = e ? e[0] != '1' : !hint_one_launch;
        return !off && vf && !has_reduce && fused && !fused_slabs && Ks == K && run_always_emit &&
               admm_persist_supported<T>(H, W, K) && !wl1.ptr && c.max_iter >= 3 &&
               !(p.flags & (F_NOBNDRY | F_AMS | F_JOINT));
    }
    PersistIterArgs<T> persist_iter_args(const sporco_amd_admm_params &p, const T *vin, T *vout,
                                         double *prow, double *pcol) {
        PersistIterArgs<T> a;
        cx<T> *Xf = cv(SPORCO_AMD_VAR_XF);
        a.fwd.y = a.fwd.u = nullptr;
        a.fwd.v = vin;
        a.fwd.flags = p.flags;
        a.fwd.C = C;
        a.fwd.N = N;
        a.fwd.dH = p.dH;
        a.fwd.dW = p.dW;
        a.fwd.s2 = T(1);
        a.fwd.t = Xf;
        a.fwd.Ks = Ks;
        a.fwd.twA = twRows;
        a.fwd.H = H;
        a.fwd.W = W;
        a.fwd.CN = CN;
        a.fwd.K = K;
        a.fwd.P = P;
        a.cols.t = Xf;
        a.cols.dft = dft;
        a.cols.sft = sft;
        a.cols.gramt = gramt;
        a.cols.twA = twA;
        a.cols.twB = twB;
        a.cols.rho = (T)p.rho;
        a.cols.H = H;
        a.cols.W = W;
        a.cols.CN = CN;
        a.cols.K = K;
        a.cols.partials = pcol;
        a.cols.Ks = Ks;
        a.post.twA = twRows;
        a.post.t_next = Xf;
        a.post.t = Xf;
        a.post.twW = planW.tw<T>();
        a.post.y = a.post.u = nullptr;
        a.post.y_out = a.post.u_out = nullptr;
        a.post.v_in = vin;
        a.post.v_out = vout;
        a.post.x = nullptr;
        a.post.scale = T(1.0 / ((double)H * (double)W));
        a.post.rlx = (T)p.rlx;
        a.post.thr = T(0);
        a.post.u_scale = T(1);
        a.post.flags = p.flags;
        a.post.H = H;
        a.post.W = W;
        a.post.C = C;
        a.post.N = N;
        a.post.K = K;
        a.post.dH = p.dH;
        a.post.dW = p.dW;
        a.post.P = P;
        a.post.Ks = Ks;
        a.post.partials = prow;
        return a;
    }
    // iterations index0 .. max_iter - 1 of the run whose control block is ctl_dev; v_prev: the
    // iterate iteration index0 - 1 left, v_other: the other V buffer.  Returns how many ran.
    int run_persist(const sporco_amd_admm_params &p, int index0, int max_iter, T *v_prev, T *v_other,
                    bool want_sums) {
        const int grid = admm_persist_grid<T>(H, W, K, CN);
        const int64_t nrow = (int64_t)H * ceil_div(P, 128), ncol = (int64_t)Wf * CN;
        if (!pst_part_rows) {
            SA_HIP(hipMalloc((void **)&pst_part_rows, sizeof(double) * 8 * nrow));
            SA_HIP(hipMalloc((void **)&pst_part_f, sizeof(double) * 2 * ncol));
            SA_HIP(hipMalloc((void **)&pst_bar, sizeof(unsigned) * kPersistBarWords));
        }
        if (pst_grid < grid) {
            if (pst_blk) {
                sync();
                SA_HIP(hipFree(pst_blk));
                SA_HIP(hipFree(pst_ctl));
            }
            SA_HIP(hipMalloc(&pst_blk, sizeof(PersistIterArgs<T>) * 2 * grid));
            SA_HIP(hipMalloc((void **)&pst_ctl, sizeof(AdmmCtl) * grid));
            pst_grid = grid;
        }
        AdmmPersistArgs<T> a;
        // iteration j reads the V that iteration j - 1 wrote
        const int p0 = index0 & 1;
        a.iter[p0] = persist_iter_args(p, v_prev, v_other, part_rows, part_f);
        a.iter[p0 ^ 1] = persist_iter_args(p, v_other, v_prev, pst_part_rows, pst_part_f);
        a.blk = static_cast<PersistIterArgs<T> *>(pst_blk);
        a.ctl_blk = pst_ctl;
        a.ctl = ctl_dev;
        a.rec = reinterpret_cast<AdmmRecord *>(rec_ring) + index0;
        a.index0 = index0;
        a.max_iter = max_iter - index0;
        a.bar = pst_bar;
        a.n_row_tiles = (int)nrow;
        a.n_col_tiles = (int)ncol;
        a.want_dfid = (p.flags & F_OBJ) ? 1 : 0;
        a.want_sums = want_sums ? 1 : 0;
        a.dfid_scale = 1.0 / ((double)H * W);
        SA_HIP(hipMemsetAsync(pst_bar, 0, sizeof(unsigned) * kPersistBarWords, st));
        {
            ProfScope ps(prof, PS_PERSIST);
            launch_admm_persist<T>(st, a, grid);
        }
        unsigned bar[16];
        SA_HIP(hipMemcpyAsync(bar, pst_bar, sizeof(bar), hipMemcpyDeviceToHost, st));
        sync();
        if (std::getenv("SPORCO_AMD_PERSIST_TIMING") && bar[3])      // (measurement builds fill these)
            std::fprintf(stderr, "admm_persist: %u iterations; ticks (10 ns) per iteration: fwd %.0f bar %.0f | cols %.0f "
                         "bar %.0f | post %.0f bar %.0f | sums %.0f ctl %.0f\n", bar[3], bar[8] / (double)bar[3],
                         bar[9] / (double)bar[3], bar[10] / (double)bar[3], bar[11] / (double)bar[3],
                         bar[12] / (double)bar[3], bar[13] / (double)bar[3], bar[14] / (double)bar[3],
                         bar[15] / (double)bar[3]);
        if (bar[2]) throw Error(SPORCO_AMD_EHIP, "one-launch solve: a grid barrier did not complete");
        xf_tiled = true;
        ++pst_runs;
        return (int)bar[3];
    }

    // one iteration of admm_iter_fused with every iteration-dependent scalar taken from ctl_dev
    // vout set: the single-array state of csc_rows.h -- the iterate is read from vin (null: from
    // (Y, U), the first iteration of such a run) and V' is written to vout
    int64_t enqueue_iter_ctl(const sporco_amd_admm_params &p, const T *vin = nullptr,
                             T *vout = nullptr) {
        T *Y = vin ? nullptr : static_cast<T *>(vars[SPORCO_AMD_VAR_Y]);
        T *U = vin ? nullptr : static_cast<T *>(vars[SPORCO_AMD_VAR_U]);
        cx<T> *Xf = cv(SPORCO_AMD_VAR_XF);
        {
            RowsFwdArgs<T> ra;
            ra.y = Y;
            ra.u = U;
            ra.v = vin;
            ra.flags = p.flags;
            ra.C = C;
            ra.N = N;
            if (vin) {
                ra.wl1 = wl1;
                ra.dH = p.dH;
                ra.dW = p.dW;
                ra.ams_bits = ams_bits_of(p);
                ra.ams_k = Ku - 1;
            }
            ra.s2 = T(1);
            ra.t = Xf;
            ra.Ks = Ks;
            ra.twA = twRows;
            ra.H = H;
            ra.W = W;
            ra.CN = CN;
            ra.K = K;
            ra.P = P;
            ra.ctl = ctl_dev;
            ProfScope ps(prof, vin ? PS_ROWS_FWD_V : PS_ROWS_FWD);
            launch_rows_fwd<T>(st, ra);
        }
        {
            FusedColsArgs<T> fa;
            fa.t = Xf;
            fa.dft = dft;
            fa.sft = sft;
            fa.gramt = gramt;
            fa.twA = twA;
            fa.twB = twB;
            fa.rho = (T)p.rho;
            fa.H = H;
            fa.W = W;
            fa.CN = CN;
            fa.K = K;
            fa.partials = part_f;
            fa.Ks = Ks;
            fa.ctl = ctl_dev;
            ProfScope ps(prof, PS_FUSED_COLS);
            if (fused_slabs) {      // 64 < K <= 256: the two slab kernels (csc_fused.h)
                FusedSlabArgs<T> sa;
                sa.c = fa;
                sa.qpart = qpart;
                part_f_rows = (int)run_slab_cols(sa);
            } else {
                part_f_rows = (int)launch_fused_cols<T>(st, fa);
            }
            xf_tiled = true;
        }
        RowsPostArgs<T> pa;
        pa.twA = twRows;
        pa.t_next = nullptr;     // plain variant first, then the emitting one: ctl->emit picks
        pa.t = Xf;
        pa.twW = planW.tw<T>();
        pa.y = Y;
        pa.u = U;
        pa.y_out = y_alt;
        pa.u_out = u_alt;
        pa.v_in = vin;
        pa.v_out = vout;
        pa.x = nullptr;
        pa.scale = T(1.0 / ((double)H * (double)W));
        pa.rlx = (T)p.rlx;
        pa.thr = T(0);
        pa.u_scale = T(1);
        pa.flags = p.flags;
        pa.H = H;
        pa.W = W;
        pa.C = C;
        pa.N = N;
        pa.K = K;
        pa.dH = p.dH;
        pa.dW = p.dW;
        pa.P = P;
        pa.wl1 = wl1;
        pa.Ks = Ks;
        pa.ams_bits = ams_bits_of(p);
        pa.ams_k = Ku - 1;
        pa.partials = part_rows;
        pa.ctl = ctl_dev;
        int64_t nt = 0;
        if (!run_always_emit) {
            ProfScope ps(prof, vin ? PS_ROWS_INV_POST_V : PS_ROWS_INV_POST);
            nt = launch_rows_inv_post<T>(st, pa);
        }
        pa.t_next = Xf;
        {
            ProfScope ps(prof, vin ? PS_ROWS_INV_POST_V_EMIT : PS_ROWS_INV_POST_EMIT);
            nt = launch_rows_inv_post<T>(st, pa);
        }
        if (!vout) {
            std::swap(vars[SPORCO_AMD_VAR_Y], reinterpret_cast<void *&>(y_alt));
            std::swap(vars[SPORCO_AMD_VAR_U], reinterpret_cast<void *&>(u_alt));
        }
        return nt;
    }

    int admm_run(const sporco_amd_admm_params &p, const sporco_amd_admm_ctrl &c,
                 sporco_amd_admm_record *records, double *rho_out, double *u_scale_out,
                 sporco_amd_reduce_fn reduce, void *user) override {
        require_ready();
        if (!admm_run_supported(p)) return -1;
        SA_REQUIRE(c.max_iter >= 0, "max_iter must not be negative");
        if (c.max_iter == 0) {
            *rho_out = p.rho;
            *u_scale_out = p.u_scale;
            return 0;
        }
        if (!y_alt) {
            SA_HIP(hipMalloc((void **)&y_alt, sizeof(T) * E));
            SA_HIP(hipMalloc((void **)&u_alt, sizeof(T) * E));
        }
        // Single-array state (csc_rows.h) for runs of several iterations: the epilogue stores
        // V' = AX + U alone, rows_fwd and the next epilogue derive (Y, U) from it.  A run of a
        // few iterations (a dictionary-learning X-step) stays in the (Y, U) form: it would pay
        // the conversion back at once.
        const bool nn = p.flags & F_NONNEG, jn = p.flags & F_JOINT;
        if (v_live && (!vform_ok(p) || !vform_same_opts(p))) ensure_yu();
        const bool vf = vform_ok(p) && (v_live || c.max_iter >= 4 || hint_vform);
        if (!vf) ensure_yu();
        const bool v_at_entry = v_live;
        T *const v_entry = v_cur;
        const T v_entry_thr = v_thr, v_entry_thr21 = v_thr21;
        if (!ctl_dev) SA_HIP(hipMalloc((void **)&ctl_dev, sizeof(AdmmCtl)));
        if (rec_cap < c.max_iter) {
            if (rec_ring) SA_HIP(hipHostFree(rec_ring));
            rec_cap = std::max(c.max_iter, 256);
            SA_HIP(hipHostMalloc((void **)&rec_ring, sizeof(AdmmRecord) * rec_cap, 0));
        }
        std::memset((void *)rec_ring, 0, sizeof(AdmmRecord) * c.max_iter);
        double *out_dev = out_dev_default;
        const bool want_sums = p.flags & (F_RESID | F_OBJ);
        AdmmCtlInit in;
        in.rho = p.rho;
        in.u_scale = p.u_scale;
        in.lmbda = p.lmbda;
        in.abstol = c.abs_tol;
        in.reltol = c.rel_tol;
        in.sqrt_nc = c.sqrt_nc;
        in.sqrt_nx = c.sqrt_nx;
        in.tau = c.rho_tau;
        in.mu = c.rho_mu;
        in.xi = c.rho_xi;
        in.mu21 = p.mu;
        in.k = c.k0;
        in.stable_run = stable_run;
        in.emitted = t_ready ? 1 : 0;
        in.is_f32 = 1;
        in.autorho = c.auto_rho;
        in.period = c.period > 0 ? c.period : 1;
        in.autoscaling = c.auto_scaling;
        in.stdres = c.std_residuals;
        in.need_resid = c.need_residuals;
        in.thr_prev = v_at_entry ? (float)v_entry_thr : 0.f;
        in.thr21_prev = v_at_entry ? (float)v_entry_thr21 : 0.f;
        in.no_speculation = (std::getenv("SPORCO_AMD_NO_SPECULATION") ||
                             ((p.flags & F_JOINT) && joint_emit_off()))
                                ? 1
                                : 0;
        // Small problems (kernels of a few microseconds): the emitting epilogue always, and its
        // plain twin is not enqueued at all -- a wasted emit costs less than a launch that
        // returns at once (SPORCO_AMD_RUN_ALWAYS_EMIT=0/1 overrides the size rule).
        {
            const char *e = std::getenv("SPORCO_AMD_RUN_ALWAYS_EMIT");
            run_always_emit = !in.no_speculation && (e ? std::atoi(e) != 0 : E <= ((int64_t)1 << 22));
            if (run_always_emit) in.no_speculation = 2;
        }
        launch_admm_ctl_init(st, ctl_dev, in);
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        const int ahead = c.lookahead > 0 ? c.lookahead : 3;
        int enq = 0, done = 0, stop_at = -1;
        // (test knob: pretend the newest `lag` records are not visible yet, so that launches
        // enqueued past the stopping iteration -- which must do nothing -- occur on any device)
        int lag = std::getenv("SPORCO_AMD_RUN_LAG") ? std::atoi(std::getenv("SPORCO_AMD_RUN_LAG")) : 0;
        // Advance `done` over finished iterations up to `upto` records: without `block` it
        // returns at the first unfinished one, with `block` it waits for each.
        auto poll = [&](int upto, bool block) {
            while (done < upto && stop_at < 0) {
                if (rec_ring[done].seq != done + 1) {
                    if (!block) return;
                    if (hipStreamQuery(st) == hipSuccess && rec_ring[done].seq != done + 1)
                        throw Error(SPORCO_AMD_EHIP, "device-driven solve: record not written");
                    continue;
                }
                if (rec_ring[done].stop) stop_at = done;
                ++done;
            }
        };
        T *vb_in = v_at_entry ? v_entry : nullptr;     // V form: input of the next enqueued iteration
        T *const vb_first = vf ? (vb_in == y_alt ? u_alt : y_alt) : nullptr;   // output of the first
        auto enqueue_one = [&]() {
            T *vb_out = vf ? (vb_in == y_alt ? u_alt : y_alt) : nullptr;
            const int64_t nt = enqueue_iter_ctl(p, vb_in, vb_out);
            vb_in = vb_out;
            if (want_sums) {
                const int slots[7] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_S2, SPORCO_AMD_OUT_AX2,
                                      SPORCO_AMD_OUT_Y2, SPORCO_AMD_OUT_U2, SPORCO_AMD_OUT_L1,
                                      SPORCO_AMD_OUT_L21};
                const double scales[7] = {1, 1, 1, 1, 1, 1, 1};
                const int fslots[2] = {SPORCO_AMD_OUT_DFID, SPORCO_AMD_OUT_RGR};
                const double fscales[2] = {1.0 / ((double)H * W), 1.0 / ((double)H * W)};
                const bool dfid = p.flags & F_OBJ;
                {
                    ProfScope ps(prof, PS_FINALIZE);
                    launch_finalize2(st, part_rows, (int)nt, 8, (p.flags & F_JOINT) ? 7 : 6, slots,
                                     scales, part_f, part_f_rows,
                                     1, dfid ? 1 : 0, fslots, fscales, out_dev);
                }
                if (reduce) reduce(user, out_dev);
            }
            launch_admm_ctl_update(st, ctl_dev, out_dev, rec_ring + enq, enq, true);
            ++enq;
        };
        // Iteration e is enqueued once the records of iterations < e - ahead have been seen
        // (a sliding window: the device always has up to `ahead` iterations queued).  With
        // image shards (`reduce`) that window is exact -- the blocking wait ignores the lag
        // knob -- so a rank never has more than stop + 1 + ahead iterations enqueued, and
        // every rank is topped up to exactly that many below: the number of collectives is
        // the same on all ranks however late each host notices the stop (the surplus
        // iterations are launches that return at once around an all-reduce nobody reads).
        const int wlag = reduce ? 0 : lag;
        if (persist_ok(p, c, vf, reduce != nullptr)) {
            // small problem: the first iteration as usual (it enters the single-array state),
            // every further one inside one launch
            enqueue_one();
            T *v0 = vb_in, *v1 = (v0 == y_alt) ? u_alt : y_alt;
            enq += run_persist(p, 1, c.max_iter, v0, v1, want_sums);
            if (c.need_residuals) poll(enq, false);
        }
        for (; enq < c.max_iter && stop_at < 0 && !persist_ok(p, c, vf, reduce != nullptr);) {
            enqueue_one();
            if (c.need_residuals) {
                poll(enq - lag, false);
                if (stop_at < 0 && enq - wlag - done > ahead) poll(enq - wlag - ahead, true);
            }
        }
        if (reduce && want_sums && c.need_residuals) {
            if (stop_at < 0) poll(enq, true);       // (the last window: a stop may sit in it)
            if (stop_at >= 0)
                while (enq < c.max_iter && enq < stop_at + 1 + ahead) enqueue_one();
        }
        sync();
        poll(enq, false);
        const int n = stop_at >= 0 ? stop_at + 1 : enq;
        // launches enqueued after the stopping iteration did nothing: undo their buffer swaps
        if (!vf && ((enq - n) & 1)) {
            std::swap(vars[SPORCO_AMD_VAR_Y], reinterpret_cast<void *&>(y_alt));
            std::swap(vars[SPORCO_AMD_VAR_U], reinterpret_cast<void *&>(u_alt));
        }
        if (vf) {
            // iteration j wrote its V' to vb_first (j even) or to the other alt buffer (j odd);
            // thresholds as the control block formed them: (float)(lambda / rho of the iteration)
            T *other_first = vb_first == y_alt ? u_alt : y_alt;
            v_cur = ((n - 1) & 1) ? other_first : vb_first;
            v_thr = (T)(p.lmbda / rec_ring[n - 1].rho);
            v_thr21 = (T)(p.mu / rec_ring[n - 1].rho);
            if (n >= 2) {
                v_prev_kind = 2;
                v_prev_thr = (T)(p.lmbda / rec_ring[n - 2].rho);
                v_prev_thr21 = (T)(p.mu / rec_ring[n - 2].rho);
            } else if (v_at_entry) {
                v_prev_kind = 2;
                v_prev_thr = v_entry_thr;
                v_prev_thr21 = v_entry_thr21;
            } else {
                v_prev_kind = 1;
            }
            v_nonneg = nn;
            v_joint = jn;
            v_opts = p.flags & (F_NOBNDRY | F_AMS);
            v_dH = p.dH;
            v_dW = p.dW;
            v_live = true;
        }
        vp_pending = false;
        AdmmCtl fin;
        SA_HIP(hipMemcpy(&fin, ctl_dev, sizeof(AdmmCtl), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) {
            const AdmmRecord &r = rec_ring[i];
            sporco_amd_admm_record &o = records[i];
            for (int j = 0; j < kOutSlots; ++j) o.sums[j] = r.sums[j];
            o.r = r.r;
            o.s = r.s;
            o.epri = r.epri;
            o.edua = r.edua;
            o.rho = r.rho;
            o.u_scale = r.u_scale;
            o.seconds = (double)r.ticks * 1e-8;
            o.k = r.k;
            o.stop = r.stop;
        }
        *rho_out = fin.rho;
        *u_scale_out = fin.u_scale;
        // host-side mirrors of the state the per-iteration path keeps
        stable_run = fin.stable_run > 0 ? fin.stable_run - 1 : 0;   // (admm_iter_fused re-derives it)
        if (fin.u_scale != 1.0) stable_run = 0;
        t_ready = fin.emitted != 0;
        last_p = p;
        last_p.rho = rec_ring[n - 1].rho;
        last_p.u_scale = rec_ring[n - 1].u_scale;
        x_stale = true;
        x_invalid = p.flags & F_NO_X;
        prev_in_alt = !vf;
        return n;
    }

    // X = irfft_W(tile-major spectrum in the Xf buffer) / (H W): the row pass of
    // rows_inv_prox_fwd with a zero threshold (soft(v, 0) = v) and no forward half
    void rows_inverse_to(T *Xout, const cx<T> *t_in = nullptr) {
        RowsProxArgs<T> ra;
        ra.t_in = t_in ? t_in : cv(SPORCO_AMD_VAR_XF);
        ra.Ks = t_in ? 0 : Ks;
        ra.t_out = nullptr;
        ra.x = Xout;
        ra.twA = twRows;
        ra.twW = planW.tw<T>();
        ra.scale = T(1.0 / ((double)H * (double)W));
        ra.thr = T(0);
        ra.flags = 0;
        ra.H = H;
        ra.W = W;
        ra.C = C;
        ra.N = N;
        ra.K = K;
        ra.dH = 1;
        ra.dW = 1;
        ra.P = P;
        ra.wl1 = Weight<T>();
        ra.partials = part_rows;
        ProfScope ps(prof, PS_FFT_C2R);
        launch_rows_inv_prox_fwd<T>(st, ra);
    }

    void launch_rows_fwd_on(const T *Yin, const T *Uin, T s2, const T *Vin = nullptr,
                            T thr_prev = T(0), uint32_t flags = 0, T thr21_prev = T(0),
                            const sporco_amd_admm_params *vp = nullptr) {
        RowsFwdArgs<T> ra;
        if (Vin && vp) {     // the options the derivation of Y from V repeats
            ra.wl1 = wl1;
            ra.dH = vp->dH;
            ra.dW = vp->dW;
            ra.ams_bits = ams_bits_of(*vp);
            ra.ams_k = Ku - 1;
        }
        ra.y = Yin;
        ra.u = Uin;
        ra.v = Vin;
        ra.thr_prev = thr_prev;
        ra.thr21_prev = thr21_prev;
        ra.flags = flags;
        ra.C = C;
        ra.N = N;
        ra.s2 = s2;
        ra.t = cv(SPORCO_AMD_VAR_XF);
        ra.Ks = Ks;
        ra.twA = twRows;
        ra.H = H;
        ra.W = W;
        ra.CN = CN;
        ra.K = K;
        ra.P = P;
        ProfScope ps(prof, Vin ? PS_ROWS_FWD_V : PS_ROWS_FWD);
        launch_rows_fwd<T>(st, ra);
    }

    // ---- ADMM --------------------------------------------------------------------
    // X-step: Xf = SM(rfftn(Y - s U)), X = irfftn(Xf); objective / check sums -> out_dev
    void xstep_impl(const sporco_amd_admm_params &p, double *out_dev) {
        require_ready();
        x_written();
        t_ready = false;
        T *Y = rv(SPORCO_AMD_VAR_Y), *U = rv(SPORCO_AMD_VAR_U), *X = rv(SPORCO_AMD_VAR_X);
        cx<T> *Xf = cv(SPORCO_AMD_VAR_XF);
        const bool gradreg = p.flags & F_GRADREG;
        if ((fused || (fused_slabs && rows_ok) || (fused_mc && rows_ok)) &&
            !(p.flags & F_XRRS)) {
            // rows -> [column FFT, Sherman-Morrison, column IFFT] in registers -> rows,
            // through the tile-major intermediate T[wf][cn][h][k] held in the Xf buffer
            const int64_t tline = (int64_t)CN * H * K, tgrp = (int64_t)H * K;
            if (rows_ok) {
                // register-resident row passes around it (ConvBPDNJoint, staged callers)
                launch_rows_fwd_on(Y, U, (T)p.u_scale);
                run_fused_cols(p, out_dev);
                rows_inverse_to(X);
                return;
            }
            {
                ProfScope ps(prof, PS_FFT_R2C);
                fft_r2c<T>(st, planW, Y, U, (T)p.u_scale, Xf, H, P, (int64_t)W * P, P, K, tline, K,
                           tgrp);
            }
            run_fused_cols(p, out_dev);
            {
                ProfScope ps(prof, PS_FFT_C2R);
                fft_c2r<T>(st, planW, Xf, X, H, P, K, tline, (int64_t)W * P, P,
                           T(1.0 / ((double)H * (double)W)), K, tgrp);
            }
            return;
        }
        xf_tiled = false;
        fwd2(Y, U, (T)p.u_scale, Xf, P);
        const bool obj = (p.flags & F_OBJ) && !(p.flags & F_FEVAL_Y);
        const bool xr = p.flags & F_XRRS;
        if (Cd > 1) {
            // multi-channel dictionary: iterated Sherman-Morrison (cbpdn.py:277-279)
            // (ConvBPDNGradReg: the identity term becomes the diagonal mu wg GHGf + rho,
            // cbpdn.py:1181-1184; AddMaskSim and ConvBPDNJoint differ in the y step only)
            GradTerm<T> gtm;
            if (gradreg) gtm = grad_term(p.mu);
            const double ism_mu_now = gradreg ? p.mu : -1.0;
            if (!ism_gam) {
                SA_HIP(hipMalloc((void **)&ism_gam, sizeof(cx<T>) * npix * Cd * K));
                SA_HIP(hipMalloc((void **)&ism_del, sizeof(cx<T>) * npix * Cd));
                SA_HIP(hipMalloc((void **)&ism_mm, sizeof(cx<T>) * npix * Cd * Cd));
            }
            if (!ism_valid || ism_rho != p.rho || ism_mu != ism_mu_now) {
                ProfScope ps(prof, PS_OTHER);
                launch_ism_setup<T>(st, cv(SPORCO_AMD_VAR_DF), ism_gam, ism_del, ism_mm, npix, Cd, K,
                                    (T)p.rho, gradreg ? &gtm : nullptr, W);
                ism_valid = true;
                ism_rho = p.rho;
                ism_mu = ism_mu_now;
            }
            int nbm;
            {
                ProfScope ps(prof, PS_SM_SOLVE);
                nbm = launch_ism_solve<T>(st, Xf, Xf, cv(SPORCO_AMD_VAR_DF), cv(SPORCO_AMD_VAR_SF),
                                          ism_gam, ism_del, ism_mm, (T)p.rho, npix, Cd, N, K, W, obj, xr,
                                          part_a, gradreg ? &gtm : nullptr);
            }
            if (obj || xr) {
                const int slots[5] = {SPORCO_AMD_OUT_DFID, SPORCO_AMD_OUT_XRRS_D2,
                                      SPORCO_AMD_OUT_XRRS_AX2, SPORCO_AMD_OUT_XRRS_B2,
                                      SPORCO_AMD_OUT_RGR};
                const double scales[5] = {1.0 / ((double)H * W), 1.0, 1.0, 1.0, 1.0 / ((double)H * W)};
                const int nv = gradreg ? 5 : 4;
                finalize(part_a, nbm, nv, nv, slots, scales, out_dev);
            }
            inv2(Xf, work_buf(), X, P);
            return;
        }
        int nb;
        GradTerm<T> gt;
        if (gradreg) gt = grad_term(p.mu);
        {
            ProfScope ps(prof, PS_SM_SOLVE);
            nb = launch_sm_solve<T>(st, Xf, Xf, cv(SPORCO_AMD_VAR_DF), cv(SPORCO_AMD_VAR_SF), gram,
                                    (T)p.rho, npix, CN, K, W, obj, xr, part_a,
                                    gradreg ? &gt : nullptr);
        }
        if (obj || xr) {
            const int slots[5] = {SPORCO_AMD_OUT_DFID, SPORCO_AMD_OUT_XRRS_D2,
                                  SPORCO_AMD_OUT_XRRS_AX2, SPORCO_AMD_OUT_XRRS_B2,
                                  SPORCO_AMD_OUT_RGR};
            const double scales[5] = {1.0 / ((double)H * W), 1.0, 1.0, 1.0, 1.0 / ((double)H * W)};
            const int nv = gradreg ? 5 : 4;
            finalize(part_a, nb, nv, nv, slots, scales, out_dev);
        }
        inv2(Xf, work_buf(), X, P);
    }

    // data fidelity evaluated at Y (fEvalX False / AuxVarObj, cbpdn.py:315-321)
    // innerb(npix, Cs, N) = sum_k Df * vf   (linalg.inner over the filter axis)
    void inner_df(const cx<T> *vf) {
        if (Cd > 1)
            launch_mc_inner<T>(st, cv(SPORCO_AMD_VAR_DF), vf, innerb, npix, Cd, N, K);
        else
            launch_inner<T>(st, cv(SPORCO_AMD_VAR_DF), vf, innerb, npix, CN, K);
    }

    void dfid_at(const T *V, double *out_dev, const sporco_amd_admm_params *gp = nullptr) {
        cx<T> *wk = work_buf();
        fwd2(V, nullptr, T(0), wk, P);
        if (gp && (gp->flags & F_GRADREG)) {
            // the gradient term follows the data-fidelity variable (cbpdn.py:1209-1213)
            int nbg;
            {
                ProfScope ps(prof, PS_OTHER);
                nbg = launch_grad_norm<T>(st, wk, grad_term(gp->mu), npix, CN, K, W, part_a);
            }
            const int gslots[1] = {SPORCO_AMD_OUT_RGR};
            const double gscales[1] = {1.0 / ((double)H * W)};
            finalize(part_a, nbg, 1, 1, gslots, gscales, out_dev);
        }
        {
            ProfScope ps(prof, PS_OTHER);
            inner_df(wk);
        }
        int nb;
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_rfl2norm2<T>(st, innerb, cv(SPORCO_AMD_VAR_SF), npix, CNs, W, part_a);
        }
        const int slots[1] = {SPORCO_AMD_OUT_DFID};
        const double scales[1] = {1.0 / ((double)H * W)};
        finalize(part_a, nb, 1, 1, slots, scales, out_dev);
    }

    // 64 < K <= 64 + kTailMax: the gradient-regularised column pass is available too
    bool grad_tail_ok() const { return tail_mode; }

    // the impulse filters AddMaskSim appended: one, or one per channel of a multi-channel
    // dictionary (cbpdn.py:2339-2346); they are the last filters the caller passed
    int ams_n() const { return Cd > 1 ? Cd : 1; }
    int ams_k0() const { return Ku - ams_n(); }

    // the AddMaskSim mask, when the call asks for it (F_AMS)
    Weight<T> ams_of(const sporco_amd_admm_params &p) const {
        if (!(p.flags & F_AMS)) return Weight<T>();
        if (!wams.ptr) throw Error(SPORCO_AMD_ESTATE, "FLAG_AMS without a mask (set_ams_mask)");
        return wams;
    }

    // the same mask, one bit per pixel in the row kernel's order (built on first use)
    const uint32_t *ams_bits_of(const sporco_amd_admm_params &p) {
        if (!(p.flags & F_AMS)) return nullptr;
        const Weight<T> m = ams_of(p);
        if (!ams_bits_valid) {
            if (!ams_bits) SA_HIP(hipMalloc((void **)&ams_bits, sizeof(uint32_t) * (int64_t)H * CN * (W / 32)));
            ProfScope ps(prof, PS_OTHER);
            launch_ams_pack<T>(st, m, ams_bits, H, W, C, N);
            ams_bits_valid = true;
        }
        return ams_bits;
    }

    // ConvBPDNJoint inside the row epilogue (csc_rows.h): scalar weights, no NoBndryCross /
    // AddMaskSim, C <= 4 channels, K a multiple of 32, single-channel dictionary
    static bool joint_emit_off() {
        const char *e = std::getenv("SPORCO_AMD_JOINT_EMIT");
        return e && e[0] == '0';
    }
    bool joint_rows_ok(const sporco_amd_admm_params &p) const {
        return rows_ok && !fused_mc && rows_joint_supported<T>(W, C, K) && !wl1.ptr && !wl21.ptr &&
               !(p.flags & (F_NOBNDRY | F_AMS | F_KEEP_X | F_GRADREG)) &&
               !std::getenv("SPORCO_AMD_JOINT_SEPARATE");
    }

    void admm_iter(const sporco_amd_admm_params &p, double *out_dev) override {
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        // (a negative lambda -- meaningless, but the reference's soft threshold is defined for it --
        // goes to the generic chain: the row kernels clamp with a threshold known to be >= 0)
        if (rows_ok && !(p.flags & F_XRRS) && (!(p.flags & F_JOINT) || joint_rows_ok(p)) &&
            (fused || fused_slabs || !(p.flags & F_GRADREG)) && p.lmbda >= 0.0 && p.rho > 0.0 &&
            !(Cd > 1 && (p.flags & (F_GRADREG | F_AMS | F_JOINT)))) {
            // (a multi-channel dictionary under ConvBPDNGradReg / AddMaskSim / ConvBPDNJoint:
            // the generic chain -- the register kernels know one impulse slice and no diagonal)
            admm_iter_fused(p, out_dev);
            return;
        }
        before_state_change();
        // (LinSolveCheck evaluates its residual from X: that combination keeps the two kernels)
        {
            const char *e = std::getenv("SPORCO_AMD_C2R_POST");
            defer_c2r = e && e[0] == '1' && !(p.flags & (F_JOINT | F_XRRS));
        }
        c2r_deferred = false;
        xstep_impl(p, out_dev);
        defer_c2r = false;
        PostParams<T> pp;
        pp.x = rv(SPORCO_AMD_VAR_X);
        pp.y = rv(SPORCO_AMD_VAR_Y);
        pp.u = rv(SPORCO_AMD_VAR_U);
        pp.rlx = (T)p.rlx;
        pp.thr = (T)(p.lmbda / p.rho);
        pp.thr21 = (T)(p.mu / p.rho);
        pp.u_scale = (T)p.u_scale;
        pp.flags = p.flags;
        pp.d = d5();
        pp.dH = p.dH;
        pp.dW = p.dW;
        pp.wl1 = wl1;
        pp.wl21 = wl21;
        pp.ams = ams_of(p);
        pp.ams_k = ams_k0();
        pp.ams_n = ams_n();
        const int slots[7] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_S2, SPORCO_AMD_OUT_AX2,
                              SPORCO_AMD_OUT_Y2, SPORCO_AMD_OUT_U2, SPORCO_AMD_OUT_L1,
                              SPORCO_AMD_OUT_L21};
        const double scales[7] = {1, 1, 1, 1, 1, 1, 1};
        if (c2r_deferred) {
            // row pass of irfftn + epilogue in one kernel: X is neither written (unless the
            // caller may read it: everything but FLAG_NO_X) nor re-read
            c2r_deferred = false;
            const int64_t nblk = fft_c2r_post_blocks<T>(planW, H, P);
            if (nblk > part_c2r_cap) {
                if (part_c2r) {
                    sync();
                    SA_HIP(hipFree(part_c2r));
                }
                SA_HIP(hipMalloc((void **)&part_c2r, sizeof(double) * 8 * nblk));
                part_c2r_cap = nblk;
            }
            T *xo = (p.flags & F_NO_X) ? nullptr : rv(SPORCO_AMD_VAR_X);
            int64_t nbp;
            {
                ProfScope ps(prof, PS_FFT_C2R);
                nbp = fft_c2r_post<T>(st, planW, work_buf(), H, P, (int64_t)Wf * P, P,
                                      T(1.0 / ((double)H * (double)W)), pp, xo, part_c2r);
            }
            finalize(part_c2r, (int)nbp, 8, 7, slots, scales, out_dev);
            if (p.flags & F_NO_X) {
                x_stale = true;     // (X of this iteration does not exist: reading it is an error)
                x_invalid = true;
            } else {
                x_written();
            }
        } else {
            x_written();
            int nb;
            {
                ProfScope ps(prof, PS_ADMM_POST);
                nb = launch_admm_post<T>(st, pp, part_b);
            }
            finalize(part_b, nb, 8, 7, slots, scales, out_dev);
        }
        if ((p.flags & F_OBJ) && (p.flags & F_FEVAL_Y)) dfid_at(rv(SPORCO_AMD_VAR_Y), out_dev, &p);
    }

    void admm_xstep(const sporco_amd_admm_params &p, double *out_dev) override {
        before_state_change();
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        xstep_impl(p, out_dev);
    }

    void admm_relax(double rlx) override {
        before_read(SPORCO_AMD_VAR_X);
        ProfScope ps(prof, PS_OTHER);
        launch_relax<T>(st, rv(SPORCO_AMD_VAR_X), rv(SPORCO_AMD_VAR_Y), rv(SPORCO_AMD_VAR_AX), (T)rlx,
                        E);
    }

    void admm_ystep(const sporco_amd_admm_params &p) override {
        ProfScope ps(prof, PS_OTHER);
        launch_ystep<T>(st, rv(SPORCO_AMD_VAR_AX), rv(SPORCO_AMD_VAR_U), rv(SPORCO_AMD_VAR_Y),
                        (T)(p.lmbda / p.rho), (T)(p.mu / p.rho), (T)p.u_scale, p.flags, d5(), p.dH,
                        p.dW, wl1, wl21, ams_of(p), ams_k0(), ams_n());
    }

    void admm_ustep(const sporco_amd_admm_params &p) override {
        ProfScope ps(prof, PS_OTHER);
        launch_ustep<T>(st, rv(SPORCO_AMD_VAR_AX), rv(SPORCO_AMD_VAR_Y), rv(SPORCO_AMD_VAR_U),
                        (T)p.u_scale, E);
    }

    void admm_stats(const sporco_amd_admm_params &p, double *out_dev) override {
        before_read(SPORCO_AMD_VAR_X);
        // keeps the xstep sums already in out_dev; fills the residual/regulariser slots
        int nb;
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_admm_stats<T>(st, rv(SPORCO_AMD_VAR_X), rv(SPORCO_AMD_VAR_Y),
                                      rv(SPORCO_AMD_VAR_YPREV), rv(SPORCO_AMD_VAR_U), p.flags, d5(),
                                      wl1, wl21, (p.flags & F_AMS) ? ams_k0() : -1, part_b,
                                      ams_n());
        }
        const int slots[7] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_S2, SPORCO_AMD_OUT_AX2,
                              SPORCO_AMD_OUT_Y2, SPORCO_AMD_OUT_U2, SPORCO_AMD_OUT_L1,
                              SPORCO_AMD_OUT_L21};
        const double scales[7] = {1, 1, 1, 1, 1, 1, 1};
        finalize(part_b, nb, 8, 7, slots, scales, out_dev);
        if ((p.flags & F_OBJ) && (p.flags & F_FEVAL_Y)) dfid_at(rv(SPORCO_AMD_VAR_Y), out_dev, &p);
    }

    void scale_u(double s) override {
        ProfScope ps(prof, PS_OTHER);
        launch_scale<T>(st, rv(SPORCO_AMD_VAR_U), (T)s, E);
    }

    void reconstruct(int var, void *dst) override {
        require_ready();
        SA_REQUIRE(!var_is_complex(var), "reconstruct needs a real state variable");
        before_read(var);
        cx<T> *wk = work_buf();
        fwd2(rv(var), nullptr, T(0), wk, P);
        {
            ProfScope ps(prof, PS_OTHER);
            inner_df(wk);
        }
        inv2(innerb, innerb, sreal, CNs);
        SA_HIP(hipMemcpyAsync(dst, sreal, sizeof(T) * (int64_t)H * W * CNs, hipMemcpyDeviceToHost, st));
        sync();
    }
    void reconstruct_dev(int var, void *dst_dev) override {
        require_ready();
        SA_REQUIRE(!var_is_complex(var), "reconstruct needs a real state variable");
        before_read(var);
        cx<T> *wk = work_buf();
        fwd2(rv(var), nullptr, T(0), wk, P);
        {
            ProfScope ps(prof, PS_OTHER);
            inner_df(wk);
        }
        inv2(innerb, innerb, sreal, CNs);
        SA_HIP(hipMemcpyAsync(dst_dev, sreal, sizeof(T) * (int64_t)H * W * CNs,
                              hipMemcpyDeviceToDevice, st));
        sync();
    }

    void dhs_absmax(double *out_host) override {
        require_ready();
        int nb;
        {
            ProfScope ps(prof, PS_OTHER);
            nb = Cd > 1 ? launch_mc_dhs_absmax<T>(st, cv(SPORCO_AMD_VAR_DF), cv(SPORCO_AMD_VAR_SF),
                                                   npix, Cd, N, K, part_a)
                        : launch_dhs_absmax<T>(st, cv(SPORCO_AMD_VAR_DF), cv(SPORCO_AMD_VAR_SF), npix,
                                               CN, K, part_a);
        }
        const int slots[1] = {0};
        const double scales[1] = {1.0};
        SA_HIP(hipMemsetAsync(out_dev_own, 0, sizeof(double) * kOutSlots, st));
        finalize(part_a, nb, 1, 1, slots, scales, out_dev_own, true);
        double tmp[kOutSlots];
        read_out(out_dev_own, tmp);
        *out_host = std::sqrt(tmp[0]);
    }

    // ---- PGM -----------------------------------------------------------------------
    // rows pass of the fused iteration: X = prox(irfft_W(t_in)), t_out = rfft_W(X)
    void pgm_rows_prox(const sporco_amd_pgm_params &p, const cx<T> *t_in, cx<T> *t_out, T *x,
                       double *out_dev) {
        RowsProxArgs<T> ra;
        ra.t_in = t_in;
        ra.t_out = t_out;
        ra.x = x;
        ra.twA = twRows;
        ra.twW = planW.tw<T>();
        ra.scale = T(1.0 / ((double)H * (double)W));
        ra.thr = (T)(p.lmbda / p.L);
        ra.flags = p.flags;
        ra.H = H;
        ra.W = W;
        ra.C = C;
        ra.N = N;
        ra.K = K;
        ra.dH = p.dH;
        ra.dW = p.dW;
        ra.P = P;
        ra.wl1 = wl1;
        ra.partials = part_rows;
        int64_t nt;
        {
            ProfScope ps(prof, PS_PGM_ROWS_PROX);
            nt = launch_rows_inv_prox_fwd<T>(st, ra);
        }
        if (out_dev) {
            const int slots[1] = {SPORCO_AMD_PGM_L1};
            const double scales[1] = {1.0};
            finalize(part_rows, (int)nt, 1, 1, slots, scales, out_dev);
        }
    }

    void pgm_iter(const sporco_amd_pgm_params &p, double *out_dev) override {
        require_single_channel_dict();
        require_ready();
        if (!pgm_fused_ok())
            throw Error(SPORCO_AMD_EINVAL, "pgm_iter: shape not served by the fused kernels");
        t_ready = false;
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        const int NHp = (K + 63) / 64;      // 64-filter slabs of the column kernels (1: K <= 64)
        if (!part_pgm) {
            SA_HIP(hipMalloc((void **)&part_pgm,
                             sizeof(double) * kPgmPartialStride * (int64_t)Wf * CN * NHp));
            if (NHp > 1) SA_HIP(hipMalloc((void **)&part_pgm2, sizeof(double) * 3 * (int64_t)Wf * CN));
        }
        if (p.hold && !pgm_ey) SA_HIP(hipMalloc((void **)&pgm_ey, sizeof(cx<T>) * (int64_t)Wf * CN * H));
        if (!pgm_tiled) {
            // enter the tile-major regime: the two live iterates are re-laid out once
            need_natural(SPORCO_AMD_VAR_XF);   // (an ADMM leftover in the Xf buffer is resolved first)
            need_natural(SPORCO_AMD_VAR_YF);
            relayout(SPORCO_AMD_VAR_XF, true);
            relayout(SPORCO_AMD_VAR_YF, true);
            (void)cv(SPORCO_AMD_VAR_XFPRV);
            (void)cv(SPORCO_AMD_VAR_YFPRV);
            (void)cv(SPORCO_AMD_VAR_VF);
            pgm_tiled = true;
        }
        if (pgm_x_stale) pgm_x_stale = false;   // X of the previous iteration is superseded
        cx<T> *Xf = cv(SPORCO_AMD_VAR_XF), *Yf = cv(SPORCO_AMD_VAR_YF);
        cx<T> *Xprv = cv(SPORCO_AMD_VAR_XFPRV), *Yprv = cv(SPORCO_AMD_VAR_YFPRV);
        cx<T> *spare = cv(SPORCO_AMD_VAR_VF), *Tm = work_buf();
        PgmColsArgs<T> ca;
        ca.dft = dft;
        ca.sft = sft;
        ca.twA = twA;
        ca.twB = twB;
        ca.inv_L = (T)(1.0 / p.L);
        ca.beta = (T)p.beta;
        ca.H = H;
        ca.W = W;
        ca.CN = CN;
        ca.K = K;
        ca.want_stats = p.want_stats || p.hold;
        ca.ey = p.hold ? pgm_ey : nullptr;
        // 1. gradient step at Yf, inverse transform along H
        ca.yf = Yf;
        ca.xf_old = nullptr;
        ca.t = Tm;
        ca.yf_new = nullptr;
        ca.partials = part_f;
        int64_t ntile;
        if (NHp > 1) {
            // K > 64: cooperating slab workgroups (csc_fused.h launch_pgm_grad_slabs)
            FusedSlabArgs<T> sa;
            sa.c.t = Tm;
            sa.c.dft = dft;
            sa.c.sft = sft;
            sa.c.twA = twA;
            sa.c.twB = twB;
            sa.c.H = H;
            sa.c.W = W;
            sa.c.CN = CN;
            sa.c.K = K;
            sa.c.partials = part_f;
            sa.qpart = qpart;
            sa.pgm_yf = Yf;
            sa.pgm_inv_L = ca.inv_L;
            sa.pgm_ey = ca.ey;
            coop_prepare(sa);
            ProfScope ps(prof, PS_PGM_GRAD_IFFT);
            ntile = launch_pgm_grad_slabs<T>(st, sa);
        } else {
            ProfScope ps(prof, PS_PGM_GRAD_IFFT);
            ntile = launch_pgm_grad_ifft<T>(st, ca);
        }
        {
            const int slots[1] = {SPORCO_AMD_PGM_FY};
            const double scales[1] = {0.5};
            finalize(part_f, (int)ntile, 1, 1, slots, scales, out_dev);
        }
        // 2. inverse along W, proximal map, forward along W
        pgm_rows_prox(p, Tm, spare, nullptr, out_dev);
        // 3. forward along H (in place: `spare` becomes the new Xf), momentum into the old
        //    Yfprv buffer, residual and objective sums
        ca.yf = Yf;
        ca.xf_old = Xf;
        ca.t = spare;
        ca.yf_new = Yprv;
        ca.partials = part_pgm;
        ca.qpart = (NHp > 1 && ca.want_stats) ? qpart : nullptr;
        int64_t nrows;
        {
            ProfScope ps(prof, PS_PGM_FFT_MOM);
            nrows = launch_pgm_fft_momentum<T>(st, ca);
        }
        if (NHp > 1) {
            // per (tile, slab): the residual sums; per tile, from the slabs' shares: the objective
            const int s0[1] = {SPORCO_AMD_PGM_RSDL}, s4[1] = {SPORCO_AMD_PGM_DXY2};
            const double c0[1] = {1.0 / ((double)H * W)}, c4[1] = {1.0};
            finalize(part_pgm, (int)nrows, kPgmPartialStride, 1, s0, c0, out_dev);
            if (p.hold) finalize(part_pgm + 4, (int)nrows, kPgmPartialStride, 1, s4, c4, out_dev);
            if (ca.want_stats) {
                {
                    ProfScope ps(prof, PS_PGM_FFT_MOM);
                    launch_pgm_stats_slabs<T>(st, ca, part_pgm2);
                }
                const int s2[3] = {SPORCO_AMD_PGM_DFID, SPORCO_AMD_PGM_F, SPORCO_AMD_PGM_LIN};
                const double c2[3] = {1.0 / ((double)H * W), 0.5, 1.0};
                finalize(part_pgm2, (int)ntile, 3, p.hold ? 3 : 2, s2, c2, out_dev);
            }
        } else {
            const int slots[5] = {SPORCO_AMD_PGM_RSDL, SPORCO_AMD_PGM_DFID, SPORCO_AMD_PGM_F,
                                  SPORCO_AMD_PGM_LIN, SPORCO_AMD_PGM_DXY2};
            const double scales[5] = {1.0 / ((double)H * W), 1.0 / ((double)H * W), 0.5, 1.0, 1.0};
            finalize(part_pgm, (int)ntile, kPgmPartialStride, p.hold ? 5 : ca.want_stats ? 3 : 1, slots,
                     scales, out_dev);
        }
        last_pgm = p;
        pgm_held = true;
        if (!p.hold) pgm_commit();
    }

    // on_iteration_start's copies as a rotation of buffers (pgm.py:835-846): the trial in the
    // spare buffers becomes the state
    void pgm_commit() override {
        SA_REQUIRE(pgm_held, "pgm_commit without a held pgm_iter");
        pgm_held = false;
        cx<T> *Xf = cv(SPORCO_AMD_VAR_XF), *Yf = cv(SPORCO_AMD_VAR_YF);
        cx<T> *Xprv = cv(SPORCO_AMD_VAR_XFPRV), *Yprv = cv(SPORCO_AMD_VAR_YFPRV);
        cx<T> *spare = cv(SPORCO_AMD_VAR_VF);
        vars[SPORCO_AMD_VAR_XF] = spare;
        vars[SPORCO_AMD_VAR_XFPRV] = Xf;
        vars[SPORCO_AMD_VAR_VF] = Xprv;
        vars[SPORCO_AMD_VAR_YF] = Yprv;
        vars[SPORCO_AMD_VAR_YFPRV] = Yf;
        xf_tiled = false;
        x_stale = false;
        x_invalid = false;
        pgm_x_stale = true;
    }

    void pgm_grad(int var, double *out_dev) override {
        require_ready();
        SA_REQUIRE(var_is_complex(var), "pgm_grad needs a frequency-domain variable");
        before_read(var);
        int nb;
        {
            ProfScope ps(prof, PS_PGM);
            nb = Cd > 1 ? launch_mc_pgm_grad<T>(st, cv(var), cv(SPORCO_AMD_VAR_DF),
                                                cv(SPORCO_AMD_VAR_SF), cv(SPORCO_AMD_VAR_GF), npix,
                                                Cd, N, K, W, part_a)
                        : launch_pgm_grad<T>(st, cv(var), cv(SPORCO_AMD_VAR_DF),
                                             cv(SPORCO_AMD_VAR_SF), cv(SPORCO_AMD_VAR_GF), npix, CN,
                                             K, W, part_a);
        }
        const int slots[2] = {SPORCO_AMD_PGM_F, SPORCO_AMD_PGM_DFID};
        const double scales[2] = {0.5, 1.0 / ((double)H * W)};
        finalize(part_a, nb, 2, 2, slots, scales, out_dev);
    }

    void pgm_eval(int var, double *out_dev) override {
        require_ready();
        SA_REQUIRE(var_is_complex(var), "pgm_eval needs a frequency-domain variable");
        before_read(var);
        int nb;
        {
            ProfScope ps(prof, PS_PGM);
            inner_df(cv(var));
            nb = launch_pair_stats<T>(st, innerb, cv(SPORCO_AMD_VAR_SF), nullptr, npix, CNs, W, part_a);
        }
        // partial layout per block: [0] weighted |d|^2, [1] Re<d,g>, [2] |d|^2, [3] |g|^2
        const int s0[1] = {SPORCO_AMD_PGM_DFID};
        const double c0[1] = {1.0 / ((double)H * W)};
        finalize(part_a, nb, 4, 1, s0, c0, out_dev);
        const int s2[1] = {SPORCO_AMD_PGM_F};
        const double c2[1] = {0.5};
        finalize(part_a + 2, nb, 4, 1, s2, c2, out_dev);
        {
            ProfScope ps(prof, PS_PGM);
            nb = launch_pair_stats<T>(st, innerb, nullptr, nullptr, npix, CNs, W, part_b);
        }
        const int s3[1] = {SPORCO_AMD_PGM_HESS};
        const double c3[1] = {1.0};
        finalize(part_b + 2, nb, 4, 1, s3, c3, out_dev);
    }

    void pgm_prox_step(double L, double lmbda, uint32_t flags, int dH, int dW,
                       double *out_dev) override {
        require_ready();
        pgm_leave_tiled();
        x_written();
        cx<T> *Vf = cv(SPORCO_AMD_VAR_VF);
        {
            ProfScope ps(prof, PS_PGM);
            launch_axpy_c<T>(st, cv(SPORCO_AMD_VAR_YF), cv(SPORCO_AMD_VAR_GF), Vf, (T)(-1.0 / L), EF);
        }
        T *X = rv(SPORCO_AMD_VAR_X);
        inv2(Vf, work_buf(), X, P);
        int nb;
        {
            ProfScope ps(prof, PS_PGM);
            nb = launch_prox_l1<T>(st, X, X, (T)(lmbda / L), flags, d5(), dH, dW, wl1, part_b);
        }
        const int slots[1] = {SPORCO_AMD_PGM_L1};
        const double scales[1] = {1.0};
        finalize(part_b, nb, 1, 1, slots, scales, out_dev);
        xf_tiled = false;
        fwd2(X, nullptr, T(0), cv(SPORCO_AMD_VAR_XF), P);
    }

    void lincomb(int dst, double a, int va, double b, int vb, double c, int vc) override {
        SA_REQUIRE(var_is_valid(dst) && var_is_complex(dst) && dst != SPORCO_AMD_VAR_SF,
                   "lincomb works on frequency-domain state variables");
        for (int v : {va, vb, vc})
            SA_REQUIRE(v < 0 || (var_is_complex(v) && var_bytes(v) == var_bytes(dst)),
                       "lincomb operand of the wrong kind");
        SA_REQUIRE(va >= 0, "lincomb needs a first operand");
        if (is_pgm_iterate(dst)) pgm_leave_tiled();
        for (int v : {va, vb, vc})
            if (v >= 0) before_read(v);
        if (dst == SPORCO_AMD_VAR_XF) xf_tiled = false;
        ProfScope ps(prof, PS_PGM);
        launch_lincomb<T>(st, cv(dst), (T)a, cv(va), (T)b, vb >= 0 ? cv(vb) : nullptr, (T)c,
                          vc >= 0 ? cv(vc) : nullptr, (int64_t)(var_bytes(dst) / sizeof(cx<T>)));
    }

    void pair_stats(int va, int vb, int vg, double *out_dev) override {
        SA_REQUIRE(va >= 0 && var_is_valid(va) && var_is_complex(va) && va != SPORCO_AMD_VAR_SF,
                   "pair_stats needs a frequency-domain first operand");
        for (int v : {vb, vg})
            SA_REQUIRE(v < 0 || (var_is_complex(v) && var_bytes(v) == var_bytes(va)),
                       "pair_stats operands must have the same shape");
        const int64_t cols = var_is_dict_sized(va) ? KD() : P;
        for (int v : {va, vb, vg})
            if (v >= 0) before_read(v);
        int nb;
        {
            ProfScope ps(prof, PS_PGM);
            nb = launch_pair_stats<T>(st, cv(va), vb >= 0 ? cv(vb) : nullptr,
                                      vg >= 0 ? cv(vg) : nullptr, npix, cols, W, part_a);
        }
        const int slots[4] = {0, 1, 2, 3};
        const double scales[4] = {1.0 / ((double)H * W), 1.0, 1.0, 1.0};
        finalize(part_a, nb, 4, 4, slots, scales, out_dev);
    }

    void fft_var(int rvar, int cvar, bool inverse) override {
        SA_REQUIRE(var_is_valid(rvar) && var_is_valid(cvar) && !var_is_complex(rvar) &&
                       var_is_complex(cvar) && cvar != SPORCO_AMD_VAR_SF &&
                       var_is_dict_sized(rvar) == var_is_dict_sized(cvar),
                   "fft_var needs a real and a complex variable of matching shape");
        const int64_t cols = var_is_dict_sized(rvar) ? KD() : P;
        if (is_pgm_iterate(cvar)) pgm_leave_tiled();
        if (inverse) {
            before_read(cvar);
            if (rvar == SPORCO_AMD_VAR_X) x_written();
        } else {
            before_read(rvar);
            if (cvar == SPORCO_AMD_VAR_XF) xf_tiled = false;
        }
        if (inverse)
            inv2(cv(cvar), var_is_dict_sized(rvar) ? dwork_buf() : work_buf(), rv(rvar), cols);
        else
            fwd2(rv(rvar), nullptr, T(0), cv(cvar), cols);
    }

    // ---- dictionary update -------------------------------------------------------------
    void ccmod_setcoef(int var) override {
        SA_REQUIRE(var_is_valid(var) && !var_is_complex(var) && !var_is_dict_sized(var),
                   "ccmod_setcoef needs an X-sized real variable");
        before_read(var);
        gramz_valid = false;
        zsf_valid = dism_valid = false;
        // A multi-channel dictionary whose coefficient maps carry the channels as well (the
        // reference's broadcasting admits it, tests/admm/test_ccmod.py:278-295: Cd independent
        // single-channel updates sharing rho and the residuals), staged in the consensus blocks'
        // layout (H, W, N, Cd, K) -- VAR_CX -- and kept in a spectrum of that size
        z_chan = Cd > 1 && var == SPORCO_AMD_VAR_CX;
        if (z_chan) {
            if (!zf_ch) SA_HIP(hipMalloc((void **)&zf_ch, sizeof(cx<T>) * EF * Cd));
            zf_tiled = false;
            fwd2(rv(var), nullptr, T(0), zf_ch, P * Cd);
            SA_HIP(hipMemsetAsync(rv(var), 0, sizeof(T) * E * Cd, st));   // (X_n = 0 before a solve)
            return;
        }
        // (the generic consensus D-step and the single-copy ADMM D-step read Zf in the natural
        // layout)
        // (K > 64: the slab forms of the column transform and of the PGM gradient; the ADMM
        // dictionary updates read Zf in the natural layout there)
        if (rows_ok && cols256 && (fused || (fused_slabs && !cns_active)) &&
            !(cns_active && !cns_fused()) && !eq_active) {
            // rows then columns, register-resident, straight into the tile-major layout
            RowsFwdArgs<T> ra;
            // (the iterate in its single-array form: Y = prox(V) is derived inside the row pass
            // -- with s2 = 0 the kernel transforms Y - 0 U = Y -- instead of being written out first)
            const bool from_v = var == SPORCO_AMD_VAR_Y && v_live;
            if (from_v) {
                ra.y = nullptr;
                ra.v = v_cur;
                ra.thr_prev = v_thr;
                ra.thr21_prev = v_thr21;
                ra.flags = (v_nonneg ? F_NONNEG : 0u) | (v_joint ? F_JOINT : 0u) | v_opts;
                ra.C = C;
                ra.N = N;
                ra.wl1 = wl1;
                ra.dH = v_dH;
                ra.dW = v_dW;
                if (v_opts & F_AMS) {
                    sporco_amd_admm_params q = last_p;
                    q.flags |= F_AMS;
                    ra.ams_bits = ams_bits_of(q);
                    ra.ams_k = Ku - 1;
                }
            } else {
                ra.y = rv(var);
            }
            ra.u = nullptr;
            ra.s2 = T(0);
            ra.t = cv(SPORCO_AMD_VAR_ZF);
            ra.twA = twRows;
            ra.H = H;
            ra.W = W;
            ra.CN = CN;
            ra.K = K;
            ra.P = P;
            {
                ProfScope ps(prof, PS_ROWS_FWD);
                launch_rows_fwd<T>(st, ra);
            }
            PgmColsArgs<T> ca;
            ca.yf = nullptr;
            ca.xf_old = nullptr;
            ca.t = cv(SPORCO_AMD_VAR_ZF);
            ca.yf_new = nullptr;
            ca.ey = nullptr;
            ca.dft = nullptr;
            ca.sft = nullptr;
            ca.twA = twA;
            ca.twB = twB;
            ca.inv_L = T(0);
            ca.beta = T(0);
            ca.H = H;
            ca.W = W;
            ca.CN = CN;
            ca.K = K;
            ca.want_stats = 0;
            ca.partials = nullptr;
            {
                ProfScope ps(prof, PS_PGM_FFT_MOM);
                launch_cols_fft<T>(st, ca);
            }
            zf_tiled = true;
            return;
        }
        zf_tiled = false;
        fwd2(rv(var), nullptr, T(0), cv(SPORCO_AMD_VAR_ZF), P);
    }

    void ccmod_grad(int var, bool write_grad, double *out_dev) override {
        if (!have_signal) throw Error(SPORCO_AMD_ESTATE, "set_signal must be called first");
        SA_REQUIRE(var_is_valid(var) && var_is_complex(var) && var_is_dict_sized(var),
                   "ccmod_grad needs a dictionary-sized frequency-domain variable");
        if (zf_tiled) {
            // fixed groups of tiles per workgroup; enough groups to fill the chip
            if (!gpart) {
                ccmod_groups = (int)ceil_div(768, Wf);
                if (ccmod_groups > CN) ccmod_groups = CN;
                if (ccmod_groups > 8) ccmod_groups = 8;
                SA_HIP(hipMalloc((void **)&gpart, sizeof(cx<T>) * npix * K * ccmod_groups));
            }
            CcmodTiledArgs<T> ga;
            ga.zf = cv(SPORCO_AMD_VAR_ZF);
            ga.d = cv(var);
            ga.sft = sft;
            ga.gpart = write_grad ? gpart : nullptr;
            ga.H = H;
            ga.W = W;
            ga.CN = CN;
            ga.K = K;
            ga.G = ccmod_groups;
            ga.partials = part_a;
            if (K > 64) {
                // (one row of sums per tile: part_f holds 2 * slabs doubles per tile)
                if (!ccmod_r) SA_HIP(hipMalloc((void **)&ccmod_r, sizeof(cx<T>) * (int64_t)Wf * CN * H));
                ga.qpart = qpart;
                ga.rbuf = ccmod_r;
                ga.partials = part_f;
            }
            int64_t nwg;
            {
                ProfScope ps(prof, PS_PGM);
                nwg = launch_ccmod_grad_tiled<T>(st, ga);
                if (write_grad)
                    launch_sum_groups<T>(st, gpart, cv(SPORCO_AMD_VAR_DGF), npix * K, ccmod_groups);
            }
            const int slots[3] = {SPORCO_AMD_PGM_F, SPORCO_AMD_PGM_DFID, SPORCO_AMD_PGM_HESS};
            const double scales[3] = {0.5, 1.0 / ((double)H * W), 1.0};
            finalize(ga.partials, (int)nwg, 4, 3, slots, scales, out_dev);
            return;
        }
        int nb;
        {
            ProfScope ps(prof, PS_PGM);
            nb = launch_ccmod_grad<T>(st, zf_nat(), cv(var), cv(SPORCO_AMD_VAR_SF),
                                      write_grad ? cv(SPORCO_AMD_VAR_DGF) : nullptr, npix, CN, K, W,
                                      part_a, Cd, z_chan);
        }
        const int slots[3] = {SPORCO_AMD_PGM_F, SPORCO_AMD_PGM_DFID, SPORCO_AMD_PGM_HESS};
        const double scales[3] = {0.5, 1.0 / ((double)H * W), 1.0};
        finalize(part_a, nb, 3, 3, slots, scales, out_dev);
    }

    void pcn_project(const T *v, T *out, int dH, int dW, bool zm, double *out_dev) {
        SA_REQUIRE(dH >= 1 && dW >= 1 && dH <= H && dW <= W, "filter support out of range");
        int nb;
        {
            ProfScope ps(prof, PS_OTHER);
            launch_pcn_stats<T>(st, v, pcn_stats_buf(), H, W, K, dH, dW, zm, Cd, fsz());
            nb = launch_pcn_apply<T>(st, v, pcn_stats_buf(), out, H, W, K, dH, dW, part_b, Ku, Cd, fsz());
        }
        if (out_dev) {
            const int slots[1] = {0};
            const double scales[1] = {1.0};
            finalize(part_b, nb, 1, 1, slots, scales, out_dev);
        }
    }

    void ccmod_prox_step(double L, int dH, int dW, bool zm) override {
        cx<T> *Vf = cv(SPORCO_AMD_VAR_DVF);
        {
            ProfScope ps(prof, PS_PGM);
            launch_axpy_c<T>(st, cv(SPORCO_AMD_VAR_DYF), cv(SPORCO_AMD_VAR_DGF), Vf, (T)(-1.0 / L),
                             npix * KD());
        }
        T *X = rv(SPORCO_AMD_VAR_DX);
        inv2(Vf, dwork_buf(), X, KD());
        pcn_project(X, X, dH, dW, zm, nullptr);
        fwd2(X, nullptr, T(0), cv(SPORCO_AMD_VAR_DXF), KD());
    }

    // One projected stochastic gradient step on the X-step's dictionary (onlinecdl.py:310-333):
    // G = irfftn(Df - eta * gradient), D = Pcn(G); out[CNSTR] = sum (Pcn(G) - G)^2 (:398).
    void ccmod_sgd_step(double eta, int dH, int dW, bool zm, double *out_dev) override {
        if (!have_dict) throw Error(SPORCO_AMD_ESTATE, "set_dict must be called first");
        SA_REQUIRE(dH >= 1 && dW >= 1 && dH <= H && dW <= W, "filter support out of range");
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        cx<T> *Vf = cv(SPORCO_AMD_VAR_DVF);
        {
            ProfScope ps(prof, PS_PGM);
            launch_axpy_c<T>(st, cv(SPORCO_AMD_VAR_DF), cv(SPORCO_AMD_VAR_DGF), Vf, (T)(-eta),
                             npix * KD());
        }
        T *X = rv(SPORCO_AMD_VAR_DX);
        inv2(Vf, dwork_buf(), X, KD());
        int nb;
        {
            ProfScope ps(prof, PS_OTHER);
            launch_pcn_stats<T>(st, X, pcn_stats_buf(), H, W, K, dH, dW, zm, Cd, fsz());
            nb = launch_pcn_apply<T>(st, X, pcn_stats_buf(), X, H, W, K, dH, dW, part_b, Ku, Cd, fsz());
        }
        const int slots[1] = {SPORCO_AMD_OUT_CNSTR};
        const double scales[1] = {1.0};
        finalize(part_b, nb, 1, 1, slots, scales, out_dev);
        fwd2(X, nullptr, T(0), cv(SPORCO_AMD_VAR_DXF), KD());
    }

    void ccmod_cnstr(int dH, int dW, bool zm, double *out_dev) override {
        pcn_project(rv(SPORCO_AMD_VAR_DX), nullptr, dH, dW, zm, out_dev);
    }

    void ccmod_getdict(int dH, int dW, void *dst) override {
        SA_REQUIRE(dH >= 1 && dW >= 1 && dH <= H && dW <= W, "filter support out of range");
        std::vector<T> tmp;
        void *out = dst;
        if (K != Ku) {
            tmp.resize((size_t)dH * dW * K);
            out = tmp.data();
        }
        // (rows of dW pixels x Cd channels x K filters; Cd > 1 is never padded)
        SA_HIP(hipMemcpy2DAsync(out, sizeof(T) * (size_t)dW * KD(), rv(SPORCO_AMD_VAR_DX),
                                sizeof(T) * (size_t)W * KD(), sizeof(T) * (size_t)dW * KD(),
                                (size_t)dH, hipMemcpyDeviceToHost, st));
        sync();
        if (K != Ku) {   // drop the padding filter
            T *o = static_cast<T *>(dst);
            for (int64_t r = 0; r < (int64_t)dH * dW; ++r)
                for (int k = 0; k < Ku; ++k) o[r * Ku + k] = tmp[(size_t)(r * K + k)];
        }
    }

    void setdict_from_dstep(int dH, int dW) override {
        before_dict_change();
        SA_HIP(hipMemcpyAsync(cv(SPORCO_AMD_VAR_DF), cv(SPORCO_AMD_VAR_DXF),
                              sizeof(cx<T>) * npix * KD(), hipMemcpyDeviceToDevice, st));
        ism_valid = false;
        if (Cd == 1) {
            ProfScope ps(prof, PS_OTHER);
            launch_gram<T>(st, cv(SPORCO_AMD_VAR_DF), gram, npix, K);
        }
        refresh_fused_dict();
        dH_ = dH;
        dW_ = dW;
        have_dict = true;
    }

    // ---- masked data fidelity (pgm ConvBPDNMask / ConvCnstrMODMask) -------------------------
    // Gradient of (1/2) ||W (sum_m d_m * x_m - s)||^2 with respect to the coefficient spectra
    // (dstep false: `var` is X-sized, result in VAR_GF; pgm/cbpdn.py:454-477) or to the
    // dictionary (dstep true: `var` is dictionary sized, result in VAR_DGF; pgm/ccmod.py:552-575):
    // residual -> irfftn -> W^2 -> rfftn -> adjoint.  With write_grad false the residual is
    // weighted by W only and nothing is written back: out[PGM_DFID] = sum (W R)^2 (twice the
    // data fidelity term, :481-489) and out[PGM_F] = (1/2) sum_half |rfftn(W R)|^2 (the
    // unnormalised DFT-domain value backtracking compares, :493-506).
    // (mode 0: evaluation, 1: gradient with W^2, 2: gradient with the residual weighted by W
    // once -- the form of the online learner's dictionary step, onlinecdl.py:578-580)
    void masked_grad(int var, bool dstep, int mode, double *out_dev) override {
        const bool write_grad = mode != 0;
        // (multi-channel dictionary, Cd = Cs > 1: the coefficient maps have no channel axis, the
        // residual has the signal's -- inner products and adjoints over (channel, filter))
        if (!have_signal) throw Error(SPORCO_AMD_ESTATE, "set_signal must be called first");
        SA_REQUIRE(var_is_valid(var) && var_is_complex(var) && var_is_dict_sized(var) == dstep &&
                       var != SPORCO_AMD_VAR_SF && var != SPORCO_AMD_VAR_DF,
                   "masked_grad: variable of the wrong kind");
        before_read(var);
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        cx<T> *Sf = cv(SPORCO_AMD_VAR_SF);
        {   // residual spectrum R = sum_m Df Xf - Sf, signal sized (npix, C N)
            ProfScope ps(prof, PS_PGM);
            if (dstep) {
                need_natural(SPORCO_AMD_VAR_ZF);
                if (Cd > 1) launch_mc_inner<T>(st, cv(var), zf_nat(), innerb, npix, Cd, N, K, z_chan);
                else launch_inner<T>(st, cv(var), cv(SPORCO_AMD_VAR_ZF), innerb, npix, CN, K);
            } else {
                require_ready();
                if (Cd > 1) launch_mc_inner<T>(st, cv(SPORCO_AMD_VAR_DF), cv(var), innerb, npix, Cd, N, K);
                else launch_inner<T>(st, cv(SPORCO_AMD_VAR_DF), cv(var), innerb, npix, CN, K);
            }
            launch_lincomb<T>(st, innerb, T(1), innerb, T(-1), Sf, T(0), nullptr, npix * CNs);
        }
        inv2(innerb, innerb, sreal, CNs);
        int nb;
        {
            ProfScope ps(prof, PS_PGM);
            nb = launch_mask_apply<T>(st, sreal, have_wdat ? wdat : Weight<T>(), mode == 1, H, W, Cs, N,
                                      part_a);
        }
        {
            const int slots[1] = {SPORCO_AMD_PGM_DFID};
            const double scales[1] = {1.0};
            finalize(part_a, nb, 1, 1, slots, scales, out_dev);
        }
        fwd2(sreal, nullptr, T(0), innerb, CNs);
        if (!write_grad) {
            {
                ProfScope ps(prof, PS_PGM);
                nb = launch_pair_stats<T>(st, innerb, nullptr, nullptr, npix, CNs, W, part_b);
            }
            const int slots[1] = {SPORCO_AMD_PGM_F};
            const double scales[1] = {0.5};
            finalize(part_b + 2, nb, 4, 1, slots, scales, out_dev);
            return;
        }
        ProfScope ps(prof, PS_PGM);
        if (dstep) {
            if (Cd > 1)
                launch_mc_zf_adjoint<T>(st, zf_nat(), innerb, cv(SPORCO_AMD_VAR_DGF), npix, Cd, N, K, z_chan);
            else
                launch_zf_adjoint<T>(st, cv(SPORCO_AMD_VAR_ZF), innerb, cv(SPORCO_AMD_VAR_DGF), npix, CN, K);
        } else if (Cd > 1) {
            launch_mc_conj_outer<T>(st, cv(SPORCO_AMD_VAR_DF), innerb, cv(SPORCO_AMD_VAR_GF), npix, Cd, N, K,
                                    false);
        } else {
            launch_conj_outer<T>(st, cv(SPORCO_AMD_VAR_DF), innerb, cv(SPORCO_AMD_VAR_GF), npix, CN, K);
        }
    }

    // ---- ADMM consensus dictionary update -------------------------------------------------
    // Multi-channel dictionary (Cd > 1; admm/ccmod.py:696-698, :766-822): one dictionary copy
    // (H, W, Cd, K) per image -- blocks laid out (H, W, N, Cd, K), so that every kernel below
    // sees N blocks of KD() = Cd K "filters" -- whose Cd channels share the image's matrix in
    // the per-image solve (launch_sm_solve per_grp = Cd, the signal spectrum transposed to
    // (npix, N, Cd) to follow the systems).  Generic chain only.
    void cns_init(const void *Y0, double rho) override {
        SA_REQUIRE(rho != 0.0, "rho must be nonzero");
        cns_active = true;
        T *Y = rv(SPORCO_AMD_VAR_DX), *U = rv(SPORCO_AMD_VAR_CU);
        (void)rv(SPORCO_AMD_VAR_CX);
        SA_HIP(hipMemsetAsync(U, 0, sizeof(T) * E * Cd, st));
        if (Y0) {
            host_copy(SPORCO_AMD_VAR_DX, const_cast<void *>(Y0), true);
            // U_n = Y0 / rho for every image: 0 - (-1/rho) * Y through the Y - s U kernel
            ProfScope ps(prof, PS_OTHER);
            launch_cns_yu<T>(st, Y, U, U, T(0), (int64_t)H * W, CN, (int)KD());
            launch_scale<T>(st, U, (T)(1.0 / rho), E * Cd);
        } else {
            SA_HIP(hipMemsetAsync(Y, 0, var_bytes(SPORCO_AMD_VAR_DX), st));
        }
        fwd2(Y, nullptr, T(0), cv(SPORCO_AMD_VAR_DXF), KD());
        sync();
    }

    cx<T> *cns_w = nullptr, *cns_sft = nullptr;   // Cd > 1: column-pass scratch, transposed Sf
    cx<T> *zf_ch = nullptr;     // Cd > 1, channel-ful coefficient maps: spectrum (npix, N, Cd, K)
    bool z_chan = false;
    const cx<T> *zf_nat() { return z_chan ? zf_ch : cv(SPORCO_AMD_VAR_ZF); }
    // multi-scale dictionary: per-filter support sizes of the constraint projection (K ints
    // each; null: the one support the calls name)
    int *flt_h = nullptr, *flt_w = nullptr;
    FilterSizes fsz() const {
        FilterSizes f;
        f.h = flt_h;
        f.w = flt_w;
        return f;
    }
    void set_filter_sizes(const int32_t *fh, const int32_t *fw) override {
        sync();
        if (flt_h) {
            SA_HIP(hipFree(flt_h));
            SA_HIP(hipFree(flt_w));
            flt_h = flt_w = nullptr;
        }
        if (!fh || !fw) return;
        std::vector<int> h((size_t)K, 0), w((size_t)K, 0);     // (padding filters: empty support)
        for (int k = 0; k < Ku; ++k) {
            SA_REQUIRE(fh[k] >= 1 && fh[k] <= H && fw[k] >= 1 && fw[k] <= W, "filter size out of range");
            h[k] = fh[k];
            w[k] = fw[k];
        }
        SA_HIP(hipMalloc((void **)&flt_h, sizeof(int) * K));
        SA_HIP(hipMalloc((void **)&flt_w, sizeof(int) * K));
        SA_HIP(hipMemcpy(flt_h, h.data(), sizeof(int) * K, hipMemcpyHostToDevice));
        SA_HIP(hipMemcpy(flt_w, w.data(), sizeof(int) * K, hipMemcpyHostToDevice));
    }
    void cns_buffers() {
        if (cns_f) return;
        const int64_t npixr = (int64_t)H * W;
        SA_HIP(hipMalloc((void **)&cns_f, sizeof(cx<T>) * EF * Cd));
        SA_HIP(hipMalloc((void **)&cns_m, sizeof(T) * npixr * KD()));
        SA_HIP(hipMalloc((void **)&cns_yold, sizeof(T) * npixr * KD()));
        if (Cd > 1) {
            SA_HIP(hipMalloc((void **)&cns_w, sizeof(cx<T>) * EF * Cd));
            SA_HIP(hipMalloc((void **)&cns_sft, sizeof(cx<T>) * npix * CNs));
        }
    }
    void *cns_mean_ptr(int64_t *count) override {
        cns_buffers();
        *count = (int64_t)H * W * KD();
        return cns_m;
    }

    // ---- consensus update with mask decoupling (ConvCnstrMODMaskDcpl_Consensus) ----------------
    void cns_md_init(const void *S) override {
        SA_REQUIRE(S != nullptr, "S is null");
        const size_t nb = sizeof(T) * (int64_t)H * W * CNs;
        if (!md_s) SA_HIP(hipMalloc((void **)&md_s, nb));
        SA_HIP(hipMemcpyAsync(md_s, S, nb, hipMemcpyHostToDevice, st));
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_DMY0), 0, nb, st));
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_DMU0), 0, nb, st));
        sync();
    }

    // One iteration of sporco/admm/ccmodmd.py:766-1083 in the order of ADMM.solve
    // (admm.py:331-367): xstep (:922-939, the consensus solve with rho = 1 and S + Y1 - U1 in
    // the signal's place), relax_AX (:899-918), ystep (:943-952), ustep (:956-963), and the
    // sums of compute_residuals (:976-1034) / obfn_dfd (:966-972).  Generic FFT chain.
    // Multi-channel dictionary (Cd > 1; the reference's examples/scripts/cdl/cbpdndl_md_clr.py):
    // one (Cd, K) block per image as in cns_iter -- systems and block spectra in (image, channel)
    // order, the signal-sized block (Y1, U1, S, the mask) in the signal's (channel, image) order,
    // its spectra transposed on the way in and out of the block arithmetic.
    void cns_md_iter(const sporco_amd_cns_params &p, double *out_dev) {
        SA_REQUIRE(md_s != nullptr, "cns_md_init must be called first");
        SA_REQUIRE(p.rho > 0.0, "rho must be positive");
        need_natural(SPORCO_AMD_VAR_ZF);
        const int64_t npixr = (int64_t)H * W, ns = npixr * CNs;
        const int KDi = (int)KD();
        const int64_t PD = P * Cd;
        const int64_t nblk = npix * CN * Cd;        // (frequency, image, channel) rows
        const T us = (T)p.u_scale;
        T *Y = rv(SPORCO_AMD_VAR_DX), *X = rv(SPORCO_AMD_VAR_CX), *U = rv(SPORCO_AMD_VAR_CU);
        T *Y1 = rv(SPORCO_AMD_VAR_DMY0), *U1 = rv(SPORCO_AMD_VAR_DMU0);
        const cx<T> *Zf = zf_nat();
        cns_buffers();
        cx<T> *wk = Cd > 1 ? cns_w : work_buf();
        // (signal-sized spectra in the blocks' order: the transposed copy when Cd > 1)
        auto to_blocks = [&](cx<T> *sig) -> cx<T> * {
            if (Cd == 1) return sig;
            ProfScope ps(prof, PS_OTHER);
            launch_swap_inner<T>(st, sig, cns_sft, npix, Cd, N);
            return cns_sft;
        };
        if (p.phase != 2) {
        // xstep: ZSf = conj(Zf) rfftn(S + Y1 - U1); X_n = irfftn(SM(Zf_n, 1, ZSf_n + rfftn(Y - U_n)))
        {
            ProfScope ps(prof, PS_OTHER);
            // (U1 takes no part in update_rho's `U /= rsf`, admm.py:573: the reference rescales
            // the consensus duals only, so the pending scale does not apply to it)
            launch_md_pre<T>(st, Y1, U1, md_s, sreal, T(1), ns);
        }
        fwd2(sreal, nullptr, T(0), innerb, CNs);
        const cx<T> *sfb = to_blocks(innerb);
        {
            ProfScope ps(prof, PS_FFT_R2C);
            fft_r2c<T>(st, planW, Y, U, us, cns_f, H, PD, (int64_t)W * PD, PD, (int64_t)Wf * PD, PD, 0,
                       0, KDi);
        }
        {
            ProfScope ps(prof, PS_FFT_C2C_FWD);
            fft_c2c<T>(st, planH, false, cns_f, cns_f, 1, (int64_t)Wf * PD, 0, (int64_t)Wf * PD, 0,
                       (int64_t)Wf * PD, T(1));
        }
        // LinSolveCheck (ccmod.py:783-792, as in cns_iter; the XRRS slots carry block-1 sums in
        // this call, so the three sums go to the L1, RGR and CGN slots)
        const bool lsc = p.flags & F_XRRS;
        if (lsc) {
            ProfScope ps(prof, PS_OTHER);
            launch_cns_xrrs_rhs<T>(st, Zf, sfb, cns_f, T(1), dwork_buf(), npix, CN * Cd, K, Cd, z_chan);
        }
        {
            ProfScope ps(prof, PS_SM_SOLVE);
            launch_sm_solve<T>(st, cns_f, cns_f, Zf, sfb, nullptr, T(1), npix, CN * Cd, K, W, false,
                               false, part_a, nullptr, z_chan ? 1 : Cd);
        }
        if (lsc) {
            int nbx;
            {
                ProfScope ps(prof, PS_OTHER);
                nbx = launch_cns_xrrs_fin<T>(st, Zf, cns_f, T(1), dwork_buf(), npix, CN * Cd, K, part_a, Cd,
                                             z_chan);
            }
            const int xslots[3] = {SPORCO_AMD_OUT_L1, SPORCO_AMD_OUT_RGR, SPORCO_AMD_OUT_CGN};
            const double xscales[3] = {1.0, 1.0, 1.0};
            finalize(part_a, nbx, 3, 3, xslots, xscales, out_dev);
        }
        // relax_AX, block 1: AX1nr_n = irfftn(sum_m Zf_{n,m} Xf_{n,m}) -- the inner product of
        // every (frequency, image) row with itself-indexed coefficients: npix * CN "pixels"
        // (the Cd channel blocks of an image against the image's row when they share it)
        {
            ProfScope ps(prof, PS_OTHER);
            if (Cd == 1) {
                launch_inner<T>(st, Zf, cns_f, innerb, npix * CN, 1, K);
            } else {
                if (z_chan) launch_inner<T>(st, Zf, cns_f, cns_sft, nblk, 1, K);
                else launch_inner<T>(st, Zf, cns_f, cns_sft, npix * CN, Cd, K);
                launch_swap_inner<T>(st, cns_sft, innerb, npix, N, Cd);
            }
        }
        inv2(cns_f, wk, X, PD);
        inv2(innerb, innerb, sreal, CNs);
        // consensus part: Y = Pcn(mean_n(alpha X_n + (1 - alpha) Y + U_n))
        SA_HIP(hipMemcpyAsync(cns_yold, Y, sizeof(T) * npixr * KDi, hipMemcpyDeviceToDevice, st));
        {
            ProfScope ps(prof, PS_OTHER);
            launch_cns_mean<T>(st, X, U, cns_yold, cns_m, (T)p.rlx, us, npixr, CN, KDi);
        }
        }   // phase != 2
        if (p.phase == 1) return;
        pcn_project(cns_m, Y, p.dH, p.dW, p.zero_mean != 0, nullptr);
        // block 1: AX1 = alpha AX1nr + (1 - alpha)(Y1 + S); Y1 = rho (AX1 + U1 - S) / (W^2 + rho);
        // U1 += AX1 - Y1 - S  -- the block-0 step of ConvBPDNMaskDcpl, same kernel
        MdY0Args<T> ya;
        ya.ax0nr = sreal;
        ya.y0 = Y1;
        ya.u0 = U1;
        ya.s = md_s;
        ya.w = have_wdat ? wdat : Weight<T>();
        ya.rho = (T)p.rho;
        ya.rlx = (T)p.rlx;
        ya.us = T(1);
        ya.geval_y = 1;
        ya.H = H;
        ya.W = W;
        ya.C = Cs;
        ya.N = N;
        int nb;
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_md_y0step<T>(st, ya, part_a);
        }
        {
            const int slots[4] = {SPORCO_AMD_OUT_XRRS_D2, SPORCO_AMD_OUT_XRRS_AX2,
                                  SPORCO_AMD_OUT_XRRS_B2, SPORCO_AMD_OUT_CGIT};
            const double scales[4] = {1, 1, 1, 1};
            finalize(part_a, nb, 5, 4, slots, scales, out_dev);
        }
        // consensus ustep + the X-sized sums
        {
            ProfScope ps(prof, PS_ADMM_POST);
            nb = launch_cns_ustep<T>(st, X, U, cns_yold, Y, (T)p.rlx, us, npixr, CN, KDi, part_b);
        }
        {
            const int slots[3] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_AX2, SPORCO_AMD_OUT_U2};
            const double scales[3] = {1, 1, 1};
            finalize(part_b, nb, 4, 3, slots, scales, out_dev);
        }
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_cns_ystats<T>(st, cns_yold, Y, npixr * KDi, part_a);
        }
        {
            const int slots[2] = {SPORCO_AMD_OUT_L21, SPORCO_AMD_OUT_Y2};   // (|Y - Yprev|^2 unused)
            const double scales[2] = {0, 1};
            finalize(part_a, nb, 2, 2, slots, scales, out_dev);
        }
        fwd2(Y, nullptr, T(0), cv(SPORCO_AMD_VAR_DXF), KDi);
        if (p.flags & F_RESID) {
            // dual residual: A^T u = U_n + irfftn(conj(Zf_n) rfftn(U1_n)), new duals (:993-996)
            fwd2(U1, nullptr, T(0), innerb, CNs);
            fwd2(U, nullptr, T(0), cns_f, PD);
            const cx<T> *u1b = to_blocks(innerb);
            {
                ProfScope ps(prof, PS_OTHER);
                if (Cd == 1 || z_chan) launch_conj_outer<T>(st, Zf, u1b, wk, nblk, 1, K);
                else launch_conj_outer<T>(st, Zf, u1b, wk, npix * CN, Cd, K);
                launch_lincomb<T>(st, wk, T(1), wk, T(1), cns_f, T(0), nullptr, EF * Cd);
                nb = launch_pair_stats<T>(st, wk, nullptr, nullptr, npix, PD, W, part_b);
            }
            const int slots[1] = {SPORCO_AMD_OUT_S2};
            const double scales[1] = {1.0 / ((double)H * W)};
            finalize(part_b, nb, 4, 1, slots, scales, out_dev);
        }
        if (p.flags & F_OBJ) {
            // (1/2) |W irfftn(sum_m Zf Yf - Sf)|^2 at the consensus variable (:961-970)
            {
                ProfScope ps(prof, PS_OTHER);
                if (Cd == 1) launch_inner<T>(st, cv(SPORCO_AMD_VAR_DXF), Zf, innerb, npix, CN, K);
                else launch_mc_inner<T>(st, cv(SPORCO_AMD_VAR_DXF), Zf, innerb, npix, Cd, N, K, z_chan);
                launch_lincomb<T>(st, innerb, T(1), innerb, T(-1), cv(SPORCO_AMD_VAR_SF), T(0),
                                  nullptr, npix * CNs);
            }
            inv2(innerb, innerb, sreal, CNs);
            {
                ProfScope ps(prof, PS_OTHER);
                nb = launch_mask_apply<T>(st, sreal, have_wdat ? wdat : Weight<T>(), false, H, W, Cs, N,
                                          part_a);
            }
            const int slots[1] = {SPORCO_AMD_OUT_DFID};
            const double scales[1] = {1.0};
            finalize(part_a, nb, 1, 1, slots, scales, out_dev);
            int nbc;
            {
                ProfScope ps(prof, PS_OTHER);
                launch_pcn_stats<T>(st, Y, pcn_stats_buf(), H, W, K, p.dH, p.dW, p.zero_mean != 0, Cd, fsz());
                nbc = launch_pcn_apply<T>(st, Y, pcn_stats_buf(), nullptr, H, W, K, p.dH, p.dW, part_b,
                                          Ku, Cd, fsz());
            }
            const int cslots[1] = {SPORCO_AMD_OUT_CNSTR};
            const double cscales[1] = {1.0};
            finalize(part_b, nbc, 1, 1, cslots, cscales, out_dev);
        }
    }

    void cns_iter(const sporco_amd_cns_params &p, double *out_dev) override {
        if (!have_signal) throw Error(SPORCO_AMD_ESTATE, "set_signal must be called first");
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        SA_REQUIRE(p.phase >= 0 && p.phase <= 2, "phase must be 0, 1 or 2");
        if (p.mask_dcpl) {
            cns_md_iter(p, out_dev);
            return;
        }
        // LinSolveCheck (F_XRRS; a diagnostic): the generic chain, whose solve sees the
        // right-hand sides in the natural layout
        const bool lsc = p.flags & F_XRRS;
        const bool fusedx = cns_fused() && !lsc && Cd == 1;
        const int KDi = (int)KD();              // "filters" of a consensus block: Cd K
        const int64_t PD = P * Cd;              // columns of the blocks' spectra: N Cd K
        // objective at the consensus variable Y (FLAG_FEVAL_Y / FLAG_GEVAL_Y: AuxVarObj, the
        // class default) or at the blocks X_n (fEvalX: admm/ccmod.py:870-889) and their mean
        // (gEvalY False: admm/admm.py:1641-1646)
        const bool dfid_x = (p.flags & F_OBJ) && !(p.flags & F_FEVAL_Y);
        const bool cns_x = (p.flags & F_OBJ) && !(p.flags & F_GEVAL_Y);
        SA_REQUIRE(!(cns_x && p.phase != 0), "the constraint measure at mean(X) needs all the images");
        if (fusedx) {
            if (!zf_tiled) relayout(SPORCO_AMD_VAR_ZF, true), zf_tiled = true;
        } else {
            need_natural(SPORCO_AMD_VAR_ZF);
        }
        const int64_t npixr = (int64_t)H * W;
        T *Y = rv(SPORCO_AMD_VAR_DX), *X = rv(SPORCO_AMD_VAR_CX), *U = rv(SPORCO_AMD_VAR_CU);
        const cx<T> *Zf = zf_nat();
        cns_buffers();
        if (p.phase != 2) {
        // xstep (ccmod.py:766-778): X_n = irfftn(SM(Zf_n, rho, conj(Zf_n) Sf_n + rho rfftn(Y - U_n)));
        // Y is broadcast over the images by the row transform itself
        if (fusedx) {
            // the three register-resident kernels of the sparse coding step, with the
            // coefficient spectra of each (frequency, image) tile in the dictionary's place
            if (!gramz_t) SA_HIP(hipMalloc((void **)&gramz_t, sizeof(T) * npix * CN));
            if (!gramz_valid) {
                ProfScope ps(prof, PS_OTHER);
                launch_gram_rows<T>(st, Zf, gramz_t, npix * CN, K);
                gramz_valid = true;
            }
            RowsFwdArgs<T> ra;
            ra.y = Y;
            ra.u = U;
            ra.s2 = (T)p.u_scale;
            ra.t = cns_f;
            ra.twA = twRows;
            ra.H = H;
            ra.W = W;
            ra.CN = CN;
            ra.K = K;
            ra.P = P;
            ra.y_bcast = 1;
            {
                ProfScope ps(prof, PS_ROWS_FWD);
                launch_rows_fwd<T>(st, ra);
            }
            FusedColsArgs<T> fa;
            fa.t = cns_f;
            fa.dft = Zf;
            fa.sft = sft;
            fa.gramt = gramz_t;
            fa.twA = twA;
            fa.twB = twB;
            fa.rho = (T)p.rho;
            fa.H = H;
            fa.W = W;
            fa.CN = CN;
            fa.K = K;
            fa.partials = part_f;
            fa.per_tile = 1;
            int64_t ntl;
            {
                ProfScope ps(prof, PS_FUSED_COLS);
                ntl = launch_fused_cols<T>(st, fa);
            }
            if (dfid_x) {     // (the column kernel's by-product: sum_n |Zf_n . Xf_n - Sf_n|^2)
                const int slots[1] = {SPORCO_AMD_OUT_DFID};
                const double scales[1] = {1.0 / ((double)H * W)};
                finalize(part_f, (int)ntl, 1, 1, slots, scales, out_dev);
            }
            rows_inverse_to(X, cns_f);
        } else {
        {
            ProfScope ps(prof, PS_FFT_R2C);
            fft_r2c<T>(st, planW, Y, U, (T)p.u_scale, cns_f, H, PD, (int64_t)W * PD, PD,
                       (int64_t)Wf * PD, PD, 0, 0, KDi);
        }
        {
            ProfScope ps(prof, PS_FFT_C2C_FWD);
            fft_c2c<T>(st, planH, false, cns_f, cns_f, 1, (int64_t)Wf * PD, 0, (int64_t)Wf * PD, 0,
                       (int64_t)Wf * PD, T(1));
        }
        // the right-hand sides' signal term follows the systems: (npix, N) -- or (npix, N, Cd)
        const cx<T> *sfs = cv(SPORCO_AMD_VAR_SF);
        if (Cd > 1) {
            ProfScope ps(prof, PS_OTHER);
            launch_swap_inner<T>(st, cv(SPORCO_AMD_VAR_SF), cns_sft, npix, Cd, N);
            sfs = cns_sft;
        }
        if (lsc) {
            ProfScope ps(prof, PS_OTHER);
            launch_cns_xrrs_rhs<T>(st, Zf, sfs, cns_f, (T)p.rho, dwork_buf(), npix, CN * Cd, K, Cd, z_chan);
        }
        int nbs;
        {   // (the per-image gram sum_k |Zf|^2 is formed inside the kernel)
            ProfScope ps(prof, PS_SM_SOLVE);
            nbs = launch_sm_solve<T>(st, cns_f, cns_f, Zf, sfs, nullptr, (T)p.rho, npix, CN * Cd, K, W,
                                     dfid_x, false, part_a, nullptr, z_chan ? 1 : Cd);
        }
        if (dfid_x) {
            const int slots[1] = {SPORCO_AMD_OUT_DFID};
            const double scales[1] = {1.0 / ((double)H * W)};
            finalize(part_a, nbs, 4, 1, slots, scales, out_dev);
        }
        if (lsc) {
            int nbx;
            {
                ProfScope ps(prof, PS_OTHER);
                nbx = launch_cns_xrrs_fin<T>(st, Zf, cns_f, (T)p.rho, dwork_buf(), npix, CN * Cd, K, part_a,
                                             Cd, z_chan);
            }
            const int xslots[3] = {SPORCO_AMD_OUT_XRRS_D2, SPORCO_AMD_OUT_XRRS_AX2, SPORCO_AMD_OUT_XRRS_B2};
            const double xscales[3] = {1.0, 1.0, 1.0};
            finalize(part_a, nbx, 3, 3, xslots, xscales, out_dev);
        }
        inv2(cns_f, Cd > 1 ? cns_w : work_buf(), X, PD);
        }
        // relax + ystep: Y = Pcn(mean_n(alpha X_n + (1 - alpha) Y + U_n))
        SA_HIP(hipMemcpyAsync(cns_yold, Y, sizeof(T) * npixr * KDi, hipMemcpyDeviceToDevice, st));
        {
            ProfScope ps(prof, PS_OTHER);
            launch_cns_mean<T>(st, X, U, cns_yold, cns_m, (T)p.rlx, (T)p.u_scale, npixr, CN, KDi);
        }
        }   // phase != 2
        if (p.phase == 1) return;
        pcn_project(cns_m, Y, p.dH, p.dW, p.zero_mean != 0, nullptr);
        // ustep + the X-sized sums
        int nb;
        {
            ProfScope ps(prof, PS_ADMM_POST);
            nb = launch_cns_ustep<T>(st, X, U, cns_yold, Y, (T)p.rlx, (T)p.u_scale, npixr, CN, KDi,
                                     part_b);
        }
        {
            const int slots[3] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_AX2, SPORCO_AMD_OUT_U2};
            const double scales[3] = {1, 1, 1};
            finalize(part_b, nb, 4, 3, slots, scales, out_dev);
        }
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_cns_ystats<T>(st, cns_yold, Y, npixr * KDi, part_a);
        }
        {
            const int slots[2] = {SPORCO_AMD_OUT_S2, SPORCO_AMD_OUT_Y2};
            const double scales[2] = {1, 1};
            finalize(part_a, nb, 2, 2, slots, scales, out_dev);
        }
        // the consensus dictionary's spectrum (for the objective, getdict / setdict_from_dstep)
        fwd2(Y, nullptr, T(0), cv(SPORCO_AMD_VAR_DXF), KDi);
        if (p.flags & F_OBJ) {
            const int slots[1] = {SPORCO_AMD_OUT_DFID};
            const double scales[1] = {1.0 / ((double)H * W)};
            if (dfid_x) {
                // (already summed by the X-step)
            } else if (zf_tiled) {
                if (!gpart) {
                    ccmod_groups = (int)ceil_div(768, Wf);
                    if (ccmod_groups > CN) ccmod_groups = CN;
                    if (ccmod_groups > 8) ccmod_groups = 8;
                    SA_HIP(hipMalloc((void **)&gpart, sizeof(cx<T>) * npix * K * ccmod_groups));
                }
                CcmodTiledArgs<T> ga;
                ga.zf = Zf;
                ga.d = cv(SPORCO_AMD_VAR_DXF);
                ga.sft = sft;
                ga.gpart = nullptr;
                ga.H = H;
                ga.W = W;
                ga.CN = CN;
                ga.K = K;
                ga.G = ccmod_groups;
                ga.partials = part_a;
                int64_t nwg;
                {
                    ProfScope ps(prof, PS_PGM);
                    nwg = launch_ccmod_grad_tiled<T>(st, ga);
                }
                finalize(part_a + 1, (int)nwg, 4, 1, slots, scales, out_dev);
            } else {
                {
                    ProfScope ps(prof, PS_OTHER);
                    nb = launch_ccmod_grad<T>(st, Zf, cv(SPORCO_AMD_VAR_DXF), cv(SPORCO_AMD_VAR_SF),
                                              nullptr, npix, CN, K, W, part_a, Cd, z_chan);
                }
                finalize(part_a + 1, nb, 3, 1, slots, scales, out_dev);
            }
            int nbc;
            const T *gv = Y;
            if (cns_x) {      // g is evaluated at mean_n(X_n)
                ProfScope ps(prof, PS_OTHER);
                launch_cns_mean<T>(st, X, U, cns_yold, cns_m, T(1), T(0), npixr, CN, KDi);
                gv = cns_m;
            }
            {
                ProfScope ps(prof, PS_OTHER);
                launch_pcn_stats<T>(st, gv, pcn_stats_buf(), H, W, K, p.dH, p.dW, p.zero_mean != 0, Cd, fsz());
                nbc = launch_pcn_apply<T>(st, gv, pcn_stats_buf(), nullptr, H, W, K, p.dH, p.dW, part_b,
                                          Ku, Cd, fsz());
            }
            const int cslots[1] = {SPORCO_AMD_OUT_CNSTR};
            const double cscales[1] = {1.0};
            finalize(part_b, nbc, 1, 1, cslots, cscales, out_dev);
        }
    }

    // ---- ADMM with mask decoupling (ConvBPDNMaskDcpl) ---------------------------------------
    void mdcpl_init(const void *S) override {
        SA_REQUIRE(S != nullptr, "S is null");
        const size_t nb = sizeof(T) * (int64_t)H * W * CNs;
        if (!md_s) SA_HIP(hipMalloc((void **)&md_s, nb));
        SA_HIP(hipMemcpyAsync(md_s, S, nb, hipMemcpyHostToDevice, st));
        before_state_change();
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_Y), 0, sizeof(T) * E, st));
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_U), 0, sizeof(T) * E, st));
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_MY0), 0, nb, st));
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_MU0), 0, nb, st));
        sync();
    }

    void mdcpl_iter(const sporco_amd_admm_params &p, double *out_dev) override {
        require_ready();
        SA_REQUIRE(md_s != nullptr, "mdcpl_init must be called first");
        SA_REQUIRE(p.rho > 0.0, "rho must be positive");
        SA_REQUIRE(!(p.flags & (F_JOINT | F_GRADREG | F_AMS)), "flag not valid for mask decoupling");
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        before_state_change();
        x_written();
        xf_tiled = false;
        // (a multi-channel dictionary, Cd > 1: block 0 keeps the signal's Cd channels, block 1 and
        // the coefficient maps have one -- cbpdn.py:1565-1574; the X-step is the iterated
        // Sherman-Morrison solve with rho = 1, :1621-1626)
        const int64_t ns = (int64_t)H * W * CNs;
        const T us = (T)p.u_scale;
        T *Y1 = rv(SPORCO_AMD_VAR_Y), *U1 = rv(SPORCO_AMD_VAR_U), *X = rv(SPORCO_AMD_VAR_X);
        T *Y0 = rv(SPORCO_AMD_VAR_MY0), *U0 = rv(SPORCO_AMD_VAR_MU0);
        cx<T> *Xf = cv(SPORCO_AMD_VAR_XF), *Df = cv(SPORCO_AMD_VAR_DF);
        cx<T> *Vf = cv(SPORCO_AMD_VAR_VF), *Gf = cv(SPORCO_AMD_VAR_GF);
        // xstep: b = conj(Df) rfftn(y0 - u0 + s) + rfftn(y1 - u1); (D^H D + I) Xf = b
        {
            ProfScope ps(prof, PS_OTHER);
            launch_md_pre<T>(st, Y0, U0, md_s, sreal, us, ns);
        }
        fwd2(sreal, nullptr, T(0), innerb, CNs);
        fwd2(Y1, U1, us, Vf, P);
        const bool xr = p.flags & F_XRRS;
        int nb;
        if (Cd > 1) {
            if (!ism_gam) {
                SA_HIP(hipMalloc((void **)&ism_gam, sizeof(cx<T>) * npix * Cd * K));
                SA_HIP(hipMalloc((void **)&ism_del, sizeof(cx<T>) * npix * Cd));
                SA_HIP(hipMalloc((void **)&ism_mm, sizeof(cx<T>) * npix * Cd * Cd));
            }
            ProfScope ps(prof, PS_SM_SOLVE);
            if (!ism_valid || ism_rho != 1.0) {
                launch_ism_setup<T>(st, Df, ism_gam, ism_del, ism_mm, npix, Cd, K, T(1));
                ism_valid = true;
                ism_rho = 1.0;
            }
            nb = launch_ism_solve<T>(st, Vf, Xf, Df, innerb, ism_gam, ism_del, ism_mm, T(1), npix, Cd,
                                     N, K, W, false, xr, part_a);
        } else {
            ProfScope ps(prof, PS_SM_SOLVE);
            nb = launch_sm_solve<T>(st, Vf, Xf, Df, innerb, gram, T(1), npix, CN, K, W, false, xr,
                                    part_a);
        }
        if (xr) {
            const int slots[4] = {SPORCO_AMD_OUT_DFID, SPORCO_AMD_OUT_XRRS_D2,
                                  SPORCO_AMD_OUT_XRRS_AX2, SPORCO_AMD_OUT_XRRS_B2};
            const double scales[4] = {0.0, 1.0, 1.0, 1.0};
            finalize(part_a, nb, 4, 4, slots, scales, out_dev);
        }
        inv2(Xf, work_buf(), X, P);
        // block 0: AXnr = D x, relax, y0, u0
        {
            ProfScope ps(prof, PS_OTHER);
            inner_df(Xf);
        }
        inv2(innerb, innerb, sreal, CNs);
        // block 1: relax, y1 = prox_l1 (+ NonNegCoef / NoBndryCross), u1 and the sums
        PostParams<T> pp;
        pp.x = X;
        pp.y = Y1;
        pp.u = U1;
        pp.rlx = (T)p.rlx;
        pp.thr = (T)(p.lmbda / p.rho);
        pp.thr21 = T(0);
        pp.u_scale = us;
        pp.flags = p.flags;
        pp.d = d5();
        pp.dH = p.dH;
        pp.dW = p.dW;
        pp.wl1 = wl1;
        pp.wl21 = wl21;
        pp.ams_k = Ku - 1;
        {
            ProfScope ps(prof, PS_ADMM_POST);
            nb = launch_admm_post<T>(st, pp, part_b);
        }
        {
            const int slots[6] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_S2, SPORCO_AMD_OUT_AX2,
                                  SPORCO_AMD_OUT_Y2, SPORCO_AMD_OUT_U2, SPORCO_AMD_OUT_L1};
            const double scales[6] = {1, 0, 1, 1, 1, 1};
            finalize(part_b, nb, 8, 6, slots, scales, out_dev);
        }
        MdY0Args<T> ya;
        ya.ax0nr = sreal;
        ya.y0 = Y0;
        ya.u0 = U0;
        ya.s = md_s;
        ya.w = have_wdat ? wdat : Weight<T>();
        ya.rho = (T)p.rho;
        ya.rlx = (T)p.rlx;
        ya.us = us;
        ya.geval_y = (p.flags & F_GEVAL_Y) ? 1 : 0;
        ya.H = H;
        ya.W = W;
        ya.C = Cs;
        ya.N = N;
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_md_y0step<T>(st, ya, part_a);
        }
        {
            const int slots[5] = {SPORCO_AMD_OUT_L21, SPORCO_AMD_OUT_RGR, SPORCO_AMD_OUT_CNSTR,
                                  SPORCO_AMD_OUT_CGIT, SPORCO_AMD_OUT_DFID};
            const double scales[5] = {1, 1, 1, 1, 1};
            finalize(part_a, nb, 5, 5, slots, scales, out_dev);
        }
        if (p.flags & F_RESID) {
            // dual residual (cbpdn.py:1814-1818): A^T u = irfftn(conj(Df) rfftn(u0)) + u1, its
            // norm through the half-spectrum Parseval sum
            fwd2(U0, nullptr, T(0), innerb, CNs);
            fwd2(U1, nullptr, T(0), Vf, P);
            {
                ProfScope ps(prof, PS_OTHER);
                if (Cd > 1) launch_mc_conj_outer<T>(st, Df, innerb, Gf, npix, Cd, N, K, false);
                else launch_conj_outer<T>(st, Df, innerb, Gf, npix, CN, K);
                launch_lincomb<T>(st, Gf, T(1), Gf, T(1), Vf, T(0), nullptr, EF);
                nb = launch_pair_stats<T>(st, Gf, nullptr, nullptr, npix, P, W, part_b);
            }
            const int slots[1] = {SPORCO_AMD_OUT_S2};
            const double scales[1] = {1.0 / ((double)H * W)};
            finalize(part_b, nb, 4, 1, slots, scales, out_dev);
        }
    }

    // ---- ADMM dictionary update with one dictionary copy (IterSM / CG) ----------------------
    void dstep_init(const void *Y0) override {
        require_single_channel_dict();
        eq_active = true;
        T *Y = rv(SPORCO_AMD_VAR_DX), *U = rv(SPORCO_AMD_VAR_DSU);
        const size_t nbytes = var_bytes(SPORCO_AMD_VAR_DX);
        if (Y0) {
            host_copy(SPORCO_AMD_VAR_DX, const_cast<void *>(Y0), true);
            SA_HIP(hipMemcpyAsync(U, Y, nbytes, hipMemcpyDeviceToDevice, st));
        } else {
            SA_HIP(hipMemsetAsync(Y, 0, nbytes, st));
            SA_HIP(hipMemsetAsync(U, 0, nbytes, st));
        }
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_DSX), 0, nbytes, st));
        SA_HIP(hipMemsetAsync(cv(SPORCO_AMD_VAR_DYF), 0, var_bytes(SPORCO_AMD_VAR_DYF), st));
        fwd2(Y, nullptr, T(0), cv(SPORCO_AMD_VAR_DXF), K);
        sync();
    }

    void dstep_md_init(const void *Y0, const void *S) override {
        dstep_init(Y0);
        const size_t nb = sizeof(T) * (int64_t)H * W * CN;
        if (S) {
            if (!md_s) SA_HIP(hipMalloc((void **)&md_s, nb));
            SA_HIP(hipMemcpyAsync(md_s, S, nb, hipMemcpyHostToDevice, st));
        }
        SA_REQUIRE(md_s != nullptr, "the real signal has not been set");
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_DMY0), 0, nb, st));
        SA_HIP(hipMemsetAsync(rv(SPORCO_AMD_VAR_DMU0), 0, nb, st));
        sync();
    }

    // two of the sums of launch_pair_stats over dictionary-sized spectra, read back:
    // sum |a|^2 and sum Re(conj(a) g)   (the vdot's of scipy's cg on the half spectrum)
    void cdots(const cx<T> *a, const cx<T> *g, double &a2, double &ag) {
        int nb;
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_pair_stats<T>(st, a, nullptr, g, npix, K, W, part_a);
        }
        SA_HIP(hipMemsetAsync(out_dev_own, 0, sizeof(double) * kOutSlots, st));
        const int slots[2] = {0, 1};
        const double scales[2] = {1.0, 1.0};
        finalize(part_a + 1, nb, 4, 2, slots, scales, out_dev_own);
        double tmp[kOutSlots];
        read_out(out_dev_own, tmp);
        ag = tmp[0];
        a2 = tmp[1];
    }

    // q = (Z^H Z + rho I) v on dictionary-sized spectra
    // (one wave per frequency, the coefficient spectra read once: csc_kernels.h launch_cg_op;
    // part_b[.][1] receives the partial sums of Re <v, q>)
    int dstep_op(const cx<T> *v, cx<T> *q, T rho) {
        ProfScope ps(prof, PS_SM_SOLVE);
        return launch_cg_op<T>(st, nullptr, false, cv(SPORCO_AMD_VAR_ZF), nullptr,
                               const_cast<cx<T> *>(v), q, rho, npix, CN, K, part_b);
    }
    // sum of column `idx` of a 4-wide partial array, read back
    double partial_sum(const double *part, int nb, int idx) {
        SA_HIP(hipMemsetAsync(out_dev_own, 0, sizeof(double) * kOutSlots, st));
        const int slots[1] = {0};
        const double scales[1] = {1.0};
        finalize(part + idx, nb, 4, 1, slots, scales, out_dev_own);
        double tmp[kOutSlots];
        read_out(out_dev_own, tmp);
        return tmp[0];
    }

    void dstep_iter(const sporco_amd_dstep_params &p, double *out_dev) override {
        require_single_channel_dict();
        if (!have_signal) throw Error(SPORCO_AMD_ESTATE, "set_signal must be called first");
        SA_REQUIRE(eq_active, "dstep_init must be called first");
        SA_REQUIRE(p.rho > 0.0, "rho must be positive");
        SA_REQUIRE(p.method == SPORCO_AMD_DSTEP_ISM || p.method == SPORCO_AMD_DSTEP_CG,
                   "unknown D-step method");
        SA_HIP(hipMemsetAsync(out_dev, 0, sizeof(double) * kOutSlots, st));
        need_natural(SPORCO_AMD_VAR_ZF);
        const int64_t npixr = (int64_t)H * W, nd = npix * K;
        // mask decoupling: the X-step system is Z^H Z + I whatever rho is, and the signal's
        // place in the right-hand side is taken by block 0 of y - u + c
        const bool md = p.mask_dcpl != 0;
        const T rho = md ? T(1) : (T)p.rho;
        if (md) SA_REQUIRE(md_s != nullptr, "dstep_md_init must be called first");
        T *Y = rv(SPORCO_AMD_VAR_DX), *X = rv(SPORCO_AMD_VAR_DSX), *U = rv(SPORCO_AMD_VAR_DSU);
        T *Y0 = md ? rv(SPORCO_AMD_VAR_DMY0) : nullptr, *U0 = md ? rv(SPORCO_AMD_VAR_DMU0) : nullptr;
        cx<T> *Zf = cv(SPORCO_AMD_VAR_ZF), *Sf = cv(SPORCO_AMD_VAR_SF);
        if (md) {
            {
                ProfScope ps(prof, PS_OTHER);
                launch_md_pre<T>(st, Y0, U0, md_s, sreal, (T)p.u_scale, (int64_t)H * W * CN);
            }
            fwd2(sreal, nullptr, T(0), innerb, CN);
            Sf = innerb;
            zsf_valid = false;
        }
        cx<T> *Xf = cv(SPORCO_AMD_VAR_DYF), *bf = cv(SPORCO_AMD_VAR_DVF);
        cx<T> *yuf = cv(SPORCO_AMD_VAR_DT2), *zsf = cv(SPORCO_AMD_VAR_DXFPRV);
        if (!cns_m) {
            SA_HIP(hipMalloc((void **)&cns_m, sizeof(T) * npixr * K));
            SA_HIP(hipMalloc((void **)&cns_yold, sizeof(T) * npixr * K));
        }
        if (!zsf_valid) {   // ZSf = sum_n conj(Zf_n) Sf_n  (setcoef, ccmod.py:327)
            ProfScope ps(prof, PS_OTHER);
            launch_zf_adjoint<T>(st, Zf, Sf, zsf, npix, CN, K);
            zsf_valid = !md;      // (block 0 changes every iteration)
        }
        // xstep: b = ZSf + rho rfftn(Y - U)
        fwd2(Y, U, (T)p.u_scale, yuf, K);
        const bool need_b = p.method == SPORCO_AMD_DSTEP_CG || (p.flags & F_XRRS);
        if (need_b) {
            ProfScope ps(prof, PS_OTHER);
            launch_lincomb<T>(st, bf, T(1), zsf, rho, yuf, T(0), nullptr, nd);
        }
        if (p.method == SPORCO_AMD_DSTEP_ISM) {
            if (!dism_gam) {
                SA_HIP(hipMalloc((void **)&dism_gam, sizeof(cx<T>) * npix * CN * K));
                SA_HIP(hipMalloc((void **)&dism_del, sizeof(cx<T>) * npix * CN));
                SA_HIP(hipMalloc((void **)&dism_mm, sizeof(cx<T>) * npix * CN * CN));
            }
            ProfScope ps(prof, PS_SM_SOLVE);
            if (!dism_valid || dism_rho != (double)rho) {
                launch_ism_setup<T>(st, Zf, dism_gam, dism_del, dism_mm, npix, CN, K, rho);
                dism_valid = true;
                dism_rho = (double)rho;
            }
            // the images are the rank-one terms, the dictionary the one right-hand side
            launch_ism_solve<T>(st, yuf, Xf, Zf, Sf, dism_gam, dism_del, dism_mm, rho, npix, CN, 1,
                                K, W, false, false, part_a);
        } else {
            // scipy.sparse.linalg.cg as linalg.solvemdbi_cg calls it (linalg.py:570-579), warm
            // started from the previous Xf
            cx<T> *r = cv(SPORCO_AMD_VAR_DT0), *pv = cv(SPORCO_AMD_VAR_DT1), *q = cv(SPORCO_AMD_VAR_DGF);
            double b2, x2, dummy, rr, rr_prev = 0.0, pq;
            cdots(bf, nullptr, b2, dummy);
            cdots(Xf, nullptr, x2, dummy);
            int info = 0, it = 0;
            if (b2 == 0.0) {
                SA_HIP(hipMemsetAsync(Xf, 0, sizeof(cx<T>) * nd, st));
            } else {
                const double atol = p.cg_tol * std::sqrt(b2);
                if (x2 != 0.0) {
                    dstep_op(Xf, q, rho);
                    ProfScope ps(prof, PS_OTHER);
                    launch_lincomb<T>(st, r, T(1), bf, T(-1), q, T(0), nullptr, nd);
                } else {
                    SA_HIP(hipMemcpyAsync(r, bf, sizeof(cx<T>) * nd, hipMemcpyDeviceToDevice, st));
                }
                info = p.cg_maxiter;
                if (!std::getenv("SPORCO_AMD_CG_HOST")) {
                    // Device-driven loop: alpha, beta and the stopping test stay on the device
                    // (csc_kernels.h CgCtl); the host enqueues iterations a few ahead of the
                    // last top-of-iteration it has seen finish and stops when the verdict is in.
                    if (!cg_dev) {
                        SA_HIP(hipMalloc((void **)&cg_dev, sizeof(CgCtl)));
                        SA_HIP(hipHostMalloc((void **)&cg_pin, sizeof(CgPinned), 0));
                    }
                    double *cgout = out_dev + SPORCO_AMD_OUT_CGIT;
                    // (the record is reset here, by the host: nothing of the previous solve is in
                    // flight, and the init kernel may not have run when the loop below first looks)
                    cg_pin->done = 0;
                    cg_pin->seq = 0;
                    cg_pin->it = 0;
                    cg_pin->info = p.cg_maxiter;
                    launch_cg_init(st, cg_dev, cg_pin, atol, p.cg_maxiter);
                    ProfScope ps(prof, PS_SM_SOLVE);
                    const int ahead = 2;
                    // <r, r> of the first iteration; later ones come out of the update kernel
                    const int nba = launch_pair_stats<T>(st, r, nullptr, nullptr, npix, K, W, part_a);
                    static const bool self_serve = !(std::getenv("SPORCO_AMD_CG_SELF") &&
                                                     std::atoi(std::getenv("SPORCO_AMD_CG_SELF")) == 0);
                    if (self_serve) {
                        // Two launches per iteration: every workgroup of the operator sums the
                        // residual partials itself (stopping test, beta), every workgroup of the
                        // update the <p, q> partials (alpha); workgroup 0 keeps the records
                        // (csc_kernels.h CgSelf).
                        CgSelf so, su;
                        so.c = su.c = cg_dev;
                        so.pin = su.pin = cg_pin;
                        so.cgout = su.cgout = cgout;
                        int nb_rr = nba;
                        for (int enq = 0; enq <= p.cg_maxiter; ++enq) {
                            so.prev = part_a;
                            so.prev_nb = nb_rr;
                            so.iter = enq;
                            const int nbb = launch_cg_op<T>(st, cg_dev, true, cv(SPORCO_AMD_VAR_ZF), r, pv,
                                                            q, rho, npix, CN, K, part_b, so);
                            su.prev = part_b;
                            su.prev_nb = nbb;
                            su.iter = enq;
                            nb_rr = launch_cg_update_xr<T>(st, cg_dev, T(0), Xf, r, pv, q, nd, part_a, su);
                            while (!cg_pin->done && enq + 1 - cg_pin->seq > ahead) {
                                if (hipStreamQuery(st) == hipSuccess && !cg_pin->done &&
                                    enq + 1 - cg_pin->seq > ahead)
                                    throw Error(SPORCO_AMD_EHIP, "CG: progress record not written");
                            }
                            if (cg_pin->done) break;
                        }
                    } else {
                    // An iteration: the scalar step at its top (stopping test, beta), the operator
                    // (p <- r + beta p, q = A p, <p, q>), the scalar step for alpha, the update (x,
                    // r, <r, r>).  (Folding the scalar steps into the last workgroup to finish the
                    // preceding kernel was measured: the per-workgroup ticket costs what the two
                    // small launches cost)
                    int nb_rr = nba;
                    for (int enq = 0; enq <= p.cg_maxiter; ++enq) {
                        launch_cg_ctl<T>(st, 0, part_a, nb_rr, cg_dev, cg_pin, cgout);
                        const int nbb = launch_cg_op<T>(st, cg_dev, true, cv(SPORCO_AMD_VAR_ZF), r, pv,
                                                        q, rho, npix, CN, K, part_b);
                        launch_cg_ctl<T>(st, 1, part_b, nbb, cg_dev, cg_pin, cgout);
                        nb_rr = launch_cg_update_xr<T>(st, cg_dev, T(0), Xf, r, pv, q, nd, part_a);
                        while (!cg_pin->done && enq + 1 - cg_pin->seq > ahead) {
                            if (hipStreamQuery(st) == hipSuccess && !cg_pin->done &&
                                enq + 1 - cg_pin->seq > ahead)
                                throw Error(SPORCO_AMD_EHIP, "CG: progress record not written");
                        }
                        if (cg_pin->done) break;
                    }
                    }
                    sync();
                    SA_REQUIRE(cg_pin->done, "CG: the device loop did not reach a verdict");
                    info = cg_pin->info;
                    it = cg_pin->it;
                } else
                {
                // the same kernels with the scalars read back every iteration (SPORCO_AMD_CG_HOST:
                // the loop the device-driven one is checked against)
                int nba = launch_pair_stats<T>(st, r, nullptr, nullptr, npix, K, W, part_a);
                for (it = 0; it < p.cg_maxiter; ++it) {
                    rr = partial_sum(part_a, nba, 2);
                    if (std::sqrt(rr) < atol) {
                        info = 0;
                        break;
                    }
                    {
                        ProfScope ps(prof, PS_OTHER);
                        if (it == 0)
                            SA_HIP(hipMemcpyAsync(pv, r, sizeof(cx<T>) * nd, hipMemcpyDeviceToDevice,
                                                  st));
                        else
                            launch_lincomb<T>(st, pv, T(1), r, (T)(rr / rr_prev), pv, T(0), nullptr,
                                              nd);
                    }
                    const int nbb = dstep_op(pv, q, rho);
                    pq = partial_sum(part_b, nbb, 1);
                    const T alpha = (T)(rr / pq);
                    {
                        ProfScope ps(prof, PS_OTHER);
                        nba = launch_cg_update_xr<T>(st, nullptr, alpha, Xf, r, pv, q, nd, part_a);
                    }
                    rr_prev = rr;
                }
                }
            }
            const double cgv[2] = {(double)info, (double)it};
            SA_HIP(hipMemcpyAsync(out_dev + SPORCO_AMD_OUT_CGIT, cgv, sizeof(cgv),
                                  hipMemcpyHostToDevice, st));
            sync();   // cgv is a stack array
        }
        inv2(Xf, dwork_buf(), X, K);
        if (p.flags & F_XRRS) {   // xstep_check (ccmod.py:343-357): rrs(Z^H Z Xf + rho Xf, b)
            cx<T> *q = cv(SPORCO_AMD_VAR_DGF);
            dstep_op(Xf, q, rho);
            int nb;
            {
                ProfScope ps(prof, PS_OTHER);
                nb = launch_pair_stats<T>(st, q, bf, bf, npix, K, W, part_a);
            }
            const int sl1[2] = {SPORCO_AMD_OUT_XRRS_D2, SPORCO_AMD_OUT_XRRS_B2};
            const double sc1[2] = {1.0, 1.0};
            finalize(part_a + 2, nb, 4, 2, sl1, sc1, out_dev);
            {
                ProfScope ps(prof, PS_OTHER);
                nb = launch_pair_stats<T>(st, q, nullptr, nullptr, npix, K, W, part_b);
            }
            const int sl2[1] = {SPORCO_AMD_OUT_XRRS_AX2};
            finalize(part_b + 2, nb, 4, 1, sl2, sc1, out_dev);
        }
        // relax + ystep: Y = Pcn(alpha X + (1 - alpha) Y + U); ustep and the sums
        SA_HIP(hipMemcpyAsync(cns_yold, Y, sizeof(T) * npixr * K, hipMemcpyDeviceToDevice, st));
        {
            ProfScope ps(prof, PS_OTHER);
            launch_cns_mean<T>(st, X, U, cns_yold, cns_m, (T)p.rlx, (T)p.u_scale, npixr, 1, K);
        }
        pcn_project(cns_m, Y, p.dH, p.dW, p.zero_mean != 0, nullptr);
        int nb;
        {
            ProfScope ps(prof, PS_ADMM_POST);
            nb = launch_cns_ustep<T>(st, X, U, cns_yold, Y, (T)p.rlx, (T)p.u_scale, npixr, 1, K,
                                     part_b);
        }
        {
            const int slots[3] = {SPORCO_AMD_OUT_R2, SPORCO_AMD_OUT_AX2, SPORCO_AMD_OUT_U2};
            const double scales[3] = {1, 1, 1};
            finalize(part_b, nb, 4, 3, slots, scales, out_dev);
        }
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_cns_ystats<T>(st, cns_yold, Y, npixr * K, part_a);
        }
        {
            const int slots[2] = {SPORCO_AMD_OUT_S2, SPORCO_AMD_OUT_Y2};
            const double scales[2] = {1, 1};
            finalize(part_a, nb, 2, 2, slots, scales, out_dev);
        }
        // the dictionary's spectrum (getdict / setdict_from_dstep / objective at Y)
        fwd2(Y, nullptr, T(0), cv(SPORCO_AMD_VAR_DXF), K);
        if (md) {
            // block 0: AXnr = Z d, relax, y0, u0 and its sums; then the dual residual
            // rho ||A^T u|| with the new u (ccmodmd.py:557-561) as a Parseval sum
            {
                ProfScope ps(prof, PS_OTHER);
                launch_inner<T>(st, Xf, Zf, innerb, npix, CN, K);
            }
            inv2(innerb, innerb, sreal, CN);
            MdY0Args<T> ya;
            ya.ax0nr = sreal;
            ya.y0 = Y0;
            ya.u0 = U0;
            ya.s = md_s;
            ya.w = have_wdat ? wdat : Weight<T>();
            ya.rho = (T)p.rho;
            ya.rlx = (T)p.rlx;
            ya.us = (T)p.u_scale;
            ya.geval_y = (p.flags & F_GEVAL_Y) ? 1 : 0;
            ya.H = H;
            ya.W = W;
            ya.C = C;
            ya.N = N;
            {
                ProfScope ps(prof, PS_OTHER);
                nb = launch_md_y0step<T>(st, ya, part_a);
            }
            {
                const int slots[5] = {SPORCO_AMD_OUT_L1, SPORCO_AMD_OUT_L21, SPORCO_AMD_OUT_RGR, 15,
                                      SPORCO_AMD_OUT_DFID};
                const double scales[5] = {1, 1, 1, 1, 1};
                finalize(part_a, nb, 5, 5, slots, scales, out_dev);
            }
            cx<T> *q = cv(SPORCO_AMD_VAR_DGF);
            fwd2(U0, nullptr, T(0), innerb, CN);
            fwd2(U, nullptr, T(0), yuf, K);
            {
                ProfScope ps(prof, PS_OTHER);
                launch_zf_adjoint<T>(st, Zf, innerb, q, npix, CN, K);
                launch_lincomb<T>(st, q, T(1), q, T(1), yuf, T(0), nullptr, nd);
                nb = launch_pair_stats<T>(st, q, nullptr, nullptr, npix, K, W, part_b);
            }
            const int slots[1] = {SPORCO_AMD_OUT_S2};
            const double scales[1] = {1.0 / ((double)H * W)};
            finalize(part_b, nb, 4, 1, slots, scales, out_dev);
        }
        if (p.flags & F_OBJ) {
            if (!md) {
                {
                    ProfScope ps(prof, PS_OTHER);
                    nb = launch_ccmod_grad<T>(st, Zf,
                                              (p.flags & F_FEVAL_Y) ? cv(SPORCO_AMD_VAR_DXF) : Xf, Sf,
                                              nullptr, npix, CN, K, W, part_a);
                }
                const int slots[1] = {SPORCO_AMD_OUT_DFID};
                const double scales[1] = {1.0 / ((double)H * W)};
                finalize(part_a + 1, nb, 3, 1, slots, scales, out_dev);
            }
            const T *gv = (p.flags & F_GEVAL_Y) ? Y : X;
            int nbc;
            {
                ProfScope ps(prof, PS_OTHER);
                launch_pcn_stats<T>(st, gv, pcn_stats_buf(), H, W, K, p.dH, p.dW, p.zero_mean != 0, Cd, fsz());
                nbc = launch_pcn_apply<T>(st, gv, pcn_stats_buf(), nullptr, H, W, K, p.dH, p.dW,
                                          part_b, Ku, Cd, fsz());
            }
            const int cslots[1] = {SPORCO_AMD_OUT_CNSTR};
            const double cscales[1] = {1.0};
            finalize(part_b, nbc, 1, 1, cslots, cscales, out_dev);
        }
    }

    void asum(int var, double *out_dev) override {
        SA_REQUIRE(var_is_valid(var) && !var_is_complex(var), "asum needs a real variable");
        before_read(var);
        int nb;
        {
            ProfScope ps(prof, PS_OTHER);
            nb = launch_asum<T>(st, rv(var), (int64_t)(var_bytes(var) / sizeof(T)), part_a);
        }
        const int slots[1] = {0};
        const double scales[1] = {1.0};
        finalize(part_a, nb, 1, 1, slots, scales, out_dev);
    }

    void copy(int dst, int src) override {
        SA_REQUIRE(var_bytes(dst) == var_bytes(src), "copy between variables of different size");
        before_read(src);
        if (is_pgm_iterate(dst)) pgm_leave_tiled();
        if (dst == SPORCO_AMD_VAR_XF) xf_tiled = false;
        if (dst == SPORCO_AMD_VAR_X) x_written();
        ProfScope ps(prof, PS_OTHER);
        SA_HIP(hipMemcpyAsync(var_ptr(dst), var_ptr(src), var_bytes(src), hipMemcpyDeviceToDevice, st));
    }
};

}  // namespace sporco_amd

using namespace sporco_amd;

struct sporco_amd_csc {
    std::unique_ptr<CscBase> impl;
    int device;
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

#define SA_HANDLE(h)                                                                   \
    SA_REQUIRE((h) != nullptr && (h)->impl, "null solver handle");                     \
    SA_HIP(hipSetDevice((h)->device));

extern "C" {

const char *sporco_amd_version(void) { return "sporco_amd 0.1.0 (gfx950)"; }
const char *sporco_amd_last_error(void) { return g_last_error.c_str(); }

int sporco_amd_device_count(int *count) {
    SA_API_BEGIN
    SA_REQUIRE(count != nullptr, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        n = 0;
        (void)hipGetLastError();
    }
    *count = n;
    SA_API_END
}

int sporco_amd_device_info(int device, char *name, size_t name_len, int *cu_count,
                           size_t *hbm_bytes) {
    SA_API_BEGIN
    hipDeviceProp_t prop;
    SA_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len) {
        std::strncpy(name, prop.name, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    SA_API_END
}

int sporco_amd_csc_create(const sporco_amd_dims *dims, int device, void *stream,
                          sporco_amd_csc_t *out) {
    return sporco_amd_csc_create_mc(dims, 1, device, stream, out);
}

int sporco_amd_csc_create_mc(const sporco_amd_dims *dims, int32_t dict_channels, int device,
                             void *stream, sporco_amd_csc_t *out) {
    SA_API_BEGIN
    SA_REQUIRE(dims && out, "null argument");
    SA_REQUIRE(dict_channels >= 1, "dict_channels must be >= 1");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        throw Error(SPORCO_AMD_EHIP, "no HIP device visible: libsporco_amd needs an AMD GPU");
    SA_REQUIRE(device >= 0 && device < n, "device index out of range");
    std::unique_ptr<sporco_amd_csc> h(new sporco_amd_csc);
    h->device = device;
    if (dims->dtype == SPORCO_AMD_F32)
        h->impl.reset(new Csc<float>(*dims, device, stream, dict_channels));
    else if (dims->dtype == SPORCO_AMD_F64)
        h->impl.reset(new Csc<double>(*dims, device, stream, dict_channels));
    else
        throw Error(SPORCO_AMD_EINVAL, "dtype must be SPORCO_AMD_F32 or SPORCO_AMD_F64");
    *out = h.release();
    SA_API_END
}

int sporco_amd_csc_destroy(sporco_amd_csc_t h) {
    SA_API_BEGIN
    if (h) {
        (void)hipSetDevice(h->device);
        delete h;
    }
    SA_API_END
}

int sporco_amd_csc_sync(sporco_amd_csc_t h) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->sync();
    SA_API_END
}

int sporco_amd_csc_stream(sporco_amd_csc_t h, void **stream) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(stream, "null argument");
    *stream = h->impl->stream_handle();
    SA_API_END
}

int sporco_amd_csc_set_hint(sporco_amd_csc_t h, int what, int value) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->set_hint(what, value);
    SA_API_END
}

int sporco_amd_csc_query(sporco_amd_csc_t h, int what, int *out) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "null output pointer");
    *out = h->impl->query(what);
    SA_API_END
}

int sporco_amd_csc_set_signal(sporco_amd_csc_t h, const void *S) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(S != nullptr, "S is null");
    h->impl->set_signal(S);
    SA_API_END
}

int sporco_amd_csc_set_dict(sporco_amd_csc_t h, const void *D, int32_t dH, int32_t dW) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(D != nullptr, "D is null");
    h->impl->set_dict(D, dH, dW);
    SA_API_END
}

int sporco_amd_csc_set_l1_weight(sporco_amd_csc_t h, const void *w, const int64_t shape[5]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(w == nullptr || shape != nullptr, "shape is null");
    h->impl->set_weight(0, w, shape);
    SA_API_END
}

int sporco_amd_csc_set_l21_weight(sporco_amd_csc_t h, const void *w, const int64_t shape[5]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(w == nullptr || shape != nullptr, "shape is null");
    h->impl->set_weight(1, w, shape);
    SA_API_END
}

int sporco_amd_csc_set_ams_mask(sporco_amd_csc_t h, const void *w, const int64_t shape[5]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(w == nullptr || shape != nullptr, "shape is null");
    h->impl->set_weight(2, w, shape);
    SA_API_END
}

int sporco_amd_csc_set_grad_weight(sporco_amd_csc_t h, const void *w) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->set_grad_weight(w);
    SA_API_END
}

int sporco_amd_csc_set_filter_sizes(sporco_amd_csc_t h, const int32_t *fh, const int32_t *fw) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE((fh == nullptr) == (fw == nullptr), "both size arrays, or neither");
    h->impl->set_filter_sizes(fh, fw);
    SA_API_END
}

int sporco_amd_csc_upload(sporco_amd_csc_t h, int var, const void *src) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(src != nullptr, "src is null");
    h->impl->upload(var, src);
    SA_API_END
}

int sporco_amd_csc_download(sporco_amd_csc_t h, int var, void *dst) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(dst != nullptr, "dst is null");
    h->impl->download(var, dst);
    SA_API_END
}

int sporco_amd_csc_device_ptr(sporco_amd_csc_t h, int var, void **ptr_dev) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(ptr_dev != nullptr, "ptr_dev is null");
    *ptr_dev = h->impl->device_ptr(var);
    SA_API_END
}

int sporco_amd_csc_admm_iter(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                             double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p && out, "null argument");
    h->impl->admm_iter(*p, h->impl->out_dev_default);
    h->impl->read_out(h->impl->out_dev_default, out);
    SA_API_END
}

int sporco_amd_csc_admm_run(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                            const sporco_amd_admm_ctrl *c, sporco_amd_admm_record *records,
                            int32_t *n_done, double *rho_out, double *u_scale_out,
                            sporco_amd_reduce_fn reduce, void *user) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p && c && records && n_done && rho_out && u_scale_out, "null argument");
    const int n = h->impl->admm_run(*p, *c, records, rho_out, u_scale_out, reduce, user);
    if (n < 0) {
        *n_done = 0;
        return SPORCO_AMD_EUNSUPPORTED;
    }
    *n_done = n;
    SA_API_END
}

int sporco_amd_csc_admm_iter_dev(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                                 double *out_dev) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p && out_dev, "null argument");
    h->impl->admm_iter(*p, out_dev);
    SA_API_END
}

int sporco_amd_csc_admm_xstep(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p && out, "null argument");
    h->impl->admm_xstep(*p, h->impl->out_dev_default);
    h->impl->read_out(h->impl->out_dev_default, out);
    SA_API_END
}

int sporco_amd_csc_admm_relax(sporco_amd_csc_t h, double rlx) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->admm_relax(rlx);
    SA_API_END
}

int sporco_amd_csc_admm_ystep(sporco_amd_csc_t h, const sporco_amd_admm_params *p) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p, "null argument");
    h->impl->admm_ystep(*p);
    SA_API_END
}

int sporco_amd_csc_admm_ustep(sporco_amd_csc_t h, const sporco_amd_admm_params *p) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p, "null argument");
    h->impl->admm_ustep(*p);
    SA_API_END
}

int sporco_amd_csc_admm_stats(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p && out, "null argument");
    h->impl->admm_stats(*p, h->impl->out_dev_default);
    h->impl->read_out(h->impl->out_dev_default, out);
    SA_API_END
}

int sporco_amd_csc_scale_u(sporco_amd_csc_t h, double s) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->scale_u(s);
    SA_API_END
}

int sporco_amd_csc_reconstruct(sporco_amd_csc_t h, int var, void *dst) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(dst != nullptr, "dst is null");
    h->impl->reconstruct(var, dst);
    SA_API_END
}

int sporco_amd_csc_dhs_absmax(sporco_amd_csc_t h, double *out) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    h->impl->dhs_absmax(out);
    SA_API_END
}

static double *stats_buf(sporco_amd_csc_t h) {
    if (!h->stats_dev) {
        SA_HIP(hipMalloc((void **)&h->stats_dev, sizeof(double) * kOutSlots));
        SA_HIP(hipMemset(h->stats_dev, 0, sizeof(double) * kOutSlots));
    }
    return h->stats_dev;
}

int sporco_amd_csc_pgm_grad(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->pgm_grad(var, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_pgm_commit(sporco_amd_csc_t h) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->pgm_commit();
    SA_API_END
}
int sporco_amd_csc_pgm_iter(sporco_amd_csc_t h, const sporco_amd_pgm_params *p,
                            double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p != nullptr && out != nullptr, "null argument");
    h->impl->pgm_iter(*p, stats_buf(h));
    h->impl->read_out(stats_buf(h), out);
    SA_API_END
}

int sporco_amd_csc_pgm_eval(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->pgm_eval(var, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_pgm_prox_step(sporco_amd_csc_t h, double L, double lmbda, uint32_t flags,
                                 int32_t dH, int32_t dW, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(L > 0.0, "L must be positive");
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->pgm_prox_step(L, lmbda, flags, dH, dW, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_lincomb(sporco_amd_csc_t h, int dst, double a, int va, double b, int vb,
                           double c, int vc) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->lincomb(dst, a, va, b, vb, c, vc);
    SA_API_END
}

int sporco_amd_csc_pair_stats(sporco_amd_csc_t h, int va, int vb, int vg,
                              double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->pair_stats(va, vb, vg, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_fft_var(sporco_amd_csc_t h, int real_var, int cplx_var) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->fft_var(real_var, cplx_var, false);
    SA_API_END
}

int sporco_amd_csc_ifft_var(sporco_amd_csc_t h, int cplx_var, int real_var) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->fft_var(real_var, cplx_var, true);
    SA_API_END
}

int sporco_amd_csc_copy(sporco_amd_csc_t h, int dst_var, int src_var) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->copy(dst_var, src_var);
    SA_API_END
}

int sporco_amd_csc_ccmod_setcoef(sporco_amd_csc_t h, int var) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->ccmod_setcoef(var);
    SA_API_END
}

int sporco_amd_csc_ccmod_grad(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->ccmod_grad(var, true, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_ccmod_eval(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->ccmod_grad(var, false, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_ccmod_prox_step(sporco_amd_csc_t h, double L, int32_t dH, int32_t dW,
                                   int32_t zero_mean) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(L > 0.0, "L must be positive");
    h->impl->ccmod_prox_step(L, dH, dW, zero_mean != 0);
    SA_API_END
}

int sporco_amd_csc_ccmod_cnstr(sporco_amd_csc_t h, int32_t dH, int32_t dW, int32_t zero_mean,
                               double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->ccmod_cnstr(dH, dW, zero_mean != 0, sb);
    h->impl->read_out(sb, out);
    out[0] = std::sqrt(out[0]);
    SA_API_END
}

int sporco_amd_csc_ccmod_getdict(sporco_amd_csc_t h, int32_t dH, int32_t dW, void *dst) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(dst != nullptr, "dst is null");
    h->impl->ccmod_getdict(dH, dW, dst);
    SA_API_END
}

int sporco_amd_csc_setdict_from_dstep(sporco_amd_csc_t h, int32_t dH, int32_t dW) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->setdict_from_dstep(dH, dW);
    SA_API_END
}

int sporco_amd_csc_set_data_mask(sporco_amd_csc_t h, const void *w, const int64_t shape[5]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(w == nullptr || shape != nullptr, "shape is null");
    h->impl->set_weight(3, w, shape);
    SA_API_END
}

int sporco_amd_csc_masked_grad(sporco_amd_csc_t h, int var, int32_t dstep, int32_t write_grad,
                               double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->masked_grad(var, dstep != 0, write_grad, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_cns_init(sporco_amd_csc_t h, const void *Y0, double rho) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->cns_init(Y0, rho);
    SA_API_END
}

int sporco_amd_csc_cns_mean_ptr(sporco_amd_csc_t h, void **ptr_dev, int64_t *count) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(ptr_dev && count, "null argument");
    *ptr_dev = h->impl->cns_mean_ptr(count);
    SA_API_END
}

int sporco_amd_csc_cns_md_init(sporco_amd_csc_t h, const void *S) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->cns_md_init(S);
    SA_API_END
}

int sporco_amd_csc_cns_iter(sporco_amd_csc_t h, const sporco_amd_cns_params *p,
                            double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p && out, "null argument");
    double *dev = stats_buf(h);
    h->impl->cns_iter(*p, dev);
    h->impl->read_out(dev, out);
    SA_API_END
}

int sporco_amd_csc_ccmod_sgd_step(sporco_amd_csc_t h, double eta, int32_t dH, int32_t dW,
                                  int32_t zero_mean, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->ccmod_sgd_step(eta, dH, dW, zero_mean != 0, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_mdcpl_init(sporco_amd_csc_t h, const void *S) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->mdcpl_init(S);
    SA_API_END
}

int sporco_amd_csc_mdcpl_iter(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p && out, "null argument");
    double *dev = stats_buf(h);
    h->impl->mdcpl_iter(*p, dev);
    h->impl->read_out(dev, out);
    SA_API_END
}

int sporco_amd_csc_dstep_init(sporco_amd_csc_t h, const void *Y0) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->dstep_init(Y0);
    SA_API_END
}

int sporco_amd_csc_dstep_md_init(sporco_amd_csc_t h, const void *Y0, const void *S) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->dstep_md_init(Y0, S);
    SA_API_END
}

int sporco_amd_csc_dstep_iter(sporco_amd_csc_t h, const sporco_amd_dstep_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(p && out, "null argument");
    double *dev = stats_buf(h);
    h->impl->dstep_iter(*p, dev);
    h->impl->read_out(dev, out);
    SA_API_END
}

int sporco_amd_csc_asum(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->asum(var, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_profile(sporco_amd_csc_t h, int enable) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->sync();
    h->impl->prof.drain();
    if (enable) {
        // create the event pool up front: hipEventCreate is slow enough to
        // distort a timed region if it happens lazily inside it
        Profiler &pr = h->impl->prof;
        while (pr.pool.size() < 512) {
            hipEvent_t e;
            SA_HIP(hipEventCreate(&e));
            pr.pool.push_back(e);
        }
    }
    h->impl->prof.on = enable != 0;
    SA_API_END
}

int sporco_amd_profile_slots(void) { return PS_COUNT; }

int sporco_amd_csc_profile_read(sporco_amd_csc_t h, int slot, const char **name, double *total_ms,
                                int64_t *launches) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(slot >= 0 && slot < PS_COUNT, "timing slot out of range");
    h->impl->prof.drain();
    if (name) *name = kProfNames[slot];
    if (total_ms) *total_ms = h->impl->prof.total_ms[slot];
    if (launches) *launches = h->impl->prof.count[slot];
    h->impl->prof.total_ms[slot] = 0.0;
    h->impl->prof.count[slot] = 0;
    SA_API_END
}

}  // extern "C"

// ---------------------------------------------------------------------------
// stateless primitives
// ---------------------------------------------------------------------------
namespace {

struct DevBuf {
    void *p = nullptr;
    explicit DevBuf(size_t bytes) { SA_HIP(hipMalloc(&p, bytes ? bytes : 1)); }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    template <typename U> U *as() { return static_cast<U *>(p); }
};

void require_gpu() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        throw Error(SPORCO_AMD_EHIP, "no HIP device visible: libsporco_amd needs an AMD GPU");
}

template <typename T> void prim_rfftn2(int H, int W, int64_t P, const void *in, void *out) {
    const int64_t Wf = W / 2 + 1;
    DevBuf din(sizeof(T) * H * W * P), dout(sizeof(cx<T>) * H * Wf * P);
    FftPlan pw, ph;
    pw.init(W);
    ph.init(H);
    SA_HIP(hipMemcpy(din.p, in, sizeof(T) * H * W * P, hipMemcpyHostToDevice));
    rfft2<T>(nullptr, pw, ph, din.as<T>(), nullptr, T(0), dout.as<cx<T>>(), H, W, P);
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(out, dout.p, sizeof(cx<T>) * H * Wf * P, hipMemcpyDeviceToHost));
    pw.destroy();
    ph.destroy();
}

template <typename T> void prim_irfftn2(int H, int W, int64_t P, const void *in, void *out) {
    const int64_t Wf = W / 2 + 1;
    DevBuf din(sizeof(cx<T>) * H * Wf * P), dout(sizeof(T) * H * W * P);
    FftPlan pw, ph;
    pw.init(W);
    ph.init(H);
    SA_HIP(hipMemcpy(din.p, in, sizeof(cx<T>) * H * Wf * P, hipMemcpyHostToDevice));
    irfft2<T>(nullptr, pw, ph, din.as<cx<T>>(), din.as<cx<T>>(), dout.as<T>(), H, W, P);
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(out, dout.p, sizeof(T) * H * W * P, hipMemcpyDeviceToHost));
    pw.destroy();
    ph.destroy();
}

// signal.tikhonov_filter on device arrays (csc_kernels.h has the elementwise pieces)
template <typename T>
void prim_tikhonov_dev(int H, int W, int64_t P, const void *s, double lmbda, int npd, void *slp,
                       void *shp) {
    const int Hp = H + 2 * npd, Wp = W + 2 * npd;
    const int64_t Wfp = Wp / 2 + 1;
    DevBuf sp(sizeof(T) * (size_t)Hp * Wp * P), spf(sizeof(cx<T>) * (size_t)Hp * Wfp * P);
    FftPlan pw, ph;
    pw.init(Wp);
    ph.init(Hp);
    launch_sympad<T>(nullptr, static_cast<const T *>(s), sp.as<T>(), H, W, P, npd);
    rfft2<T>(nullptr, pw, ph, sp.as<T>(), nullptr, T(0), spf.as<cx<T>>(), Hp, Wp, P);
    launch_tikhonov_divide<T>(nullptr, spf.as<cx<T>>(), Hp, Wp, P, lmbda);
    irfft2<T>(nullptr, pw, ph, spf.as<cx<T>>(), spf.as<cx<T>>(), sp.as<T>(), Hp, Wp, P);
    launch_crop_highpass<T>(nullptr, sp.as<T>(), static_cast<const T *>(s), static_cast<T *>(slp),
                            static_cast<T *>(shp), H, W, P, npd);
    SA_HIP(hipDeviceSynchronize());
    pw.destroy();
    ph.destroy();
}

template <typename T>
void prim_fftconv_dev(int ha, int wa, const int64_t *da, const void *a, int hb, int wb,
                      const int64_t *db, const void *b, int oh, int ow, void *out) {
    const int H = std::max(ha, hb), W = std::max(wa, wb);
    const int64_t Wf = W / 2 + 1;
    int64_t d[3], sa[3], sb[3], pa = 1, pb = 1, po = 1;
    for (int i = 0; i < 3; ++i) {
        d[i] = std::max(da[i], db[i]);
        SA_REQUIRE((da[i] == 1 || da[i] == d[i]) && (db[i] == 1 || db[i] == d[i]) && d[i] >= 1,
                   "fftconv: the trailing axes must broadcast");
        pa *= da[i];
        pb *= db[i];
        po *= d[i];
    }
    int64_t ra = 1, rb = 1;
    for (int i = 2; i >= 0; --i) {
        sa[i] = da[i] == 1 ? 0 : ra;
        sb[i] = db[i] == 1 ? 0 : rb;
        ra *= da[i];
        rb *= db[i];
    }
    DevBuf pada(sizeof(T) * (size_t)H * W * pa), padb(sizeof(T) * (size_t)H * W * pb);
    DevBuf af(sizeof(cx<T>) * (size_t)H * Wf * pa), bf(sizeof(cx<T>) * (size_t)H * Wf * pb);
    DevBuf of(sizeof(cx<T>) * (size_t)H * Wf * po), tmp(sizeof(T) * (size_t)H * W * po);
    FftPlan pw, ph;
    pw.init(W);
    ph.init(H);
    launch_zeropad2<T>(nullptr, static_cast<const T *>(a), pada.as<T>(), ha, wa, H, W, pa);
    launch_zeropad2<T>(nullptr, static_cast<const T *>(b), padb.as<T>(), hb, wb, H, W, pb);
    rfft2<T>(nullptr, pw, ph, pada.as<T>(), nullptr, T(0), af.as<cx<T>>(), H, W, pa);
    rfft2<T>(nullptr, pw, ph, padb.as<T>(), nullptr, T(0), bf.as<cx<T>>(), H, W, pb);
    launch_cmul_bcast<T>(nullptr, af.as<cx<T>>(), bf.as<cx<T>>(), of.as<cx<T>>(), (int64_t)H * Wf, d, sa,
                         sb, pa, pb);
    const bool roll = oh != 0 || ow != 0;
    T *dst = roll ? tmp.as<T>() : static_cast<T *>(out);
    irfft2<T>(nullptr, pw, ph, of.as<cx<T>>(), of.as<cx<T>>(), dst, H, W, po);
    if (roll) launch_roll2<T>(nullptr, tmp.as<T>(), static_cast<T *>(out), H, W, po, oh, ow);
    SA_HIP(hipDeviceSynchronize());
    pw.destroy();
    ph.destroy();
}

template <typename T> void prim_axpby(int64_t n, double a, const void *x, double b, const void *y,
                                      void *out) {
    launch_axpby<T>(nullptr, (T)a, static_cast<const T *>(x), (T)b, static_cast<const T *>(y),
                    static_cast<T *>(out), n);
    SA_HIP(hipDeviceSynchronize());
}

template <typename T>
void prim_solvedbi_sm(int64_t npix, int64_t CN, int K, const void *ah, double rho, const void *b,
                      void *x) {
    // General right-hand side b: solve through the same kernel by passing
    // yuf = b / rho and Sf = 0  (b = conj(Df)*0 + rho*yuf).
    DevBuf dah(sizeof(cx<T>) * npix * K), db(sizeof(cx<T>) * npix * CN * K),
        dsf(sizeof(cx<T>) * npix * CN), dg(sizeof(T) * npix),
        dpart(sizeof(double) * kMaxPartialBlocks * 4);
    SA_HIP(hipMemcpy(dah.p, ah, sizeof(cx<T>) * npix * K, hipMemcpyHostToDevice));
    SA_HIP(hipMemcpy(db.p, b, sizeof(cx<T>) * npix * CN * K, hipMemcpyHostToDevice));
    SA_HIP(hipMemset(dsf.p, 0, sizeof(cx<T>) * npix * CN));
    launch_scale<T>(nullptr, db.as<T>(), (T)(1.0 / rho), 2 * npix * CN * K);
    launch_gram<T>(nullptr, dah.as<cx<T>>(), dg.as<T>(), npix, K);
    launch_sm_solve<T>(nullptr, db.as<cx<T>>(), db.as<cx<T>>(), dah.as<cx<T>>(), dsf.as<cx<T>>(),
                       dg.as<T>(), (T)rho, npix, (int)CN, K, 2, false, false, dpart.as<double>());
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(x, db.p, sizeof(cx<T>) * npix * CN * K, hipMemcpyDeviceToHost));
}

template <typename T>
void prim_inner(int64_t npix, int64_t CN, int K, const void *x, const void *y, void *out) {
    DevBuf dx(sizeof(cx<T>) * npix * K), dy(sizeof(cx<T>) * npix * CN * K),
        dout(sizeof(cx<T>) * npix * CN);
    SA_HIP(hipMemcpy(dx.p, x, sizeof(cx<T>) * npix * K, hipMemcpyHostToDevice));
    SA_HIP(hipMemcpy(dy.p, y, sizeof(cx<T>) * npix * CN * K, hipMemcpyHostToDevice));
    launch_inner<T>(nullptr, dx.as<cx<T>>(), dy.as<cx<T>>(), dout.as<cx<T>>(), npix, (int)CN, K);
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(out, dout.p, sizeof(cx<T>) * npix * CN, hipMemcpyDeviceToHost));
}

template <typename T> void prim_prox_l1(int64_t n, const void *v, double alpha, void *out) {
    DevBuf dv(sizeof(T) * n), dpart(sizeof(double) * kMaxPartialBlocks);
    SA_HIP(hipMemcpy(dv.p, v, sizeof(T) * n, hipMemcpyHostToDevice));
    // view as (1, 1, 1, 1, n) when n fits an int, else split
    SA_REQUIRE(n < (int64_t)1 << 31, "prox_l1 primitive: too many elements");
    Dims5 d{1, 1, 1, 1, (int)n};
    launch_prox_l1<T>(nullptr, dv.as<T>(), dv.as<T>(), (T)alpha, 0u, d, 1, 1, Weight<T>(),
                      dpart.as<double>());
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(out, dv.p, sizeof(T) * n, hipMemcpyDeviceToHost));
}

// array-valued threshold: alpha has extent 1 or the full extent on each of the five axes
template <typename T>
void prim_prox_l1w(const int64_t *shape, const void *v, const int64_t *ashape, const void *alpha,
                   void *out) {
    int64_t n = 1, na = 1;
    for (int i = 0; i < 5; ++i) {
        SA_REQUIRE(shape[i] >= 1 && shape[i] < ((int64_t)1 << 31), "bad shape");
        SA_REQUIRE(ashape[i] == 1 || ashape[i] == shape[i],
                   "alpha must have extent 1 or the full extent on every axis");
        n *= shape[i];
        na *= ashape[i];
    }
    DevBuf dv(sizeof(T) * n), da(sizeof(T) * na), dpart(sizeof(double) * kMaxPartialBlocks);
    SA_HIP(hipMemcpy(dv.p, v, sizeof(T) * n, hipMemcpyHostToDevice));
    SA_HIP(hipMemcpy(da.p, alpha, sizeof(T) * na, hipMemcpyHostToDevice));
    Weight<T> w;
    w.ptr = da.as<T>();
    int64_t st = 1;
    for (int i = 4; i >= 0; --i) {
        w.stride[i] = ashape[i] == 1 ? 0 : st;
        st *= ashape[i];
    }
    Dims5 d{(int)shape[0], (int)shape[1], (int)shape[2], (int)shape[3], (int)shape[4]};
    launch_prox_l1<T>(nullptr, dv.as<T>(), dv.as<T>(), T(1), 0u, d, 1, 1, w, dpart.as<double>());
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(out, dv.p, sizeof(T) * n, hipMemcpyDeviceToHost));
}

template <typename T>
void prim_prox_sl1l2(int64_t outer, int C, int64_t inner, const void *v, double alpha, double beta,
                     void *out) {
    const int64_t n = outer * C * inner;
    DevBuf dv(sizeof(T) * n), dout(sizeof(T) * n);
    SA_HIP(hipMemcpy(dv.p, v, sizeof(T) * n, hipMemcpyHostToDevice));
    launch_prox_sl1l2<T>(nullptr, dv.as<T>(), dout.as<T>(), (T)alpha, (T)beta, outer, C, inner);
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(out, dout.p, sizeof(T) * n, hipMemcpyDeviceToHost));
}

template <typename T> void prim_rfl2norm2(int H, int W, int64_t P, const void *xf, double *out) {
    const int64_t npix = (int64_t)H * (W / 2 + 1);
    DevBuf dx(sizeof(cx<T>) * npix * P), dpart(sizeof(double) * kMaxPartialBlocks),
        dout(sizeof(double) * kOutSlots);
    SA_HIP(hipMemcpy(dx.p, xf, sizeof(cx<T>) * npix * P, hipMemcpyHostToDevice));
    const int nb = launch_rfl2norm2<T>(nullptr, dx.as<cx<T>>(), nullptr, npix, P, W,
                                       dpart.as<double>());
    const int slots[1] = {0};
    const double scales[1] = {1.0 / ((double)H * W)};
    launch_finalize(nullptr, dpart.as<double>(), nb, 1, 1, slots, scales, false, dout.as<double>());
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(out, dout.p, sizeof(double), hipMemcpyDeviceToHost));
}

}  // namespace

extern "C" {

#define SA_DISPATCH(dtype, fn, ...)                                                    \
    require_gpu();                                                                     \
    if ((dtype) == SPORCO_AMD_F32)                                                     \
        fn<float>(__VA_ARGS__);                                                        \
    else if ((dtype) == SPORCO_AMD_F64)                                                \
        fn<double>(__VA_ARGS__);                                                       \
    else                                                                               \
        throw Error(SPORCO_AMD_EINVAL, "dtype must be SPORCO_AMD_F32 or SPORCO_AMD_F64");

int sporco_amd_rfftn2(int dtype, int32_t H, int32_t W, int64_t P, const void *in, void *out) {
    SA_API_BEGIN
    SA_REQUIRE(in && out && H >= 1 && W >= 1 && P >= 1, "bad argument");
    SA_DISPATCH(dtype, prim_rfftn2, H, W, P, in, out)
    SA_API_END
}

int sporco_amd_irfftn2(int dtype, int32_t H, int32_t W, int64_t P, const void *in, void *out) {
    SA_API_BEGIN
    SA_REQUIRE(in && out && H >= 1 && W >= 1 && P >= 1, "bad argument");
    SA_DISPATCH(dtype, prim_irfftn2, H, W, P, in, out)
    SA_API_END
}

int sporco_amd_solvedbi_sm(int dtype, int64_t npix, int64_t CN, int32_t K, const void *ah,
                           double rho, const void *b, void *x) {
    SA_API_BEGIN
    SA_REQUIRE(ah && b && x && npix >= 1 && CN >= 1 && K >= 1 && rho != 0.0, "bad argument");
    SA_DISPATCH(dtype, prim_solvedbi_sm, npix, CN, K, ah, rho, b, x)
    SA_API_END
}

int sporco_amd_inner(int dtype, int64_t npix, int64_t CN, int32_t K, const void *x, const void *y,
                     void *out) {
    SA_API_BEGIN
    SA_REQUIRE(x && y && out && npix >= 1 && CN >= 1 && K >= 1, "bad argument");
    SA_DISPATCH(dtype, prim_inner, npix, CN, K, x, y, out)
    SA_API_END
}

int sporco_amd_prox_l1(int dtype, int64_t n, const void *v, double alpha, void *out) {
    SA_API_BEGIN
    SA_REQUIRE(v && out && n >= 1, "bad argument");
    SA_DISPATCH(dtype, prim_prox_l1, n, v, alpha, out)
    SA_API_END
}

int sporco_amd_dev_malloc(size_t bytes, void **ptr_dev) {
    SA_API_BEGIN
    SA_REQUIRE(ptr_dev != nullptr, "null argument");
    require_gpu();
    SA_HIP(hipMalloc(ptr_dev, bytes ? bytes : 1));
    SA_API_END
}
int sporco_amd_dev_free(void *ptr_dev) {
    SA_API_BEGIN
    if (ptr_dev) SA_HIP(hipFree(ptr_dev));
    SA_API_END
}
int sporco_amd_dev_upload(void *dst_dev, const void *src_host, size_t bytes) {
    SA_API_BEGIN
    SA_REQUIRE(dst_dev && src_host, "null argument");
    SA_HIP(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    SA_API_END
}
int sporco_amd_dev_download(void *dst_host, const void *src_dev, size_t bytes) {
    SA_API_BEGIN
    SA_REQUIRE(dst_host && src_dev, "null argument");
    SA_HIP(hipDeviceSynchronize());
    SA_HIP(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    SA_API_END
}
int sporco_amd_dev_axpby(int dtype, int64_t n, double a, const void *x, double b, const void *y,
                         void *out) {
    SA_API_BEGIN
    SA_REQUIRE(x && out && n >= 1, "bad argument");
    SA_DISPATCH(dtype, prim_axpby, n, a, x, b, y, out)
    SA_API_END
}
int sporco_amd_tikhonov_filter_dev(int dtype, int32_t H, int32_t W, int64_t P, const void *s_dev,
                                   double lmbda, int32_t npd, void *slp_dev, void *shp_dev) {
    SA_API_BEGIN
    SA_REQUIRE(s_dev && slp_dev && shp_dev && H >= 1 && W >= 1 && P >= 1 && npd >= 0, "bad argument");
    SA_DISPATCH(dtype, prim_tikhonov_dev, H, W, P, s_dev, lmbda, npd, slp_dev, shp_dev)
    SA_API_END
}
int sporco_amd_fftconv_dev(int dtype, int32_t ha, int32_t wa, const int64_t da[3], const void *a_dev,
                           int32_t hb, int32_t wb, const int64_t db[3], const void *b_dev,
                           int32_t origin_h, int32_t origin_w, void *out_dev) {
    SA_API_BEGIN
    SA_REQUIRE(da && db && a_dev && b_dev && out_dev && ha >= 1 && wa >= 1 && hb >= 1 && wb >= 1,
               "bad argument");
    SA_DISPATCH(dtype, prim_fftconv_dev, ha, wa, da, a_dev, hb, wb, db, b_dev, origin_h, origin_w,
                out_dev)
    SA_API_END
}
int sporco_amd_csc_set_signal_dev(sporco_amd_csc_t h, const void *S_dev) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(S_dev != nullptr, "S_dev is null");
    h->impl->set_signal_dev(S_dev);
    SA_API_END
}
int sporco_amd_csc_reconstruct_dev(sporco_amd_csc_t h, int var, void *dst_dev) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(dst_dev != nullptr, "dst_dev is null");
    h->impl->reconstruct_dev(var, dst_dev);
    SA_API_END
}
int sporco_amd_transfer_stats(int64_t out[4], int reset) {
    SA_API_BEGIN
    SA_REQUIRE(out != nullptr, "null argument");
    for (int i = 0; i < 4; ++i) {
        out[i] = g_xfer[i];
        if (reset) g_xfer[i] = 0;
    }
    SA_API_END
}

int sporco_amd_prox_l1w(int dtype, const int64_t shape[5], const void *v, const int64_t ashape[5],
                        const void *alpha, void *out) {
    SA_API_BEGIN
    SA_REQUIRE(shape && v && ashape && alpha && out, "null argument");
    SA_DISPATCH(dtype, prim_prox_l1w, shape, v, ashape, alpha, out)
    SA_API_END
}

int sporco_amd_prox_sl1l2(int dtype, int64_t outer, int32_t C, int64_t inner, const void *v,
                          double alpha, double beta, void *out) {
    SA_API_BEGIN
    SA_REQUIRE(v && out && outer >= 1 && C >= 1 && inner >= 1, "bad argument");
    SA_DISPATCH(dtype, prim_prox_sl1l2, outer, C, inner, v, alpha, beta, out)
    SA_API_END
}

int sporco_amd_rfl2norm2(int dtype, int32_t H, int32_t W, int64_t P, const void *xf, double *out) {
    SA_API_BEGIN
    SA_REQUIRE(xf && out && H >= 1 && W >= 1 && P >= 1, "bad argument");
    SA_DISPATCH(dtype, prim_rfl2norm2, H, W, P, xf, out)
    SA_API_END
}

}  // extern "C"
