// csc_api.hip -- the solver behind the C ABI of libsporco_amd.so (include/sporco_amd.h).
//
// The handle owns every device array of one ConvBPDN problem; the host side
// (sporco_amd/*.py, which mirrors the reference classes) keeps only scalars and
// the iteration loop.  Arrays stay resident in HBM for the life of the handle.
//
// template Csc<T> is one class; its member functions are grouped by solver family in the
// api_*.inc files included inside the class body below (they are not stand-alone sources).
// The extern "C" entry points are in csc_abi.hip, the stateless primitives in csc_prims.hip.
#include "csc_impl.h"

#include <chrono>
#include <string>

namespace sporco_amd {

thread_local std::string g_last_error;
std::atomic<int64_t> g_xfer[4];
const char *kProfNames[PS_COUNT] = {"fft_r2c_rows",     "fft_c2c_cols_fwd", "sm_solve",
                                           "fft_c2c_cols_inv", "fft_c2r_rows",     "admm_post",
                                           "fused_cols_sm",    "rows_fwd",         "rows_inv_post",
                                           "rows_inv_post_emit",
                                           "rows_fwd_v",       "rows_inv_post_v",  "rows_inv_post_v_emit",
                                           "pgm_grad_ifft",    "pgm_rows_prox",    "pgm_fft_momentum",
                                           "finalize",         "pgm_elementwise",  "other",
                                           "admm_persist_run",
                                    "setcoef_rows",     "setcoef_cols",     "ccmod_grad_tiled",
                                    "fft_c2r_vpost",    "fft_c2r_vpost_emit"};

// Environment switches (include/sporco_amd.h lists them; tests and measurements, none is needed in
// normal use).  Read ONCE, when a handle is made -- except SPORCO_AMD_HOST_LOOP and
// SPORCO_AMD_RUN_LAG, which callers flip between runs of one handle (bench.py's profiled pass, the
// early-stop tests) and sporco_amd_csc_admm_run reads at its entry.
struct Switches {
    bool unfused, old_rows, no_pad, no_vform, no_speculation, no_cols_sm, md_generic, cns_generic, cg_host,
        placement_off, placement_force;
    int persist;              // SPORCO_AMD_PERSIST: -1 unset (the hint decides), 0, 1
    int cols_sm_force_slab;   // SPORCO_AMD_COLS_SM_FORCE_SLAB: 0 unset
    static bool set(const char *name) { return std::getenv(name) != nullptr; }
    static Switches read() {
        Switches s;
        s.unfused = set("SPORCO_AMD_UNFUSED");
        s.old_rows = set("SPORCO_AMD_OLD_ROWS");
        s.no_pad = set("SPORCO_AMD_NO_PAD");
        s.no_vform = set("SPORCO_AMD_NO_VFORM");
        s.no_speculation = set("SPORCO_AMD_NO_SPECULATION");
        s.no_cols_sm = set("SPORCO_AMD_NO_COLS_SM");
        s.md_generic = set("SPORCO_AMD_MD_GENERIC");
        s.cns_generic = set("SPORCO_AMD_CNS_GENERIC");
        s.cg_host = set("SPORCO_AMD_CG_HOST");
        const char *e = std::getenv("SPORCO_AMD_PLACEMENT");
        s.placement_off = e && e[0] == '0';
        s.placement_force = e && e[0] == 'f';      // "force": also for small arrays (tests)
        e = std::getenv("SPORCO_AMD_PERSIST");
        s.persist = e ? (e[0] == '1' ? 1 : 0) : -1;
        e = std::getenv("SPORCO_AMD_COLS_SM_FORCE_SLAB");
        s.cols_sm_force_slab = e ? std::atoi(e) : 0;
        return s;
    }
};

template <typename T> struct Csc : CscBase {
    const Switches sw = Switches::read();
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
        return rows_ok && fused && cols256 && !mr && !sw.cns_generic;
    }
    bool ism_valid = false;
    double ism_rho = 0.0, ism_mu = -1.0;   // (ism_mu: mu of the gradient diagonal, -1 = none)
    int64_t P, E, npix, EF;  // P = C*N*K, E = H*W*P, npix = H*Wf, EF = npix*P
    FftPlan planW, planH;
    // Volumes (dimN = 3, sporco/cnvrep.py:33-198 with three spatial axes): the first two axes
    // (depth, height) arrive folded into H = depth * Hs -- the memory layout (depth, Hs, W, C, N, K)
    // IS the layout (H, W, C, N, K) -- and everything per pixel, per frequency or per row works on
    // the folded array as it stands; only the transform along the folded axis is two passes, Hs
    // then depth (api_transforms.inc fwd2 / inv2).  Generic chain only, ADMM sparse coding only
    // (the C ABI refuses the other entry points: csc_impl.h SA_HANDLE).
    int depth = 1, Hs = 0;
    FftPlan planD;
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
    // Complex-valued signals and dictionary (set_dict_imag): the real and imaginary parts are the
    // channels c and c + C/2 of the real machinery, df_im holds rfftn of the dictionary's imaginary
    // part beside VAR_DF, and the X-step / inner products pair them (ck_admm.hip sm_cplx_kernel).
    // Generic chain only.
    bool cplx = false;
    cx<T> *df_im = nullptr;
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
    // Mixed-radix shape (round 6): H or W one of 320 / 384 / 448 / 480 -- the register-resident kernels
    // exist for a single-channel dictionary: ConvBPDN (scalar or array L1Weight, NonNegCoef,
    // NoBndryCross), ConvBPDNJoint, ConvBPDNGradReg, AddMaskSim with K <= 256; FISTA and the tile-major
    // dictionary update with K <= 64 (csc_rows_mr.hip, csc_pgm_mr.hip, csc_fused.h); LinSolveCheck,
    // mask decoupling and consensus stay on the generic chain for such a handle.
    bool mr = false;
    bool mr_ok(const sporco_amd_admm_params &p) const {
        return !mr || (Cd == 1 && !wl21.ptr && !(p.flags & F_XRRS));
    }
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
    bool md_x_pending = false;   // X of a fused mask-decoupled iteration: rows_inverse of the Xf buffer
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
    // The generic chain's single-array state (api_admm.inc admm_iter): while gv_live the iterate
    // is V = AX + U in the buffer of vars[U] (updated in place; vars[Y] is stale), produced with
    // the threshold gv_thr.  Never live together with v_live; ensure_yu() resolves either.
    bool gv_live = false, gv_nonneg = false;
    T gv_thr = T(0);
    uint64_t gen_epoch = ~(uint64_t)0;
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

    int padded_filters(int H_, int W_, int K_) const {
        if (K_ % 2 == 0 || sw.unfused || sw.no_pad)
            return K_;
        const bool cols = fused_cols_supported<T>(H_, K_ + 1) || fused_slabs_supported<T>(H_, K_ + 1);
        return (cols && rows_supported<T>(W_, K_ + 1)) ? K_ + 1 : K_;
    }

    Csc(const sporco_amd_dims &d, int dev, void *stream, int cd, int depth_ = 1) : dm(d), device(dev) {
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
        depth = depth_;
        SA_REQUIRE(depth >= 1 && H % depth == 0 && (depth == 1 || cd == 1),
                   "volume handle: depth divides H, single-channel dictionary");
        Hs = H / depth;
        const bool generic_only = sw.unfused || depth > 1;
        planW.init(W);
        planH.init(Hs);
        if (depth > 1) planD.init(depth);
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
                !generic_only;
        cols256 = H == 128 || H == 256 || H == 512;   // (every column kernel family has the 32 x 4 split)
        fused_slabs = Cd == 1 && fused_slabs_supported<T>(H, K) && !generic_only;
        fused_mc = Cd > 1 && fused_mc_supported<T>(H, K, Cd) && K % 2 == 0 &&
                   !generic_only;
        if (fused_mc) {
            SA_HIP(hipMalloc((void **)&dft_mc, sizeof(cx<T>) * npix * Cd * K));
            SA_HIP(hipMalloc((void **)&sft_mc, sizeof(cx<T>) * npix * CNs));
            SA_HIP(hipMalloc((void **)&bt_mc, sizeof(T) * npix * 2 * Cd * Cd));
            SA_HIP(hipMalloc((void **)&part_f, sizeof(double) * 2 * (int64_t)Wf * CN));
            const int ntw = fused_twiddle_count(H);
            SA_HIP(hipMalloc((void **)&twA, sizeof(cx<T>) * ntw));
            SA_HIP(hipMalloc((void **)&twB, sizeof(cx<T>) * ntw));
            std::vector<cx<T>> ta(ntw), tb(ntw);
            fused_twiddles<T>(H, K, ta.data(), tb.data());
            SA_HIP(hipMemcpy(twA, ta.data(), sizeof(cx<T>) * ntw, hipMemcpyHostToDevice));
            SA_HIP(hipMemcpy(twB, tb.data(), sizeof(cx<T>) * ntw, hipMemcpyHostToDevice));
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
            const int ntw = fused_twiddle_count(H);
            SA_HIP(hipMalloc((void **)&twA, sizeof(cx<T>) * ntw));
            SA_HIP(hipMalloc((void **)&twB, sizeof(cx<T>) * ntw));
            std::vector<cx<T>> ta(ntw), tb(ntw);
            fused_twiddles<T>(H, K, ta.data(), tb.data());
            SA_HIP(hipMemcpy(twA, ta.data(), sizeof(cx<T>) * ntw, hipMemcpyHostToDevice));
            SA_HIP(hipMemcpy(twB, tb.data(), sizeof(cx<T>) * ntw, hipMemcpyHostToDevice));
        }
        rows_ok = (fused || fused_slabs || fused_mc) && rows_supported<T>(W, K) &&
                  !sw.old_rows;
        mr = std::is_same<T, float>::value && (fused_mr_height(H) || rows_mr_width(W));
        if (mr && !((fused || fused_slabs) && rows_ok)) {
            // (a mixed-radix side needs the register kernels on BOTH sides and K <= 64: otherwise
            // the whole handle is a generic-chain one)
            rows_ok = false;
            if (fused_mr_height(H)) fused = fused_slabs = false;
            mr = false;
        }
        if (mr) cols256 = false;      // (consensus, mask decoupling, the K > 64 families: generic)
        // (a mixed-radix height has no tail form: 64 < K <= 72 runs two slabs there)
        tail_mode = fused_slabs && K - 64 <= kTailMax && !fused_mr_height(H);
        Ks = (rows_ok && tail_mode) ? 80 : K;
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

    // Large device buffers start on 64 MiB boundaries (hipMalloc hands out 2 MiB-aligned blocks).
    // Which REGION of the device memory an array lies in is what the kernels with two concurrent
    // write streams feel (api_placement.inc) -- the alignment is kept from round 4 because it costs
    // 64 MiB per large buffer and never measured slower, not because it places anything.
    std::vector<std::pair<void *, void *>> skew_reg;   // (pointer handed out, base of its allocation)
    void big_alloc(void **p, size_t bytes) {
        // a candidate an earlier placement search set aside (api_placement.inc) serves first
        for (size_t i = 0; i < spare_big.size(); ++i)
            if (spare_big[i].second >= bytes && spare_big[i].second <= bytes + bytes / 8) {
                *p = spare_big[i].first;
                spare_big.erase(spare_big.begin() + i);
                spare_vs.erase(spare_vs.begin() + i);
                return;
            }
        big_alloc_fresh(p, bytes);
    }
    void big_alloc_fresh(void **p, size_t bytes) {
        if (bytes >= ((size_t)64 << 20)) {
            const size_t al = (size_t)kAllocAlignMb << 20;
            void *base = nullptr;
            SA_HIP(hipMalloc(&base, bytes + al));
            const uintptr_t b = reinterpret_cast<uintptr_t>(base);
            *p = reinterpret_cast<void *>((b + al - 1) / al * al);
            skew_reg.emplace_back(*p, base);
            return;
        }
        SA_HIP(hipMalloc(p, bytes));
    }
    void big_free(void *p) {
        if (!p) return;
        for (auto &e : skew_reg)
            if (e.first == p) {
                (void)hipFree(e.second);
                e.first = nullptr;
                return;
            }
        (void)hipFree(p);
    }

    ~Csc() override {
        (void)hipSetDevice(device);
        (void)hipStreamSynchronize(st);
        place_release_spares();
        big_free(gemit);
        if (part_vpost) (void)hipFree(part_vpost);
        big_free(cols_out[0]);
        big_free(cols_out[1]);
        for (hipEvent_t e : place_ev)
            if (e) (void)hipEventDestroy(e);
        for (auto &v : vars) big_free(v);
        big_free(y_alt);
        big_free(u_alt);
        big_free(work);
        y_alt = u_alt = nullptr;
        work = nullptr;
        for (void *p : {(void *)pst_part_rows, (void *)pst_part_f, pst_blk, (void *)pst_ctl, (void *)pst_bar,
                        (void *)cns_w, (void *)cns_sft, (void *)flt_h, (void *)flt_w,
                        (void *)zf_ch, (void *)zfv_buf, (void *)md_sft, (void *)md_coef, (void *)pgm_rx[0], (void *)pgm_rx[1]})
            if (p) (void)hipFree(p);
        for (void *p : {(void *)dft, (void *)sft, (void *)gramt, (void *)part_f, (void *)twA, (void *)twB,
                        (void *)twRows, (void *)part_rows, (void *)y_alt, (void *)u_alt, (void *)part_pgm, (void *)part_pgm2, (void *)ccmod_r, (void *)pgm_ey, (void *)gpart,
                        (void *)qpart, (void *)coop_flags, (void *)ghh, (void *)ghw, (void *)wg, (void *)g1t, (void *)ism_gam, (void *)ism_del, (void *)ism_mm, (void *)dft_mc, (void *)sft_mc, (void *)bt_mc, (void *)cns_f, (void *)cns_m, (void *)sft_eff, (void *)coef_t, (void *)ams_bits, (void *)gramz_t,
                        (void *)cns_yold, (void *)md_s, (void *)dism_gam, (void *)dism_del, (void *)dism_mm,
                        (void *)dwork, (void *)pcn_stats, (void *)work, (void *)innerb, (void *)gram, (void *)dpad, (void *)sreal,
                        (void *)wl1_buf, (void *)wl21_buf, (void *)wams_buf, (void *)wdat_buf, (void *)part_a, (void *)part_b, (void *)part_cs, (void *)df_im, (void *)pgm_es[0], (void *)pgm_es[1], (void *)pgm_es[2], (void *)pgm_es[3],
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
        if (depth > 1) planD.destroy();
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

    // (the Xf buffer also holds the tile-major spectrum, whose rows may be padded)
    size_t var_alloc_bytes(int var) const {
        return var == SPORCO_AMD_VAR_XF ? sizeof(cx<T>) * (size_t)std::max(EF, EFt) : var_bytes(var);
    }
    void *var_ptr(int var) {
        SA_REQUIRE(var_is_valid(var), "unknown state variable id");
        if (var == SPORCO_AMD_VAR_Y || var == SPORCO_AMD_VAR_U) {
            ++touch_epoch;
            if (v_live || gv_live) ensure_yu();
        }
        if (!vars[var]) {
            const size_t nb = var_alloc_bytes(var);
            big_alloc(&vars[var], nb);
            SA_HIP(hipMemsetAsync(vars[var], 0, nb, st));
        }
        return vars[var];
    }
    T *rv(int var) { return static_cast<T *>(var_ptr(var)); }
    cx<T> *cv(int var) { return static_cast<cx<T> *>(var_ptr(var)); }
    cx<T> *work_buf() {
        if (!work) big_alloc((void **)&work, sizeof(cx<T>) * std::max(EF, EFt));
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
    // (a mixed-radix handle: the FISTA and tile-major dictionary-update column kernels exist at its
    // height too, csc_pgm_mr.hip)
    bool pgm_fused_ok() const {
        return rows_ok && (cols256 || mr) && (fused || (fused_slabs && !tail_mode && !mr));
    }
    bool hint_vform = false, hint_one_launch = false;
    // SPORCO_AMD_MODE_COMPLEX_PAIR: the two channels of the handle are the real and the imaginary
    // part of complex data (dictionary updates only; csc_kernels.h launch_pm_butterfly)
    bool cplx_pair = false;
    void set_hint(int what, int value) override {
        if (what == SPORCO_AMD_HINT_KEEP_VFORM) hint_vform = value != 0;
        else if (what == SPORCO_AMD_HINT_ONE_LAUNCH) hint_one_launch = value != 0;
        else if (what == SPORCO_AMD_MODE_COMPLEX_PAIR) {
            SA_REQUIRE(Cd == 2 && Cs == 2, "complex pair mode: a handle with two channels (real, imaginary)");
            SA_REQUIRE(!have_signal && !zf_ch, "complex pair mode is set before the signal and the coefficient maps");
            cplx_pair = value != 0;
        } else if (what == SPORCO_AMD_VOLUME_FILTER_DEPTH) {
            SA_REQUIRE(depth > 1 && value >= 1 && value <= depth, "filter depth: a volume handle, 1 <= value <= depth");
            SA_REQUIRE(!flt_h, "volume handle: one filter support (no multi-scale dictionary)");
            vol_dD = value;
        } else throw Error(SPORCO_AMD_EINVAL, "unknown hint");
    }
    // (A, B) -> (A + iB, A - iB) and back on an array (npix, mid, 2, inner) of such a handle
    void pm_pair(const cx<T> *src, cx<T> *dst, int mid, int inner, int mode) {
        ProfScope ps(prof, PS_OTHER);
        launch_pm_butterfly<T>(st, src, dst, npix, mid, inner, W, mode);
    }
    std::string placement() override { return placement_report(); }
    // the second pair of iterate buffers (the (Y, U) ping-pong, and the two V buffers of the
    // single-array state): each clear of the spectrum buffer it is written together with by the
    // emitting row epilogue (api_placement.inc)
    // Striped output of the column pass (csc_fused.h FusedColsArgs::out_even / out_odd): two
    // half-sized spectra in different regions of the device memory, for the fused iteration of large
    // single-slab problems (the arrays the placement search serves).
    cx<T> *cols_out[2] = {nullptr, nullptr};
    bool cols_striped() const { return cols_out[0] != nullptr; }
    void alloc_cols_out() {
        if (cols_out[0] || !std::is_same<T, float>::value || !fused || fused_slabs || tail_mode || Ks != K || !rows_ok || mr)
            return;
        const size_t plane = sizeof(cx<T>) * (size_t)CN * H * K;
        const size_t be = plane * (size_t)((Wf + 1) / 2), bo = plane * (size_t)(Wf / 2);
        if (!placement_on(be)) return;
        void *p0 = nullptr;
        big_alloc(&p0, be);
        cols_out[0] = static_cast<cx<T> *>(p0);
        cols_out[1] = static_cast<cx<T> *>(place_alloc(bo, {{p0, be}}, "cols_out_odd"));
    }
    void alloc_alt_pair() {
        if (y_alt) return;
        (void)var_ptr(SPORCO_AMD_VAR_XF);
        // The iterate goes round all four state buffers -- the (Y, U) ping-pong swaps the pairs, and
        // the V buffers of the single-array state are whichever pair is "alt" when a run enters it
        // -- so the spectrum buffer is (moved) clear of the two that exist, and the two new ones
        // are taken clear of it.
        place_var(SPORCO_AMD_VAR_XF, {SPORCO_AMD_VAR_Y, SPORCO_AMD_VAR_U}, "T");
        void *t = vars[SPORCO_AMD_VAR_XF];
        const size_t tb = var_alloc_bytes(SPORCO_AMD_VAR_XF), nb = sizeof(T) * (size_t)E;
        y_alt = static_cast<T *>(place_alloc(nb, {{t, tb}}, "V0"));
        u_alt = static_cast<T *>(place_alloc(nb, {{t, tb}}, "V1"));
        // (the spectrum buffer could not be moved clear of Y / U -- a long region: they move instead,
        // to where the search has arrived by now; no-ops when they are clear already)
        place_var(SPORCO_AMD_VAR_Y, {SPORCO_AMD_VAR_XF}, "Y");
        place_var(SPORCO_AMD_VAR_U, {SPORCO_AMD_VAR_XF}, "U");
        alloc_cols_out();      // (after the decisions above: its candidates must not disturb them)
        place_release_spares();   // (what no decision took goes back: up to 8 GiB otherwise idle)
    }
    int query(int what) override {
        if (what == SPORCO_AMD_QUERY_FUSED_COLS) return (fused || fused_slabs || fused_mc) ? 1 : 0;
        if (what == SPORCO_AMD_QUERY_FUSED_ROWS) return rows_ok ? 1 : 0;
        if (what == SPORCO_AMD_QUERY_FUSED_PGM) return pgm_fused_ok() ? 1 : 0;
        if (what == SPORCO_AMD_QUERY_DEVICE_FILTERS) return K;
        if (what == SPORCO_AMD_QUERY_VFORM_LIVE) return (v_live || gv_live) ? 1 : 0;
        if (what == SPORCO_AMD_QUERY_PERSIST_RUNS) return pst_runs;
        if (what == SPORCO_AMD_QUERY_CCMOD_GROUPS) return ccmod_group_count();
        throw Error(SPORCO_AMD_EINVAL, "unknown query");
    }

#include "api_placement.inc"
#include "api_transforms.inc"
#include "api_setup.inc"
#include "api_admm_run.inc"
#include "api_admm.inc"
#include "api_pgm.inc"
#include "api_dictupdate.inc"
#include "api_consensus.inc"
#include "api_maskdcpl.inc"
#include "api_dstep.inc"
};

CscBase *make_csc(const sporco_amd_dims &dims, int dict_channels, int device, void *stream, int depth) {
    if (dims.dtype == SPORCO_AMD_F32) return new Csc<float>(dims, device, stream, dict_channels, depth);
    if (dims.dtype == SPORCO_AMD_F64) return new Csc<double>(dims, device, stream, dict_channels, depth);
    throw Error(SPORCO_AMD_EINVAL, "dtype must be SPORCO_AMD_F32 or SPORCO_AMD_F64");
}

}  // namespace sporco_amd
