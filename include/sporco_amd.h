/*
 * sporco_amd.h -- C ABI of libsporco_amd.so, the MI355X (gfx950) backend for
 * SPORCO's FFT-domain convolutional sparse coding path.
 *
 * The reference (bwohlberg/sporco) is pure Python and has no FFI of its own
 * (SURVEY.md section 8(b)); its "operator API" is the Python class API.  This
 * header is therefore the binding surface that a reference-side backend shim
 * would load with ctypes (INTEGRATION.md shows that shim).  Every entry point
 * cites the reference function(s) whose arithmetic it replaces, as
 * `sporco/<file>:<lines>` relative to the reference root.
 *
 * Conventions
 *   - plain C: opaque handle, pointers and sizes only, no C++/torch types;
 *   - every function returns 0 on success, a negative code on failure; the
 *     message is available from sporco_amd_last_error() (thread local);
 *   - pointers are HOST pointers unless the parameter name ends in `_dev`;
 *   - arrays use the reference's internal C-contiguous layout
 *     (H, W, C, N, K), filter index K fastest (sporco/cnvrep.py:86-111;
 *     BASELINE letters: N = images = SPORCO `K`, K = filters = SPORCO `M`);
 *     frequency-domain arrays are (H, W/2+1, C, N, K) interleaved complex;
 *   - `dtype` is SPORCO_AMD_F32 (float / complex64) or SPORCO_AMD_F64;
 *   - a handle owns one HIP stream (or borrows the one passed at creation);
 *     calls on one handle must not be issued concurrently from two threads.
 *
 * Environment switches of the library (15).  None is needed in normal use: they pin one code path
 * against another in the tests and in A/B measurements.  All are read ONCE, when a handle is made
 * (csc_api.hip struct Switches) -- except HOST_LOOP and RUN_LAG, which callers flip between runs of
 * one handle and sporco_amd_csc_admm_run reads at its entry -- and RCCL_LIB, read when RCCL is first
 * opened.  Launch forms, staggers and the variants of reverted experiments are constants of the
 * sources since round 5, not switches.
 *   SPORCO_AMD_UNFUSED=1        generic kernel chain (line FFTs + streaming kernels) for every shape
 *   SPORCO_AMD_OLD_ROWS=1       generic row passes around the register-resident column kernel
 *   SPORCO_AMD_NO_PAD=1         no zero filter appended to an odd filter count
 *   SPORCO_AMD_NO_VFORM=1       keep the ADMM iterate as (Y, U): no single-array state (csc_rows.h)
 *   SPORCO_AMD_NO_SPECULATION=1 the row epilogue never emits the next iteration's row spectra
 *   SPORCO_AMD_HOST_LOOP=1      one sporco_amd_csc_admm_iter per iteration, no sporco_amd_csc_admm_run
 *   SPORCO_AMD_RUN_LAG=n        (tests) admm_run pretends not to have seen its newest n records
 *   SPORCO_AMD_PERSIST=1|0      small problems: a run of iterations as ONE launch (off unless the
 *                               handle has SPORCO_AMD_HINT_ONE_LAUNCH; see there)
 *   SPORCO_AMD_NO_COLS_SM=1     generic chain: the column pass as three kernels, not one (fft.h)
 *   SPORCO_AMD_COLS_SM_FORCE_SLAB=k   (tests) ... in slabs of k filters even where the tile fits
 *   SPORCO_AMD_MD_GENERIC=1     ConvBPDNMaskDcpl on the generic chain
 *   SPORCO_AMD_CNS_GENERIC=1    consensus dictionary update on the generic chain
 *   SPORCO_AMD_CG_HOST=1        CG dictionary update: the scalars read back every iteration
 *   SPORCO_AMD_PLACEMENT=0|force   no placement search for concurrently written arrays / the search
 *                               (and the striped column output) also for small arrays -- tests
 *                               (sporco_amd_csc_placement_report)
 *   SPORCO_AMD_RCCL_LIB=path    the RCCL library to open (default librccl.so.1, librccl.so)
 * Outside the library: SPORCO_AMD_LIBRARY=path and SPORCO_AMD_SHARE_TORCH_RUNTIME=0 (the Python
 * binding: which build to load / do not pre-load torch's HIP runtime), SPORCO_AMD_BENCH_* (bench.py).
 */
#ifndef SPORCO_AMD_H
#define SPORCO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPORCO_AMD_F32 0
#define SPORCO_AMD_F64 1

/* error codes */
#define SPORCO_AMD_OK 0
#define SPORCO_AMD_EINVAL (-1)   /* bad argument / unsupported size        */
#define SPORCO_AMD_EHIP (-2)     /* HIP runtime error (message has detail) */
#define SPORCO_AMD_ENOMEM (-3)
#define SPORCO_AMD_ESTATE (-4)   /* call order violated (e.g. no dictionary set) */

/* ---- library-level ------------------------------------------------------ */

const char *sporco_amd_version(void);
const char *sporco_amd_last_error(void);
/* Number of visible HIP devices (0 when none: product code must then fail). */
int sporco_amd_device_count(int *count);
/* Device name and compute-unit count of `device`. */
int sporco_amd_device_info(int device, char *name, size_t name_len, int *cu_count,
                           size_t *hbm_bytes);

/* ---- solver handle ------------------------------------------------------ */

typedef struct sporco_amd_csc *sporco_amd_csc_t;

typedef struct {
    int32_t H, W;   /* spatial size Nv (sporco/cnvrep.py:186)                  */
    int32_t C;      /* channels of S and of X; dictionary is single channel   */
    int32_t N;      /* images,  SPORCO cri.K (cnvrep.py:176-180)              */
    int32_t K;      /* filters, SPORCO cri.M (cnvrep.py:183)                  */
    int32_t dtype;  /* SPORCO_AMD_F32 | SPORCO_AMD_F64                        */
} sporco_amd_dims;

/* State arrays addressable through upload/download. */
#define SPORCO_AMD_VAR_Y 0     /* real  (H,W,C,N,K)   ADMM auxiliary variable        */
#define SPORCO_AMD_VAR_U 1     /* real  (H,W,C,N,K)   ADMM scaled dual variable       */
#define SPORCO_AMD_VAR_X 2     /* real  (H,W,C,N,K)   primal variable                 */
#define SPORCO_AMD_VAR_XF 3    /* cplx  (H,Wf,C,N,K)  rfftn(X)                        */
#define SPORCO_AMD_VAR_DF 4    /* cplx  (H,Wf,1,1,K)  rfftn(D zero-padded)            */
#define SPORCO_AMD_VAR_SF 5    /* cplx  (H,Wf,C,N,1)  rfftn(S)                        */
#define SPORCO_AMD_VAR_YF 6    /* cplx  (H,Wf,C,N,K)  PGM auxiliary state Yf          */
#define SPORCO_AMD_VAR_XFPRV 7 /* cplx  PGM previous Xf                               */
#define SPORCO_AMD_VAR_YFPRV 8 /* cplx  PGM previous Yf                               */
#define SPORCO_AMD_VAR_VF 9    /* cplx  PGM gradient-step buffer Vf / generic scratch */
#define SPORCO_AMD_VAR_GF 10   /* cplx  PGM gradient buffer                           */
#define SPORCO_AMD_VAR_AX 11   /* real  relaxed AX of the staged ADMM path            */
#define SPORCO_AMD_VAR_YPREV 12 /* real Y of the previous iteration (staged ADMM path) */
#define SPORCO_AMD_VAR_T0 13   /* cplx  scratch state for host-composed policies        */
#define SPORCO_AMD_VAR_T1 14   /* cplx  (BB step size: previous x / gradient;           */
#define SPORCO_AMD_VAR_T2 15   /* cplx   robust backtracking: Z; monotone FISTA: ZZ)    */
#define SPORCO_AMD_VAR_ZF 16   /* cplx  (H,Wf,C,N,K) rfftn of the coefficient maps (D-step)   */
#define SPORCO_AMD_VAR_CX 17   /* real  (H,W,C,N,K) consensus D-step: dictionary copy per image */
#define SPORCO_AMD_VAR_CU 18   /* real  (H,W,C,N,K) consensus D-step: scaled dual per image     */
#define SPORCO_AMD_VAR_MY0 19  /* real  (H,W,C,N,1) ConvBPDNMaskDcpl: block 0 of Y (masked residual) */
#define SPORCO_AMD_VAR_MU0 20  /* real  (H,W,C,N,1) ConvBPDNMaskDcpl: block 0 of U                 */
#define SPORCO_AMD_VAR_DMY0 21 /* real  (H,W,C,N,1) mask-decoupling D-step: block 0 of Y           */
#define SPORCO_AMD_VAR_DMU0 22 /* real  (H,W,C,N,1) mask-decoupling D-step: block 0 of U           */
/* Dictionary-sized state of the D-step (pgm.ccmod.ConvCnstrMOD, admm.ccmod consensus Y =
 * DX): real (H,W,K) / complex (H,Wf,K).  Ids 19..31 are reserved. */
#define SPORCO_AMD_VAR_DX 32      /* real  dictionary iterate X (zero-padded filters)    */
#define SPORCO_AMD_VAR_DXF 33     /* cplx  rfftn(DX)                                     */
#define SPORCO_AMD_VAR_DYF 34     /* cplx  auxiliary (momentum) state                    */
#define SPORCO_AMD_VAR_DXFPRV 35  /* cplx  previous DXF                                  */
#define SPORCO_AMD_VAR_DYFPRV 36  /* cplx  previous DYF                                  */
#define SPORCO_AMD_VAR_DVF 37     /* cplx  gradient-step buffer                          */
#define SPORCO_AMD_VAR_DGF 38     /* cplx  gradient                                      */
#define SPORCO_AMD_VAR_DT0 39     /* cplx  scratch for host-composed policies            */
#define SPORCO_AMD_VAR_DT1 40
#define SPORCO_AMD_VAR_DT2 41
#define SPORCO_AMD_VAR_DSX 42     /* real  X of the single-copy ADMM D-step (dstep_iter)   */
#define SPORCO_AMD_VAR_DSU 43     /* real  its scaled dual variable U                      */
#define SPORCO_AMD_VAR_COUNT 44

/* Create a solver on HIP device `device`.  `stream` is a hipStream_t to borrow
 * (e.g. torch.cuda.current_stream().cuda_stream) or NULL to own a new one.
 * Replaces the allocation half of GenericConvBPDN.__init__
 * (sporco/admm/cbpdn.py:224-236) and PGM ConvBPDN.__init__
 * (sporco/pgm/cbpdn.py:216-233). */
int sporco_amd_csc_create(const sporco_amd_dims *dims, int device, void *stream,
                          sporco_amd_csc_t *out);
/* The same for a multi-channel dictionary (SPORCO cri.Cd > 1, sporco/cnvrep.py:186-194):
 * dict_channels == dims->C channels in D (host layout (dH,dW,Cd,K)) and in S; the coefficient
 * arrays then have a single channel, (H,W,1,N,K), Df is (H,Wf,Cd,1,K), and the X-step is the
 * iterated Sherman-Morrison solve linalg.solvemdbi_ism (sporco/linalg.py:370-444, call site
 * sporco/admm/cbpdn.py:277-279).  ADMM ConvBPDN only; dict_channels == 1 is
 * sporco_amd_csc_create. */
int sporco_amd_csc_create_mc(const sporco_amd_dims *dims, int32_t dict_channels, int device,
                             void *stream, sporco_amd_csc_t *out);
/* Volumes -- three spatial axes, dimN = 3 of sporco/cnvrep.py:33-198 (the constructor contract of
 * sporco/admm/cbpdn.py:175; the reference's examples/scripts/cdl/cbpdndl_video.py:74 is the use):
 * arrays (depth, height, W, C, N, K) are handed over as they lie in memory with dims->H =
 * depth * height, and `depth` tells the handle where the folded axis splits.  Everything per
 * pixel / frequency / row is unchanged on the folded array; the transform along the folded axis
 * runs as two passes.  Single-channel dictionary, given zero-padded to the full volume
 * (sporco_amd_csc_set_dict with dH = dims->H, dW = dims->W).  Such a handle serves the ADMM sparse
 * coding calls (set_signal, set_dict, set_l1_weight, upload / download, admm_iter / _run and the
 * staged steps, the staged PGM steps, reconstruct, dhs_absmax, asum) without NoBndryCross /
 * gradient term / AddMaskSim, and the PGM and consensus dictionary updates (ccmod_setcoef, _grad,
 * _eval, _prox_step, _cnstr, setdict_from_dstep, cns_init, cns_iter without mask decoupling;
 * SPORCO_AMD_VOLUME_FILTER_DEPTH below); every other
 * entry point returns SPORCO_AMD_EINVAL for it.  Generic transform chain. */
int sporco_amd_csc_create_volume(const sporco_amd_dims *dims, int32_t depth, int device, void *stream,
                                 sporco_amd_csc_t *out);
int sporco_amd_csc_destroy(sporco_amd_csc_t h);
int sporco_amd_csc_sync(sporco_amd_csc_t h);
/* The hipStream_t every launch of this handle goes to (the `stream` given at creation, or the
 * stream the handle created for itself): a caller that enqueues its own work between two
 * calls -- the all-reduce hook of sporco_amd_csc_admm_run under torch.distributed -- orders
 * it on this stream (no reference counterpart: the reference is synchronous NumPy). */
int sporco_amd_csc_stream(sporco_amd_csc_t h, void **stream);
/* Which kernels serve this handle's shape: *out = 1 when the fused path named by
 * `what` is active (float32, H and/or W in {128, 256, 512}, even K <= 64 -- or even
 * 64 < K <= 256, where the column pass runs as cooperating 64-filter slab workgroups; since
 * round 6 also H, W in 16 x {10, 12, 14, 15, 18, 20, 21, 24, 25, 27, 28, 30}, K <= 256, for the
 * ADMM ConvBPDN / Joint / GradReg / AddMaskSim / mask-decoupling calls, the fused PGM iteration and
 * the tile-major dictionary-update gradient with K <= 64: such a handle serves LinSolveCheck,
 * multi-channel dictionaries and the consensus update on its generic chain), else 0. */
#define SPORCO_AMD_QUERY_FUSED_COLS 0  /* register-resident column FFT + Sherman-Morrison */
#define SPORCO_AMD_QUERY_FUSED_ROWS 1  /* three-launch ADMM iteration                      */
#define SPORCO_AMD_QUERY_FUSED_PGM 2   /* fused PGM iteration / tile-major D-step               */
#define SPORCO_AMD_QUERY_DEVICE_FILTERS 3 /* filter count of the device-resident arrays: K, or
                                            K + 1 when an odd K was padded with one all-zero
                                            filter to reach the fused kernels (host arrays
                                            always have K; only sporco_amd_csc_device_ptr
                                            exposes the padded layout) */
#define SPORCO_AMD_QUERY_VFORM_LIVE 4   /* 1 while the ADMM iterate lives in the single-array
                                            form of the fused iteration (V = AX + U; Y and U
                                            are derived on the next access) -- diagnostics */
#define SPORCO_AMD_QUERY_PERSIST_RUNS 5  /* how many sporco_amd_csc_admm_run calls of this handle
                                            ran their iterations as one launch (small problems:
                                            see sporco_amd_csc_admm_run) -- diagnostics */
#define SPORCO_AMD_QUERY_CCMOD_GROUPS 6  /* image groups per row frequency of the tile-major
                                          * dictionary-update gradient (their partial gradients
                                          * are written and summed: bench.py's byte model) */
int sporco_amd_csc_query(sporco_amd_csc_t h, int what, int *out);
/* Diagnostics, no reference counterpart: where the handle put the X-sized arrays that one kernel
 * writes at the same time (the spectrum buffer T and the iterate buffers of the fused ADMM
 * iteration).  Two arrays that take streaming stores concurrently share ~4.7 TB/s on the MI355X
 * when their physical memory lies in the same region of the device memory and get ~6.3 TB/s when
 * it does not (tools/ubench/place_probe.hip, profiles/r05_placement_notes.md), so the handle takes
 * such a buffer from a few candidate allocations, each timed against the arrays it must keep
 * clear of.  The report is a JSON list, one object per decision: role, bytes, candidates tried,
 * the same-region reference rate, probe rate / reference of the first and of the chosen candidate
 * (>= 1.09: clear).  Arrays below 256 MiB are placed as they come (empty list);
 * SPORCO_AMD_PLACEMENT=0 in the environment switches the search off. */
int sporco_amd_csc_placement_report(sporco_amd_csc_t h, char *buf, size_t cap);
/* Hints about how the handle will be used (never needed for correctness; no reference
 * counterpart).  KEEP_VFORM: the caller alternates short device-driven runs with
 * sporco_amd_csc_ccmod_setcoef(VAR_Y) and does not read Y or U in between -- the loop of
 * dictlrn.DictLearn.solve (sporco/dictlrn/dictlrn.py:319-375) -- so even a one-iteration
 * sporco_amd_csc_admm_run keeps the iterate in its single-array form and setcoef derives Y from
 * it on the way into its row transform. */
#define SPORCO_AMD_HINT_KEEP_VFORM 0
/* ONE_LAUNCH: this process has the device to itself and the caller accepts iterates that equal
 * the default loop's to float32 rounding (not bit for bit): sporco_amd_csc_admm_run then runs a
 * small problem (float32, square images of 128 or 256 pixels, K <= 64, at most 4 Mi coefficients,
 * plain ConvBPDN options: scalar weights, NonNegCoef) as ONE kernel launch whose workgroups wait
 * for each other between the passes of an iteration.  9-12 % faster at 256 x 256, K = 32, N = 1
 * (profiles/r03_persist.md).  A wait that cannot complete -- the device was shared after all --
 * ends the call with SPORCO_AMD_EHIP after a few seconds; the handle's iterate is then void.
 * The environment variable SPORCO_AMD_PERSIST=1 / 0 overrides the hint. */
#define SPORCO_AMD_HINT_ONE_LAUNCH 1
/* COMPLEX_PAIR -- not a hint: it changes what the dictionary-update calls compute.  The handle
 * (two-channel dictionary, Cd = 2; set before sporco_amd_csc_set_signal) serves complex-valued
 * signals, coefficient maps and dictionary, whose real and imaginary parts are its two channels:
 * sporco_amd_csc_dstep_iter / _cns_iter then solve the complex problem of
 * sporco/admm/ccmod.py:103-907 given complex input (its fftn / ifftn path, :219-231; the
 * reference's tests/admm/test_ccmod.py:49-140) with the coefficient maps staged per channel
 * (sporco_amd_csc_ccmod_setcoef(VAR_CX)).  Spectral arrays of such a handle hold A + iB and
 * A - iB (A, B the half spectra of the two parts) scaled as csrc/csc_kernels.h
 * launch_pm_butterfly describes; the real arrays (VAR_DX, VAR_DSX, VAR_DSU, VAR_CX, VAR_CU) are
 * (re, im) channel pairs.  Not with mask decoupling, image shards or the objective at X. */
#define SPORCO_AMD_MODE_COMPLEX_PAIR 2
/* VOLUME_FILTER_DEPTH -- a setting, for a volume handle (sporco_amd_csc_create_volume): how many
 * depth slabs the filter support spans.  The dictionary-update calls that take a support (dH, dW) --
 * sporco_amd_csc_ccmod_prox_step, _ccmod_cnstr, _cns_iter -- then crop to (value, dH, dW) (cnvrep.bcrop with
 * dimN = 3, sporco/cnvrep.py:553-606). */
#define SPORCO_AMD_VOLUME_FILTER_DEPTH 3
int sporco_amd_csc_set_hint(sporco_amd_csc_t h, int what, int value);

/* S: real (H,W,C,N) in the handle dtype.  Computes Sf = rfftn(S, axes=(0,1))
 * on device -- sporco/admm/cbpdn.py:228-231, sporco/fft.py:257-286. */
int sporco_amd_csc_set_signal(sporco_amd_csc_t h, const void *S);

/* D: real (dH,dW,K) filters.  Zero-pads to (H,W), Df = rfftn, and caches the
 * per-pixel Sherman-Morrison denominator sum_k |Df|^2 -- setdict,
 * sporco/admm/cbpdn.py:242-256 (DSf is never materialised), and
 * sporco/pgm/cbpdn.py:239-245. */
int sporco_amd_csc_set_dict(sporco_amd_csc_t h, const void *D, int32_t dH, int32_t dW);
/* Complex-valued signals and dictionaries (sporco/admm/cbpdn.py:209-217: real_dtype False, fftn /
 * ifftn in place of rfftn / irfftn; test tests/admm/test_cbpdn.py:179-201).  D_imag: the imaginary
 * part of the dictionary, (dH,dW,K) like D of set_dict, which then holds its real part.  The handle
 * is created with C = 2 Cc channels: the signal passed to set_signal is (Re S, Im S) along the
 * channel axis, and so is every coefficient array -- every transform stays a real one, and the
 * X step pairs the channel halves (A + iB at frequency f, A - iB = the conjugate at -f; two
 * Sherman-Morrison solves per stored frequency, linalg.solvedbi_sm).  The complex soft threshold
 * of the y step is the l2 shrinkage over the pair: callers run FLAG_JOINT with lmbda = 0 and
 * mu = lambda.  Generic kernel chain, single-channel dictionary, no FLAG_XRRS / FLAG_GRADREG.
 * D_imag == NULL returns the handle to a real dictionary. */
int sporco_amd_csc_set_dict_imag(sporco_amd_csc_t h, const void *D_imag, int32_t dH, int32_t dW);

/* l1 weight array (L1Weight option, sporco/admm/cbpdn.py:596-597 after
 * cnvrep.l1Wshape, sporco/cnvrep.py:492-550).  shape[d] is 1 (broadcast) or the
 * full extent of (H,W,C,N,K)[d]; w == NULL restores the scalar weight 1. */
int sporco_amd_csc_set_l1_weight(sporco_amd_csc_t h, const void *w, const int64_t shape[5]);
/* l2,1 weight (L21Weight, sporco/admm/cbpdn.py:781): broadcastable against
 * (H,W,1,N,K); shape[2] must be 1. */
int sporco_amd_csc_set_l21_weight(sporco_amd_csc_t h, const void *w, const int64_t shape[5]);
/* Per-filter weights of the gradient penalty (GradWeight option of ConvBPDNGradReg,
 * sporco/admm/cbpdn.py:1063-1071, :1134-1139): K values of the handle's dtype;
 * w == NULL restores the scalar weight 1. */
int sporco_amd_csc_set_grad_weight(sporco_amd_csc_t h, const void *w);
/* Multi-scale dictionary (a `dsz` made of size blocks, sporco/cnvrep.py:729-812: e.g.
 * ((8, 8, 32), (12, 12, 32), (16, 16, 32))): the support (rows fh[k], columns fw[k]) of each of
 * the K filters for every constraint projection this handle makes (cnvrep.Pcn: crop, zero mean
 * and norm over the filter's own support, cnvrep.py:634-662, :868-913).  The dH / dW arguments of
 * the dictionary-update calls then name the largest support (what bcrop returns).  NULL, NULL
 * returns to one support for all filters. */
int sporco_amd_csc_set_filter_sizes(sporco_amd_csc_t h, const int32_t *fh, const int32_t *fw);
/* Mask of the additive-mask-simulation wrapper (AddMaskSim, sporco/admm/cbpdn.py:2287-2485):
 * broadcastable against (H,W,C,N,1), shape[4] must be 1.  With SPORCO_AMD_FLAG_AMS the last
 * filter of the dictionary is taken to be the appended impulse (:2345-2353); its slice of Y
 * is AX + U zeroed where the mask is nonzero (:2378-2394) and is left out of the l1 / l2,1
 * sums (:2398-2412).  w == NULL removes the mask.
 * Handles made by sporco_amd_csc_create_mc (multi-channel dictionary, Cd channels): the last Cd
 * filters are the appended impulses, one per channel (:2339-2346), and the mask is
 * broadcastable against (H,W,1,N,Cd) -- shape[4] is 1 or Cd, the mask's channels on the filter
 * axis as the reference keeps them (:2358-2364). */
int sporco_amd_csc_set_ams_mask(sporco_amd_csc_t h, const void *w, const int64_t shape[5]);

/* Host <-> device transfer of one state array in the reference layout. */
int sporco_amd_csc_upload(sporco_amd_csc_t h, int var, const void *src);
int sporco_amd_csc_download(sporco_amd_csc_t h, int var, void *dst);
/* Raw device pointer of a state array (plumbing for torch.distributed / tests).
 * LIFETIME: the pointer is valid until the next call on this handle that iterates or changes state
 * (sporco_amd_csc_admm_iter / _admm_run / _pgm_iter / the dictionary-update steps, set_dict,
 * set_signal, upload): the iterates ping-pong between buffers, and the first fused iteration of a
 * handle may MOVE the variable -- the arrays a kernel writes at the same time are placed by
 * measurement then (sporco_amd_csc_placement_report) and the old buffer is freed.  Ask again after
 * such a call; do not cache the value across it. */
int sporco_amd_csc_device_ptr(sporco_amd_csc_t h, int var, void **ptr_dev);

/* ---- ADMM iteration (sporco/admm/admm.py:331-367 loop body) -------------- */

#define SPORCO_AMD_FLAG_NONNEG (1u << 0)     /* NonNegCoef    cbpdn.py:306-307 */
#define SPORCO_AMD_FLAG_NOBNDRY (1u << 1)    /* NoBndryCross  cbpdn.py:308-311 */
#define SPORCO_AMD_FLAG_JOINT (1u << 2)      /* ConvBPDNJoint ystep cbpdn.py:785-794 */
#define SPORCO_AMD_FLAG_RESID (1u << 3)      /* fill residual sums out[0..4]   */
#define SPORCO_AMD_FLAG_OBJ (1u << 4)        /* fill objective sums out[5..7]  */
#define SPORCO_AMD_FLAG_XRRS (1u << 5)       /* LinSolveCheck sums out[8..10]  */
#define SPORCO_AMD_FLAG_GEVAL_Y (1u << 6)    /* regularisers evaluated at Y (gEvalY) */
#define SPORCO_AMD_FLAG_FEVAL_Y (1u << 7)    /* data fidelity evaluated at Y (fEvalX False) */
#define SPORCO_AMD_FLAG_KEEP_X (1u << 8)     /* write X during the call (default: X, Xf are rebuilt
                                                on demand from the previous iterate, which
                                                the fused path keeps) */
#define SPORCO_AMD_FLAG_NO_X (1u << 9)       /* caller will not read X / Xf of this iteration:
                                                they are neither written nor recoverable, and
                                                reading them fails with SPORCO_AMD_ESTATE */
#define SPORCO_AMD_FLAG_GRADREG (1u << 10)   /* ConvBPDNGradReg xstep / objective
                                                (cbpdn.py:1163-1214): diagonal mu*GradWeight*GHGf
                                                + rho, params.mu = gradient weight mu; multi-channel
                                                dictionary: the iterated solve with that diagonal
                                                (:1181-1184) */
#define SPORCO_AMD_FLAG_AMS (1u << 11)       /* AddMaskSim y step / regulariser sums on the
                                                last filter slice (set_ams_mask) */
#define SPORCO_AMD_FLAG_DMASK (1u << 12)     /* pgm_iter: data fidelity (1/2)||W (D x - s)||^2 with
                                                the mask of set_data_mask (pgm.cbpdn.ConvBPDNMask,
                                                sporco/pgm/cbpdn.py:387-506): the residual goes
                                                through the spatial domain for W^2 between the
                                                inner product and the gradient; out[PGM_DFID] =
                                                sum (W R)^2 at the new Xf (want_stats; PGM_F and
                                                PGM_FY are not evaluated: masked_grad has them);
                                                K <= 64, no held trial */

typedef struct {
    double rho;      /* penalty parameter for this iteration                     */
    double lmbda;    /* l1 weight lambda                                          */
    double mu;       /* l2,1 weight (JOINT) / gradient penalty weight (GRADREG)   */
    double rlx;      /* RelaxParam alpha (sporco/admm/admm.py:877-885)            */
    double u_scale;  /* pending `U /= rsf` of update_rho (admm.py:573), applied
                        to U as it is read: U_true = u_scale * U_stored          */
    uint32_t flags;
    int32_t dH, dW;  /* filter support, for NOBNDRY                               */
} sporco_amd_admm_params;

/* Output slots of admm_iter (raw sums; the host forms r, s, eps exactly as
 * sporco/admm/admm.py:462-486 with ADMMEqual.rsdl_* :959-983). */
#define SPORCO_AMD_OUT_R2 0      /* sum (AXnr - Y)^2                               */
#define SPORCO_AMD_OUT_S2 1      /* sum (Y - Yprev)^2   (rho applied by host)      */
#define SPORCO_AMD_OUT_AX2 2     /* sum AXnr^2                                     */
#define SPORCO_AMD_OUT_Y2 3      /* sum Y^2                                        */
#define SPORCO_AMD_OUT_U2 4      /* sum U^2 (new U, before any rho rescale)        */
#define SPORCO_AMD_OUT_DFID 5    /* half-spectrum Parseval sum of |Df.Xf - Sf|^2 / (H W)
                                    (cbpdn.py:337-344, fft.py:449-484); host halves it */
#define SPORCO_AMD_OUT_L1 6      /* sum |wl1 * g-variable|   (cbpdn.py:624-630)    */
#define SPORCO_AMD_OUT_L21 7     /* sum wl21 * sqrt(sum_c g^2) (cbpdn.py:798-807)  */
#define SPORCO_AMD_OUT_XRRS_D2 8 /* sum |ax - b|^2 of the X-step system            */
#define SPORCO_AMD_OUT_XRRS_AX2 9
#define SPORCO_AMD_OUT_XRRS_B2 10
#define SPORCO_AMD_OUT_RGR 11    /* Parseval sum of GradWeight GHGf |Xf|^2 / (H W)  (twice RegGrad,
                                  * cbpdn.py:1204-1214; FLAG_GRADREG only)         */
#define SPORCO_AMD_OUT_CNSTR 12  /* sum (Pcn(Y) - Y)^2 of the consensus D-step (ccmod.py:888-894) */
#define SPORCO_AMD_OUT_CGIT 13   /* CG D-step: scipy's cg() status, 0 = converged, else MaxIter
                                    (what the reference records as XSlvCGIt, linalg.py:578) */
#define SPORCO_AMD_OUT_CGN 14    /* CG D-step: iterations actually run                     */
#define SPORCO_AMD_OUT_COUNT 16

/* One full ADMM iteration on device: xstep (cbpdn.py:267-281: rfftn(Y-U),
 * Sherman-Morrison solve linalg.py:232-297, irfftn), relax_AX, ystep
 * (prox_l1 prox/_lp.py:144-183 or prox_sl1l2 prox/_l21.py:51-88, NonNeg,
 * NoBndryCross), ustep (admm.py:434-437) and all reductions of
 * compute_residuals / eval_objfn.  Blocks until `out` is valid. */
int sporco_amd_csc_admm_iter(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                             double out[SPORCO_AMD_OUT_COUNT]);
/* Same, but leaves the sums in device memory at `out_dev` (double[16]) and does
 * not synchronise: used to all-reduce them over RCCL before reading. */
int sporco_amd_csc_admm_iter_dev(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                                 double *out_dev);

/* ---- Device-driven solve: the whole loop of ADMM.solve (sporco/admm/admm.py:331-377) ----
 * The residuals and tolerances (:462-486), the adaptive penalty parameter (:549-575) and the
 * stopping test (:375-377) are evaluated on the device at the end of every iteration, in the
 * precisions of the host code, and the next iteration's kernels read rho, lambda/rho and the
 * pending U scale from device memory: the host only enqueues launches (a few iterations
 * ahead) and reads the per-iteration records back when the run ends.  Iterates and statistics
 * are identical to those of the same number of sporco_amd_csc_admm_iter calls driven by the
 * host rule.  Available for the three-launch float32 path (query FUSED_ROWS; single-channel
 * dictionary; K <= 64, or 72 < K <= 256 on the slab column kernels -- not the 64 < K <= 72
 * tail form) without FLAG_XRRS / GRADREG / KEEP_X / FEVAL_Y; FLAG_JOINT is served when the
 * l2,1 row epilogue is (scalar weights, C <= 4, K a multiple of 32, no NoBndryCross /
 * AddMaskSim); otherwise the call returns SPORCO_AMD_EUNSUPPORTED and changes nothing.
 * With a `reduce` hook (image shards) the number of hook calls is a function of the
 * stopping iteration alone -- min(max_iter, stop + 1 + lookahead) -- so that every rank
 * issues the same number of collectives whatever its host's timing. */
#define SPORCO_AMD_EUNSUPPORTED (-5)
typedef struct {
    double abs_tol, rel_tol;        /* AbsStopTol, RelStopTol                                */
    double sqrt_nc, sqrt_nx;        /* sqrt of the constraint / primal sizes (admm.py:478-485) */
    double rho_tau, rho_mu, rho_xi; /* AutoRho Scaling, RsdlRatio, RsdlTarget (solver precision) */
    int32_t auto_rho, period, auto_scaling, std_residuals;
    int32_t need_residuals;         /* AutoRho enabled or not FastSolve                      */
    int32_t k0;                     /* iteration counter at entry (the solver's k)           */
    int32_t max_iter;               /* MaxMainIter                                            */
    int32_t lookahead;              /* iterations enqueued ahead of the last finished one; 0 = 3 */
} sporco_amd_admm_ctrl;
typedef struct {
    double sums[SPORCO_AMD_OUT_COUNT];
    double r, s, epri, edua;        /* residuals and tolerances of the iteration              */
    double rho, u_scale;            /* the values the iteration ran with                      */
    double seconds;                 /* device clock at the end of the iteration, from the start of the run */
    int32_t k, stop;
} sporco_amd_admm_record;
/* Optional hook between the local sums and the control update of every iteration: sum the 16
 * doubles at sums_dev over the ranks that hold the other images, in stream order (RCCL on the
 * handle's stream) or synchronously.  NULL for a single rank. */
typedef void (*sporco_amd_reduce_fn)(void *user, double *sums_dev);
int sporco_amd_csc_admm_run(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                            const sporco_amd_admm_ctrl *c, sporco_amd_admm_record *records,
                            int32_t *n_done, double *rho_out, double *u_scale_out,
                            sporco_amd_reduce_fn reduce, void *user);

/* Staged path, for callers that override individual ADMM steps
 * (SURVEY.md section 8(b) "monkey-patch hazard").  Each mirrors one method. */
int sporco_amd_csc_admm_xstep(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]); /* GenericConvBPDN.xstep */
int sporco_amd_csc_admm_relax(sporco_amd_csc_t h, double rlx);   /* ADMMEqual.relax_AX    */
int sporco_amd_csc_admm_ystep(sporco_amd_csc_t h, const sporco_amd_admm_params *p);
int sporco_amd_csc_admm_ustep(sporco_amd_csc_t h, const sporco_amd_admm_params *p);
int sporco_amd_csc_admm_stats(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]);
/* U *= s (the eager form of update_rho's `U /= rsf`, admm.py:573). */
int sporco_amd_csc_scale_u(sporco_amd_csc_t h, double s);

/* irfftn(sum_k Df * rfftn(V)) for V = state `var` (Y or X): reconstruct,
 * sporco/admm/cbpdn.py:373-380.  dst: real (H,W,C,N). */
int sporco_amd_csc_reconstruct(sporco_amd_csc_t h, int var, void *dst);
/* max |conj(Df) * Sf| for the default lambda rule, sporco/admm/cbpdn.py:573-578. */
int sporco_amd_csc_dhs_absmax(sporco_amd_csc_t h, double *out);

/* ---- PGM / FISTA steps (sporco/pgm/pgm.py:779-846, pgm/cbpdn.py:263-372) -- */

#define SPORCO_AMD_PGM_F 0       /* 0.5 * sum |Df.v - Sf|^2 in the unnormalised DFT domain
                                    (obfn_f, pgm/cbpdn.py:358-372)                       */
#define SPORCO_AMD_PGM_DFID 1    /* Parseval-weighted sum |Df.v - Sf|^2 / (H W), not halved
                                    (obfn_dfd, pgm/cbpdn.py:332-345)                     */
#define SPORCO_AMD_PGM_L1 2      /* sum |wl1 * X| of the last prox step (obfn_reg :347-356) */
#define SPORCO_AMD_PGM_HESS 3    /* sum |sum_k Df v|^2 = Re <v, hessian_f(v)> (:302-312)  */

#define SPORCO_AMD_PGM_RSDL 4    /* rfl2norm2(Xf - Yfprv) of pgm_iter (rsdl, pgm/cbpdn.py:314-320) */
#define SPORCO_AMD_PGM_FY 5      /* f at the Yf the step started from (pgm_iter)             */
#define SPORCO_AMD_PGM_LIN 6     /* sum Re(conj(Xf - Yf) grad f(Yf)) of a held pgm_iter
                                    (eval_linear_approx, pgm.py:886-894)                  */
#define SPORCO_AMD_PGM_DXY2 7    /* sum |Xf - Yf|^2, unweighted (backtrack.py:98-100)       */

/* One whole default-option FISTA iteration on device (float32, H and W in {128, 256, 512},
 * even K <= 64 or 72 < K <= 256; since round 6 also H, W in 16 x {10, 12, 14, 15, 18, 20, 21, 24, 25,
 * 27, 28, 30} with even K <= 64 -- SPORCO_AMD_QUERY_FUSED_PGM says which;
 * SPORCO_AMD_EINVAL otherwise -- compose the calls below instead):
 * on_iteration_start (Xfprv = Xf, Yfprv = Yf, by buffer rotation), PGMDFT.xstep
 * (grad_f at Yf, Vf = Yf - grad/L, X = prox_g(irfftn(Vf)), Xf = rfftn(X);
 * sporco/pgm/pgm.py:779-811) and PGMDFT.ystep with the caller's momentum factor
 * Yf = Xf + beta (Xf - Xfprv) (pgm.py:815-831).  out[PGM_F], out[PGM_DFID] (at the
 * new Xf; only with `want_stats`), out[PGM_L1], out[PGM_RSDL], out[PGM_FY].
 * The spectral iterates stay in an internal tile-major layout between such calls
 * and X is rebuilt on demand; any other entry point sees the reference layout. */
typedef struct {
    double L;        /* inverse step size                                   */
    double lmbda;    /* l1 weight (times a scalar L1Weight)                 */
    double beta;     /* momentum factor (t_prev - 1) / t                    */
    uint32_t flags;  /* SPORCO_AMD_FLAG_NONNEG | _NOBNDRY | _DMASK           */
    int32_t dH, dW;  /* filter support, for NOBNDRY                         */
    int32_t want_stats; /* evaluate the objective at the new Xf             */
    int32_t hold;    /* backtracking trial: also out[PGM_LIN], out[PGM_DXY2] (the terms of
                        Q_L, backtrack.py:95-100); the new iterates are kept aside and the
                        call may be repeated with another L from the same Yf until
                        sporco_amd_csc_pgm_commit adopts the last trial.  2: the same for a
                        rule that forms Yf itself (BacktrackRobust, backtrack.py:162-208: Yf
                        is set by sporco_amd_csc_lincomb before the call): no momentum
                        output is written, and after the commit VAR_YF holds the Yf of the
                        PREVIOUS iteration (what PGM.rsdl compares with, pgm.py:835-846,
                        pgm/cbpdn.py:314-320) and VAR_YFPRV the one this trial used        */
} sporco_amd_pgm_params;
int sporco_amd_csc_pgm_iter(sporco_amd_csc_t h, const sporco_amd_pgm_params *p,
                            double out[SPORCO_AMD_OUT_COUNT]);
/* Adopt the iterates of the last held pgm_iter (the buffer rotation of on_iteration_start,
 * pgm.py:835-846); SPORCO_AMD_EINVAL without one. */
int sporco_amd_csc_pgm_commit(sporco_amd_csc_t h);

/* GF = conj(Df) * (sum_k Df*v - Sf) for v = complex state `var` (grad_f,
 * pgm/cbpdn.py:263-279); out[PGM_F], out[PGM_DFID] receive f(v). */
int sporco_amd_csc_pgm_grad(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]);
/* f(v) only (no gradient written): out[PGM_F], out[PGM_DFID], out[PGM_HESS]. */
int sporco_amd_csc_pgm_eval(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]);
/* Proximal step of PGMDFT.xstep (pgm.py:800-803): Vf = Yf - GF/L;
 * X = prox_g(irfftn(Vf)) with prox_l1(., lmbda/L * wl1) + NonNeg/NoBndry
 * (pgm/cbpdn.py:288-300); Xf = rfftn(X); out[PGM_L1] = ||wl1 X||_1. */
int sporco_amd_csc_pgm_prox_step(sporco_amd_csc_t h, double L, double lmbda, uint32_t flags,
                                 int32_t dH, int32_t dW, double out[SPORCO_AMD_OUT_COUNT]);
/* dst = a*va + b*vb + c*vc over complex state arrays (vb, vc may be -1):
 * the momentum step Yf = Xf + beta (Xf - Xfprv) (PGMDFT.ystep, pgm.py:815-831),
 * robust-backtracking and monotone combinations (backtrack.py:181-200).  Between fused
 * iterations (pgm_iter) a combination of iterates / scratch spectra (VAR_T0..T2) into VAR_YF
 * or a scratch spectrum is formed in the internal tile-major layout, like pair_stats of such
 * operands: the iterates stay where the fused kernels want them. */
int sporco_amd_csc_lincomb(sporco_amd_csc_t h, int dst, double a, int va, double b, int vb,
                           double c, int vc);
/* Statistics of d = va - vb (vb may be -1) and g = vg (may be -1):
 *   out[0] = rfl2norm2(d) (half-spectrum Parseval, / (H W): rsdl, pgm/cbpdn.py:314-320)
 *   out[1] = sum Re(conj(d) g)   (eval_linear_approx, pgm.py:886-894)
 *   out[2] = sum |d|^2           out[3] = sum |g|^2 */
int sporco_amd_csc_pair_stats(sporco_amd_csc_t h, int va, int vb, int vg,
                              double out[SPORCO_AMD_OUT_COUNT]);
/* Residual spectra for the step-size policies (sporco/pgm/stepsize.py:67-145), so that they run
 * beside the fused iteration without an X-sized gradient array: the gradient at v is
 * g = conj(Df) e with e = sum_m Df v - Sf (grad_f, pgm/cbpdn.py:263-279), signal sized.
 * pgm_resid stores e of the X-sized spectrum `var` in slot 0..3 (one read pass; shapes the fused
 * kernels serve, single-channel dictionary).  pgm_resid_stats: with d1 = e[a] - e[b],
 * d2 = e[c] - e[d] (b, c, d may be -1; c = -1: d2 = d1) and G = sum_m |Df|^2,
 *   out[0] = sum G |d1|^2        = <g1, g1>                (StepSizePolicyCauchy den, BB num)
 *   out[1] = sum G^2 |d1|^2      = <g1, hessian_f(g1)>     (StepSizePolicyCauchy num, :84-87)
 *   out[2] = sum Re(conj(d2) d1) = <dx, dg> for iterates whose residuals differ by d2 and
 *            gradients by conj(Df) d1                      (StepSizePolicyBB den, :137-141)
 * summed over the rfftn arrays as they are (no Parseval weights), as the reference's np.sum does. */
int sporco_amd_csc_pgm_resid(sporco_amd_csc_t h, int var, int slot);
int sporco_amd_csc_pgm_resid_stats(sporco_amd_csc_t h, int a, int b, int c, int d,
                                   double out[SPORCO_AMD_OUT_COUNT]);
/* cplx_var = rfftn(real_var) / real_var = irfftn(cplx_var) between X-sized state
 * arrays (Xf = rfftn(X), pgm/cbpdn.py:231; Zf = rfftn(Z), pgm/ccmod.py:264-279). */
int sporco_amd_csc_fft_var(sporco_amd_csc_t h, int real_var, int cplx_var);
int sporco_amd_csc_ifft_var(sporco_amd_csc_t h, int cplx_var, int real_var);
/* dst = src (state copy of equal size: on_iteration_start, pgm.py:835-846). */
int sporco_amd_csc_copy(sporco_amd_csc_t h, int dst_var, int src_var);

/* ---- dictionary update (pgm.ccmod.ConvCnstrMOD, sporco/pgm/ccmod.py:139-404) --- */

/* ZF = rfftn(real state `var`): setcoef (pgm/ccmod.py:264-279).  With a handle
 * shared between the X-step and the D-step, var = VAR_Y keeps the coefficient
 * maps on the device (DictLearn.post_xstep, dictlrn/dictlrn.py:379-382).
 * Handles of sporco_amd_csc_create_mc: var = VAR_CX hands over coefficient maps that carry the
 * dictionary's channels, in the layout of that array (H, W, N, Cd, K) -- Cd single-channel
 * updates sharing the penalty / step size (the reference's broadcasting:
 * tests/admm/test_ccmod.py:278-295); VAR_CX is zeroed afterwards.  Served by
 * sporco_amd_csc_ccmod_grad, sporco_amd_csc_masked_grad (dstep) and sporco_amd_csc_cns_iter. */
int sporco_amd_csc_ccmod_setcoef(sporco_amd_csc_t h, int var);
/* DGF = sum_n conj(Zf) (sum_k Zf*v - Sf) for the D-sized complex state `var`
 * (grad_f, pgm/ccmod.py:295-309: inner over axisM then over axisK, channels of
 * a single-channel dictionary folded into the image axis); out[PGM_F],
 * out[PGM_DFID], out[PGM_HESS] as for pgm_grad. */
int sporco_amd_csc_ccmod_grad(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]);
/* Same sums without writing the gradient (obfn_f / obfn_dfd, ccmod.py:340-372). */
int sporco_amd_csc_ccmod_eval(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]);
/* DVF = DYF - DGF/L; DX = Pcn(irfftn(DVF)); DXF = rfftn(DX)  (PGMDFT.xstep with
 * prox_g = Pcn, pgm/ccmod.py:320-323; cnvrep.Pcn, sporco/cnvrep.py:868-913:
 * crop to (dH,dW), zero-pad, optional zero-mean, unit l2 norm per filter). */
int sporco_amd_csc_ccmod_prox_step(sporco_amd_csc_t h, double L, int32_t dH, int32_t dW,
                                   int32_t zero_mean);
/* out[0] = ||Pcn(DX) - DX||_2  (obfn_cns, pgm/ccmod.py:350-355). */
int sporco_amd_csc_ccmod_cnstr(sporco_amd_csc_t h, int32_t dH, int32_t dW, int32_t zero_mean,
                               double out[SPORCO_AMD_OUT_COUNT]);
/* dst(dH,dW,K) = DX[:dH,:dW,:]  (getdict(crop=True), pgm/ccmod.py:283-291). */
int sporco_amd_csc_ccmod_getdict(sporco_amd_csc_t h, int32_t dH, int32_t dW, void *dst);
/* Df = DXF and its Sherman-Morrison denominators, device to device: the
 * X-step's setdict(dstep.getdict()) of DictLearn.post_dstep without PCIe
 * (dictlrn/dictlrn.py:386-389, admm/cbpdn.py:242-256). */
int sporco_amd_csc_setdict_from_dstep(sporco_amd_csc_t h, int32_t dH, int32_t dW);
/* out[0] = sum |v| over the real state `var` (RegL1 of DictLearn.evaluate,
 * dictlrn/cbpdndl.py:519). */
/* ---- masked data fidelity (pgm.cbpdn.ConvBPDNMask, sporco/pgm/cbpdn.py:387-506; ----------
 * pgm.ccmod.ConvCnstrMODMask, sporco/pgm/ccmod.py:408-604) --------------------------------
 * W (cnvrep.mskWshape layout, broadcastable against (H,W,C,N,1)); NULL removes it. */
int sporco_amd_csc_set_data_mask(sporco_amd_csc_t h, const void *w, const int64_t shape[5]);
/* Gradient of (1/2)||W (sum_m d_m * x_m - s)||^2: residual -> irfftn -> W^2 -> rfftn -> adjoint.
 * dstep == 0: `var` is a coefficient spectrum, the gradient goes to SPORCO_AMD_VAR_GF
 * (conj(Df) .), pgm/cbpdn.py:454-477; dstep != 0: `var` is a dictionary spectrum, the gradient
 * goes to SPORCO_AMD_VAR_DGF (sum over images of conj(Zf) .), pgm/ccmod.py:552-575.
 * write_grad == 0: evaluation only -- out[SPORCO_AMD_PGM_DFID] = sum (W R)^2 (spatial domain,
 * twice the data fidelity term) and out[SPORCO_AMD_PGM_F] = (1/2) sum |rfftn(W R)|^2 over the
 * half spectrum (the value backtracking compares, :493-506).  With write_grad != 0 only
 * out[SPORCO_AMD_PGM_DFID] is filled.
 * write_grad == 2 (dstep only in practice): as 1 with the residual weighted by W once instead of
 * W^2 -- the gradient of OnlineConvBPDNMaskDictLearn.dstep (onlinecdl.py:574-590). */
int sporco_amd_csc_masked_grad(sporco_amd_csc_t h, int var, int32_t dstep, int32_t write_grad,
                               double out[SPORCO_AMD_OUT_COUNT]);

/* ---- ADMM consensus dictionary update (admm.ccmod.ConvCnstrMOD_Consensus, -----------
 * sporco/admm/ccmod.py:605-908 on admm.ADMMConsensus, sporco/admm/admm.py:1441-1707) ----
 * One dictionary copy X_n and dual U_n per image (VAR_CX, VAR_CU), consensus variable
 * Y = VAR_DX (its spectrum VAR_DXF is kept current, so sporco_amd_csc_ccmod_getdict and
 * sporco_amd_csc_setdict_from_dstep serve this D-step too).  Coefficient maps come from
 * sporco_amd_csc_ccmod_setcoef.  Handles of sporco_amd_csc_create_mc (a dictionary with Cd > 1
 * channels, ccmod.py:696-698): VAR_CX / VAR_CU are (H, W, N, Cd, K) -- one (Cd, K) block per
 * image --, VAR_DX is (H, W, Cd, K); both the plain and the mask-decoupled update serve them. */
/* Y = Y0 (real (H,W,K), zero-padded filters) or 0; U_n = Y0 / rho or 0 (uinit, ccmod.py:734-742). */
int sporco_amd_csc_cns_init(sporco_amd_csc_t h, const void *Y0, double rho);
typedef struct {
    double rho;      /* penalty parameter                                               */
    double rlx;      /* RelaxParam                                                      */
    double u_scale;  /* pending U /= rsf, as in sporco_amd_admm_params                  */
    uint32_t flags;  /* SPORCO_AMD_FLAG_RESID | SPORCO_AMD_FLAG_OBJ                     */
    int32_t dH, dW;  /* filter support of the constraint set                            */
    int32_t zero_mean;
    int32_t mask_dcpl;  /* 1: ConvCnstrMODMaskDcpl_Consensus (sporco/admm/ccmodmd.py:766-1083), the
                           consensus update with the masked data fidelity split off into a
                           signal-sized block (Y1, U1 = VAR_DMY0, VAR_DMU0; mask by
                           sporco_amd_csc_set_data_mask; state by sporco_amd_csc_cns_md_init).
                           out then holds: R2 = sum_n |X_n - Y|^2, AX2 = sum_n |X_n|^2, Y2 = |Y|^2,
                           U2 = sum_n |U_n|^2, S2 = Parseval sum of |rfftn(U_n) + conj(Zf_n)
                           rfftn(U1_n)|^2 / (H W) (the dual residual's A^T u, :993-996), and for
                           the signal-sized block XRRS_D2 = |AX1nr - Y1 - S|^2, XRRS_AX2 =
                           |AX1nr|^2, XRRS_B2 = |Y1|^2, CGIT = |U1|^2; DFID = |W irfftn(sum_m Zf Yf
                           - Sf)|^2 (:961-970), CNSTR as for the unmasked update.              */
    int32_t phase;      /* image shards over ranks (one process per GPU): 0 = the whole iteration;
                           1 = up to the local mean over this rank's images of alpha X_n +
                           (1 - alpha) Y + U_n, left in the buffer of sporco_amd_csc_cns_mean_ptr;
                           2 = the rest (constraint projection, dual update, sums), after the
                           caller has made that buffer the mean over ALL images -- the consensus
                           average is the one array all-reduce of this update (SURVEY.md 8(e);
                           reference: admm.py:1585-1591).  The X-sized sums of phase 2 are
                           rank-local and are added over the ranks by the caller; S2, Y2 and
                           CNSTR of the unmasked update are dictionary sized (identical on every
                           rank). */
} sporco_amd_cns_params;
/* The dictionary-sized real buffer (H, W, K) that holds the consensus mean between the two
 * phases, and its element count. */
int sporco_amd_csc_cns_mean_ptr(sporco_amd_csc_t h, void **ptr_dev, int64_t *count);
/* State of the masked consensus update: the real signal S (H,W,C,N) kept on the device, Y1 = U1 = 0
 * (ccmodmd.py:869-871).  Call after sporco_amd_csc_cns_init. */
int sporco_amd_csc_cns_md_init(sporco_amd_csc_t h, const void *S);
/* One iteration: xstep per image by Sherman-Morrison (ccmod.py:766-778), relax_AX
 * (admm.py:1608-1616), ystep Y = Pcn(mean_n(AX_n + U_n)) (admm.py:1585-1591, ccmod.py:832-835),
 * ustep.  out: R2 = sum_n |X_n - Y|^2, S2 = |Y - Yprev|^2, AX2 = sum_n |X_n|^2, Y2 = |Y|^2,
 * U2 = sum_n |U_n|^2 (the host applies the sqrt(Nb) and rho factors of admm.py:1673-1707),
 * DFID = Parseval sum of |sum_m Zf Yf - Sf|^2 / (H W) and CNSTR, both evaluated at Y
 * (fEvalX False, gEvalY True: the class defaults, ccmod.py:853-894). */
int sporco_amd_csc_cns_iter(sporco_amd_csc_t h, const sporco_amd_cns_params *p,
                            double out[SPORCO_AMD_OUT_COUNT]);

/* ---- ADMM with mask decoupling: sporco.admm.cbpdn.ConvBPDNMaskDcpl (cbpdn.py:2066-2283) on
 * ConvTwoBlockCnstrnt (:1401-1826) / ADMMTwoBlockCnstrnt (admm.py:989-1437) ------------------
 * Constraint [D; I] x - [y0; y1] = [s; 0].  Block 1 (coefficient sized) lives in the ADMM state
 * VAR_Y / VAR_U, block 0 (signal sized) in VAR_MY0 / VAR_MU0; X / Xf as for ConvBPDN.  The mask W
 * is the data mask of sporco_amd_csc_set_weights(which = 3); single-channel dictionaries. */
/* S: the real signal (H,W,C,N) (the handle otherwise keeps only its spectrum); zeroes Y, U. */
int sporco_amd_csc_mdcpl_init(sporco_amd_csc_t h, const void *S);
/* One iteration.  params: rho, lmbda, rlx, u_scale, flags (NONNEG | NOBNDRY | OBJ | XRRS |
 * GEVAL_Y = AuxVarObj), dH, dW.  xstep (:1610-1643, rho-free), relax_AX (:1664-1677), ystep
 * (:2236-2247, :1647-1660), ustep.  out, block 1 sums in the ADMM slots and block 0 sums beside
 * them (the host adds the pairs): R2 / L21 = |AXnr - y - c|^2, AX2 / RGR = |AXnr|^2, Y2 / CNSTR
 * = |y|^2, U2 / CGIT = |u|^2; S2 = |A^T u|^2 = |irfftn(conj(Df) rfftn(u0)) + u1|^2 (:1814-1818);
 * DFID = |W g0|^2 (twice the data fidelity, :2262-2268), L1 = |wl1 g1|_1; XRRS sums. */
int sporco_amd_csc_mdcpl_iter(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]);

/* ---- online dictionary learning (sporco.dictlrn.onlinecdl.OnlineConvBPDNDictLearn.dstep,
 * onlinecdl.py:310-333) -------------------------------------------------------------------
 * After sporco_amd_csc_ccmod_setcoef and sporco_amd_csc_ccmod_grad(h, SPORCO_AMD_VAR_DF) (the
 * gradient at the current dictionary, left in VAR_DGF): G = irfftn(Df - eta * gradient),
 * D = Pcn(G) -> VAR_DX (and its spectrum VAR_DXF; read with sporco_amd_csc_ccmod_getdict).
 * out[SPORCO_AMD_OUT_CNSTR] = sum (Pcn(G) - G)^2, the square of the reference's Cnstr (:398). */
int sporco_amd_csc_ccmod_sgd_step(sporco_amd_csc_t h, double eta, int32_t dH, int32_t dW,
                                  int32_t zero_mean, double out[SPORCO_AMD_OUT_COUNT]);

/* ---- ADMM dictionary update with one dictionary copy ---------------------------------------
 * sporco.admm.ccmod.ConvCnstrMOD_IterSM / ConvCnstrMOD_CG (ccmod.py:433-601) on ConvCnstrMODBase
 * (:103-429) and ADMMEqual (admm.py:808-983).  X = VAR_DSX, Y = VAR_DX (spectrum VAR_DXF kept
 * current, as for the consensus update), U = VAR_DSU, all (H,W,K) real; Xf = VAR_DYF persists
 * between calls (the CG warm start, ccmod.py:583,594-597).  Coefficient maps come from
 * sporco_amd_csc_ccmod_setcoef.  Single-channel dictionaries. */
#define SPORCO_AMD_DSTEP_ISM 0   /* X-step by linalg.solvemdbi_ism over the images (<= 8 images
                                    times channels), ccmod.py:496-505                        */
#define SPORCO_AMD_DSTEP_CG 1    /* X-step by linalg.solvemdbi_cg (scipy cg semantics: stop at
                                    ||r|| < tol ||b|| at the top of an iteration), :587-601   */
/* Y = U = Y0 (uinit, ccmod.py:298-307) or 0; Xf = 0. */
int sporco_amd_csc_dstep_init(sporco_amd_csc_t h, const void *Y0);
typedef struct {
    double rho;      /* penalty parameter                                               */
    double rlx;      /* RelaxParam                                                      */
    double u_scale;  /* pending U /= rsf, as in sporco_amd_admm_params                  */
    double cg_tol;   /* CG StopTol                                                      */
    uint32_t flags;  /* SPORCO_AMD_FLAG_OBJ | _XRRS | _FEVAL_Y | _GEVAL_Y               */
    int32_t dH, dW;  /* filter support of the constraint set                            */
    int32_t zero_mean;
    int32_t method;  /* SPORCO_AMD_DSTEP_ISM / SPORCO_AMD_DSTEP_CG                      */
    int32_t cg_maxiter;
    int32_t mask_dcpl; /* != 0: ConvCnstrMODMaskDcpl_IterSM / _CG (sporco/admm/ccmodmd.py:27-762):
                        constraint [Z; I] d - [y0; y1] = [s; 0] with block 0 in VAR_DMY0 / _DMU0,
                        mask = the data mask (set_data_mask), X-step system Z^H Z + I (rho-free),
                        y0 = rho (AX0 + u0 - s) / (W^2 + rho).  out then carries the block-0 sums
                        beside the block-1 ones -- L1 / R2 = |AXnr - y - c|^2, L21 / AX2 = |AXnr|^2,
                        RGR / Y2 = |y|^2, slot 15 / U2 = |u|^2 --, S2 = |A^T u|^2 (:557-561),
                        DFID = |W g0|^2 with g0 = y0 (FLAG_GEVAL_Y) or Z d - s (:508-524).     */
} sporco_amd_dstep_params;
/* One iteration: xstep (b = sum_n conj(Zf_n) Sf_n + rho rfftn(Y - U); Xf = (Z^H Z + rho I)^-1 b),
 * relax_AX (admm.py:877-885), ystep Y = Pcn(AX + U) (ccmod.py:363-368), ustep.  out: R2 = |X - Y|^2,
 * S2 = |Y - Yprev|^2, AX2 = |X|^2, Y2, U2; with FLAG_OBJ: DFID (at Xf, or at rfftn(Y) with
 * FLAG_FEVAL_Y) and CNSTR (at X, or at Y with FLAG_GEVAL_Y), ccmod.py:372-410; with FLAG_XRRS the
 * sums of xstep_check (:343-357); CGIT / CGN for the CG method. */
/* State of the mask-decoupling variant: as sporco_amd_csc_dstep_init plus block 0 zeroed and the
 * real signal S (H,W,C,N) kept on the device (NULL: keep the one a previous
 * sporco_amd_csc_mdcpl_init / dstep_md_init call stored). */
int sporco_amd_csc_dstep_md_init(sporco_amd_csc_t h, const void *Y0, const void *S);
int sporco_amd_csc_dstep_iter(sporco_amd_csc_t h, const sporco_amd_dstep_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]);

int sporco_amd_csc_asum(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]);

/* ---- per-kernel timing (HIP events on the handle's stream) ---------------- */

int sporco_amd_csc_profile(sporco_amd_csc_t h, int enable);
/* Reads and clears accumulated time of timing slot `slot`; returns its name. */
int sporco_amd_csc_profile_read(sporco_amd_csc_t h, int slot, const char **name,
                                double *total_ms, int64_t *launches);
int sporco_amd_profile_slots(void);

/* ---- stateless primitives on host arrays (parity entry points) ------------ */

/* rfftn over axes (0,1) of real (H,W,P) -> complex (H,W/2+1,P); sporco/fft.py:257-286. */
int sporco_amd_rfftn2(int dtype, int32_t H, int32_t W, int64_t P, const void *in, void *out);
/* irfftn with s=(H,W): complex (H,W/2+1,P) -> real (H,W,P); sporco/fft.py:288-314. */
int sporco_amd_irfftn2(int dtype, int32_t H, int32_t W, int64_t P, const void *in, void *out);
/* solvedbi_sm (sporco/linalg.py:232-297) with ah: complex (npix,1,K),
 * b, x: complex (npix,CN,K); solves (rho I + a a^H) x = b along K. */
int sporco_amd_solvedbi_sm(int dtype, int64_t npix, int64_t CN, int32_t K, const void *ah,
                           double rho, const void *b, void *x);
/* inner(x, y, axis=K) (sporco/linalg.py:41-88): x complex (npix,1,K) broadcast
 * over CN, y complex (npix,CN,K) -> out complex (npix,CN). */
int sporco_amd_inner(int dtype, int64_t npix, int64_t CN, int32_t K, const void *x,
                     const void *y, void *out);
/* ---- Device arrays and the pre / post-processing around the solvers -------------------------
 * The reference's example pipelines (examples/scripts/csc/cbpdn_gry.py:45-47, :66-77) highpass
 * the image with signal.tikhonov_filter, sparse-code the highpass part and add the lowpass part
 * back to the reconstruction.  With the entry points below the whole chain stays in HBM: one
 * upload of the image, one download of the result.  Buffers are plain device pointers. */
int sporco_amd_dev_malloc(size_t bytes, void **ptr_dev);
int sporco_amd_dev_free(void *ptr_dev);
int sporco_amd_dev_upload(void *dst_dev, const void *src_host, size_t bytes);
int sporco_amd_dev_download(void *dst_host, const void *src_dev, size_t bytes);
/* out = a x + b y on real device arrays of n elements (y may be NULL). */
int sporco_amd_dev_axpby(int dtype, int64_t n, double a, const void *x, double b, const void *y,
                         void *out);
/* signal.tikhonov_filter (sporco/signal.py:244-301) on a device array s (H, W, P), P = product of
 * the remaining axes: symmetric padding by npd, division by 1 + lmbda sum_i |G_i|^2 in the DFT
 * domain, crop; slp and shp = s - slp are device arrays of the shape of s. */
int sporco_amd_tikhonov_filter_dev(int dtype, int32_t H, int32_t W, int64_t P, const void *s_dev,
                                   double lmbda, int32_t npd, void *slp_dev, void *shp_dev);
/* fft.fftconv (sporco/fft.py:376-417) over axes (0, 1) of real device arrays a (ha, wa, ...) and
 * b (hb, wb, ...) whose remaining (up to three, after merging) axes have the extents da / db --
 * each 1 or the output extent d -- circular over (max(ha,hb), max(wa,wb)), result rolled by
 * -(origin_h, origin_w).  out: (H, W, d0, d1, d2). */
int sporco_amd_fftconv_dev(int dtype, int32_t ha, int32_t wa, const int64_t da[3], const void *a_dev,
                           int32_t hb, int32_t wb, const int64_t db[3], const void *b_dev,
                           int32_t origin_h, int32_t origin_w, void *out_dev);
/* set_signal from a device array (H, W, C, N) (no host copy). */
int sporco_amd_csc_set_signal_dev(sporco_amd_csc_t h, const void *S_dev);
/* reconstruct into a device array (H, W, C, N). */
int sporco_amd_csc_reconstruct_dev(sporco_amd_csc_t h, int var, void *dst_dev);
/* Bytes and calls of host <-> device copies made by this library since the last reset:
 * out = {h2d_bytes, h2d_calls, d2h_bytes, d2h_calls}. */
int sporco_amd_transfer_stats(int64_t out[4], int reset);

/* ---- image shards over the GPUs of a node: RCCL inside the library ---------------------------
 * (SURVEY.md 8(b) `allreduce_scalars`, 8(e); the reference's own multi-process split is
 * sporco/dictlrn/prlcnscdl.py:241,508 -- multiprocessing over shared memory, no counterpart of a
 * communicator.)  One communicator per process (one process per GPU): rank 0 obtains an id,
 * the host program hands its 128 bytes to the other ranks by whatever means it has (MPI,
 * torch.distributed's store, a file), every rank creates its communicator from it.  librccl is
 * opened at run time (dlopen: SPORCO_AMD_RCCL_LIB, then librccl.so.1 / librccl.so): the library
 * has no link-time dependency on it, and SPORCO_AMD_EUNSUPPORTED comes back when it is absent. */
typedef struct sporco_amd_comm *sporco_amd_comm_t;
#define SPORCO_AMD_COMM_ID_BYTES 128
#define SPORCO_AMD_COMM_SUM 0
#define SPORCO_AMD_COMM_MAX 2
int sporco_amd_comm_unique_id(void *id128);                      /* ncclGetUniqueId          */
int sporco_amd_comm_create(const void *id128, int32_t rank, int32_t world, int32_t device,
                           sporco_amd_comm_t *out);              /* ncclCommInitRank         */
int sporco_amd_comm_destroy(sporco_amd_comm_t c);
int sporco_amd_comm_info(sporco_amd_comm_t c, int32_t *rank, int32_t *world);
/* In-place all-reduce of `count` reals (SPORCO_AMD_F32 / _F64) of DEVICE memory, enqueued on
 * `stream` (a hipStream_t; NULL: the null stream) -- no host synchronisation. */
int sporco_amd_comm_allreduce(sporco_amd_comm_t c, void *buf_dev, int64_t count, int dtype, int op,
                              void *stream);
/* The same for a handful of HOST doubles (staged through a device buffer of the communicator;
 * returns when the result is in `vals`). */
int sporco_amd_comm_allreduce_host(sporco_amd_comm_t c, double *vals, int32_t n, int op);
/* Attach a communicator to a solver handle (NULL detaches): sporco_amd_csc_admm_run then sums the
 * 16 per-iteration doubles over the ranks itself, on the handle's stream, between the local sums
 * and the control update -- no callback into the host program per iteration (a `reduce` hook
 * given to admm_run takes precedence).  The communicator must outlive the handle's use of it. */
int sporco_amd_csc_set_comm(sporco_amd_csc_t h, sporco_amd_comm_t c);

/* prox_l1 (sporco/prox/_lp.py:144-183), real v of n elements, scalar alpha. */
int sporco_amd_prox_l1(int dtype, int64_t n, const void *v, double alpha, void *out);
/* The same with an array-valued threshold (sporco/prox/_lp.py:144-183 accepts one): v viewed as
 * a 5-D array of `shape`, alpha of `ashape` with extent 1 or the full extent on each axis. */
int sporco_amd_prox_l1w(int dtype, const int64_t shape[5], const void *v, const int64_t ashape[5],
                        const void *alpha, void *out);
/* prox_sl1l2 (sporco/prox/_l21.py:51-88) with the l2 norm over the middle axis
 * of real v shaped (outer, C, inner). */
int sporco_amd_prox_sl1l2(int dtype, int64_t outer, int32_t C, int64_t inner, const void *v,
                          double alpha, double beta, void *out);
/* rfl2norm2 (sporco/fft.py:449-484) of complex xf (H,W/2+1,P) for spatial (H,W). */
int sporco_amd_rfl2norm2(int dtype, int32_t H, int32_t W, int64_t P, const void *xf,
                         double *out);

#ifdef __cplusplus
}
#endif
#endif /* SPORCO_AMD_H */
