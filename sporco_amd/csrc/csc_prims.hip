// csc_prims.hip -- the stateless primitives of the C ABI (include/sporco_amd.h: rfftn2, irfftn2,
// solvedbi_sm, inner, prox_*, rfl2norm2, device arrays, tikhonov_filter / fftconv pipelines) and
// the handle entry points that take device-resident operands.
#include "csc_impl.h"

using namespace sporco_amd;

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
