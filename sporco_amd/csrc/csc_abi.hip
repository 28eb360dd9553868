// csc_abi.hip -- the extern "C" entry points of the solver handle (include/sporco_amd.h): argument
// checks, device selection, exceptions -> return codes.  All work happens behind CscBase
// (csc_impl.h; implemented by template Csc<T> in csc_api.hip).
#include "csc_impl.h"

using namespace sporco_amd;

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
    h->impl.reset(make_csc(*dims, dict_channels, device, stream));
    *out = h.release();
    SA_API_END
}

int sporco_amd_csc_create_volume(const sporco_amd_dims *dims, int32_t depth, int device, void *stream,
                                 sporco_amd_csc_t *out) {
    SA_API_BEGIN
    SA_REQUIRE(dims && out, "null argument");
    SA_REQUIRE(depth >= 1 && dims->H % depth == 0, "depth must divide the folded first axis dims->H");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        throw Error(SPORCO_AMD_EHIP, "no HIP device visible: libsporco_amd needs an AMD GPU");
    SA_REQUIRE(device >= 0 && device < n, "device index out of range");
    std::unique_ptr<sporco_amd_csc> h(new sporco_amd_csc);
    h->device = device;
    h->depth = depth;
    h->impl.reset(make_csc(*dims, 1, device, stream, depth));
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
    SA_HANDLE_ANY(h);
    h->impl->sync();
    SA_API_END
}

int sporco_amd_csc_stream(sporco_amd_csc_t h, void **stream) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(stream, "null argument");
    *stream = h->impl->stream_handle();
    SA_API_END
}

int sporco_amd_csc_set_hint(sporco_amd_csc_t h, int what, int value) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    h->impl->set_hint(what, value);
    SA_API_END
}

int sporco_amd_csc_query(sporco_amd_csc_t h, int what, int *out) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(out != nullptr, "null output pointer");
    *out = h->impl->query(what);
    SA_API_END
}

int sporco_amd_csc_placement_report(sporco_amd_csc_t h, char *buf, size_t cap) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(buf != nullptr && cap > 0, "null or empty report buffer");
    const std::string r = h->impl->placement();
    std::snprintf(buf, cap, "%s", r.c_str());
    SA_API_END
}

int sporco_amd_csc_set_signal(sporco_amd_csc_t h, const void *S) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(S != nullptr, "S is null");
    h->impl->set_signal(S);
    SA_API_END
}

int sporco_amd_csc_set_dict(sporco_amd_csc_t h, const void *D, int32_t dH, int32_t dW) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(D != nullptr, "D is null");
    h->impl->set_dict(D, dH, dW);
    SA_API_END
}

int sporco_amd_csc_set_dict_imag(sporco_amd_csc_t h, const void *D_imag, int32_t dH, int32_t dW) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->set_dict_imag(D_imag, dH, dW);
    SA_API_END
}

int sporco_amd_csc_set_l1_weight(sporco_amd_csc_t h, const void *w, const int64_t shape[5]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(w == nullptr || shape != nullptr, "shape is null");
    h->impl->set_weight(0, w, shape);
    SA_API_END
}

int sporco_amd_csc_set_l21_weight(sporco_amd_csc_t h, const void *w, const int64_t shape[5]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
    SA_REQUIRE(src != nullptr, "src is null");
    h->impl->upload(var, src);
    SA_API_END
}

int sporco_amd_csc_download(sporco_amd_csc_t h, int var, void *dst) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(dst != nullptr, "dst is null");
    h->impl->download(var, dst);
    SA_API_END
}

int sporco_amd_csc_device_ptr(sporco_amd_csc_t h, int var, void **ptr_dev) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(ptr_dev != nullptr, "ptr_dev is null");
    *ptr_dev = h->impl->device_ptr(var);
    SA_API_END
}

int sporco_amd_csc_admm_iter(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                             double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
    SA_REQUIRE(p && out_dev, "null argument");
    h->impl->admm_iter(*p, out_dev);
    SA_API_END
}

int sporco_amd_csc_admm_xstep(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(p && out, "null argument");
    h->impl->admm_xstep(*p, h->impl->out_dev_default);
    h->impl->read_out(h->impl->out_dev_default, out);
    SA_API_END
}

int sporco_amd_csc_admm_relax(sporco_amd_csc_t h, double rlx) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    h->impl->admm_relax(rlx);
    SA_API_END
}

int sporco_amd_csc_admm_ystep(sporco_amd_csc_t h, const sporco_amd_admm_params *p) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(p, "null argument");
    h->impl->admm_ystep(*p);
    SA_API_END
}

int sporco_amd_csc_admm_ustep(sporco_amd_csc_t h, const sporco_amd_admm_params *p) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(p, "null argument");
    h->impl->admm_ustep(*p);
    SA_API_END
}

int sporco_amd_csc_admm_stats(sporco_amd_csc_t h, const sporco_amd_admm_params *p,
                              double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(p && out, "null argument");
    h->impl->admm_stats(*p, h->impl->out_dev_default);
    h->impl->read_out(h->impl->out_dev_default, out);
    SA_API_END
}

int sporco_amd_csc_scale_u(sporco_amd_csc_t h, double s) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    h->impl->scale_u(s);
    SA_API_END
}

int sporco_amd_csc_reconstruct(sporco_amd_csc_t h, int var, void *dst) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(dst != nullptr, "dst is null");
    h->impl->reconstruct(var, dst);
    SA_API_END
}

int sporco_amd_csc_dhs_absmax(sporco_amd_csc_t h, double *out) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->pgm_grad(var, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_pgm_commit(sporco_amd_csc_t h) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->pgm_eval(var, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_pgm_prox_step(sporco_amd_csc_t h, double L, double lmbda, uint32_t flags,
                                 int32_t dH, int32_t dW, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
    h->impl->lincomb(dst, a, va, b, vb, c, vc);
    SA_API_END
}

int sporco_amd_csc_pair_stats(sporco_amd_csc_t h, int va, int vb, int vg,
                              double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->pair_stats(va, vb, vg, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_pgm_resid(sporco_amd_csc_t h, int var, int slot) {
    SA_API_BEGIN
    SA_HANDLE(h);
    h->impl->pgm_resid(var, slot);
    SA_API_END
}

int sporco_amd_csc_pgm_resid_stats(sporco_amd_csc_t h, int a, int b, int c, int d,
                                   double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->pgm_resid_stats(a, b, c, d, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_fft_var(sporco_amd_csc_t h, int real_var, int cplx_var) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    h->impl->fft_var(real_var, cplx_var, false);
    SA_API_END
}

int sporco_amd_csc_ifft_var(sporco_amd_csc_t h, int cplx_var, int real_var) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    h->impl->fft_var(real_var, cplx_var, true);
    SA_API_END
}

int sporco_amd_csc_copy(sporco_amd_csc_t h, int dst_var, int src_var) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    h->impl->copy(dst_var, src_var);
    SA_API_END
}

int sporco_amd_csc_ccmod_setcoef(sporco_amd_csc_t h, int var) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    h->impl->ccmod_setcoef(var);
    SA_API_END
}

int sporco_amd_csc_ccmod_grad(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->ccmod_grad(var, true, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_ccmod_eval(sporco_amd_csc_t h, int var, double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->ccmod_grad(var, false, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_ccmod_prox_step(sporco_amd_csc_t h, double L, int32_t dH, int32_t dW,
                                   int32_t zero_mean) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
    SA_REQUIRE(L > 0.0, "L must be positive");
    h->impl->ccmod_prox_step(L, dH, dW, zero_mean != 0);
    SA_API_END
}

int sporco_amd_csc_ccmod_cnstr(sporco_amd_csc_t h, int32_t dH, int32_t dW, int32_t zero_mean,
                               double out[SPORCO_AMD_OUT_COUNT]) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
    SA_REQUIRE(out != nullptr, "out is null");
    double *sb = stats_buf(h);
    h->impl->asum(var, sb);
    h->impl->read_out(sb, out);
    SA_API_END
}

int sporco_amd_csc_profile(sporco_amd_csc_t h, int enable) {
    SA_API_BEGIN
    SA_HANDLE_ANY(h);
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
    SA_HANDLE_ANY(h);
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
