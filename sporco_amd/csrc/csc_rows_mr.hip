// csc_rows_mr.hip -- the register-resident row passes (csc_rows.h) at the mixed-radix widths
// W = 16 N1 = 160, 192, 224, 240, 288, 320, 336, 384, 400, 432, 448, 480: 16 waves x N1 = 10 ... 30
// points per thread, the in-register transform of N1 points from regfft.h (radices 7, 5, 3 then 2).  The kernels are the
// templates of csc_rows.hip instantiated with that N1 (a translation unit of its own: they compile
// side by side with the power-of-two ones); only the variants of plain ConvBPDN are built -- scalar
// weights, no NoBndryCross / AddMaskSim / Joint -- which is what sporco/admm/cbpdn.py:267-311, 614-630
// runs with default options at image sizes such as the reference's own odd-sized tests
// (tests/admm/test_cbpdn.py:204-225).
#define SA_ROWS_MR_TU
#include "csc_rows.hip"

namespace sporco_amd {

namespace {

template <int N1> void fwd_mr(hipStream_t st, RowsFwdArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<16>(&rows_fwd_kernel<16, false, false, false, 0, N1>);
        set_lds_attr<16>(&rows_fwd_kernel<16, false, true, false, 0, N1>);
    }
    const dim3 grid = rows_grid(a, 16, ceil_div(a.P, 128), a.H, 0);
    if (a.v)
        hipLaunchKernelGGL((rows_fwd_kernel<16, false, true, false, 0, N1>), grid, dim3(16 * 64), rows_lds_bytes(16),
                           st, a);
    else
        hipLaunchKernelGGL((rows_fwd_kernel<16, false, false, false, 0, N1>), grid, dim3(16 * 64), rows_lds_bytes(16),
                           st, a);
}

template <int N1, bool EMIT> void post_mr(hipStream_t st, const RowsPostArgs<float> &a, dim3 grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, 0, EMIT, false, 0, N1>);
        set_lds_attr<16>(&rows_inv_post_kernel<16, true, 0, EMIT, false, 0, N1>);
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, 0, EMIT, false, 1, N1>);
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, 0, EMIT, false, 2, N1>);
    }
    const dim3 block(16 * 64);
    const size_t lds = rows_lds_bytes(16);
    if (a.v_out) {
        SA_REQUIRE(!a.x, "the V form has no X output");
        if (a.v_in) hipLaunchKernelGGL((rows_inv_post_kernel<16, false, 0, EMIT, false, 2, N1>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((rows_inv_post_kernel<16, false, 0, EMIT, false, 1, N1>), grid, block, lds, st, a);
        return;
    }
    SA_REQUIRE(!a.v_in, "a V-form input needs a V-form output");
    if (a.x) hipLaunchKernelGGL((rows_inv_post_kernel<16, true, 0, EMIT, false, 0, N1>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((rows_inv_post_kernel<16, false, 0, EMIT, false, 0, N1>), grid, block, lds, st, a);
}

template <int N1> void prox_mr(hipStream_t st, const RowsProxArgs<float> &a, dim3 grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) set_lds_attr<16>(&rows_inv_prox_fwd_kernel<16, false, N1>);
    hipLaunchKernelGGL((rows_inv_prox_fwd_kernel<16, false, N1>), grid, dim3(16 * 64), rows_lds_bytes(16), st, a);
}

}  // namespace

void launch_rows_fwd_mr(hipStream_t st, const RowsFwdArgs<float> &a_in) {
    RowsFwdArgs<float> a = a_in;
    SA_REQUIRE(rows_mr_width(a.W) && a.K % 2 == 0 && a.H <= 65535, "shape not handled by the mixed-radix row kernels");
    SA_REQUIRE(!a.y_bcast && !(a.flags & (F_JOINT | F_NOBNDRY)) && !a.wl1.ptr && !a.ams_bits,
               "mixed-radix widths: plain ConvBPDN options only");
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: fwd_mr<n>(st, a); break;
    SA_MR_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "width not handled by the mixed-radix row kernels");
    }
    SA_HIP(hipGetLastError());
}

int64_t launch_rows_inv_post_mr(hipStream_t st, const RowsPostArgs<float> &a_in) {
    RowsPostArgs<float> a = a_in;
    SA_REQUIRE(rows_mr_width(a.W) && a.K % 2 == 0 && a.H <= 65535, "shape not handled by the mixed-radix row kernels");
    SA_REQUIRE(!(a.flags & (F_JOINT | F_NOBNDRY)) && !a.wl1.ptr && !a.ams_bits && !a.emit_u && !a.t_odd,
               "mixed-radix widths: plain ConvBPDN options only");
    const int64_t tx = ceil_div(a.P, 128);
    const bool emit = a.t_next != nullptr;
    const dim3 grid = rows_grid(a, 16, tx, a.H, emit);
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: emit ? post_mr<n, true>(st, a, grid) : post_mr<n, false>(st, a, grid); break;
    SA_MR_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "width not handled by the mixed-radix row kernels");
    }
    SA_HIP(hipGetLastError());
    return tx * a.H;
}

int64_t launch_rows_inv_prox_fwd_mr(hipStream_t st, const RowsProxArgs<float> &a_in) {
    RowsProxArgs<float> a = a_in;
    SA_REQUIRE(rows_mr_width(a.W) && a.K % 2 == 0 && a.H <= 65535, "shape not handled by the mixed-radix row kernels");
    SA_REQUIRE(!a.wl1.ptr && !(a.flags & F_NOBNDRY) && a.thr >= 0.f, "mixed-radix widths: plain options only");
    const int64_t tx = ceil_div(a.P, 128);
    const dim3 grid = rows_grid(a, 16, tx, a.H, 0);
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: prox_mr<n>(st, a, grid); break;
    SA_MR_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "width not handled by the mixed-radix row kernels");
    }
    SA_HIP(hipGetLastError());
    return tx * a.H;
}

}  // namespace sporco_amd
