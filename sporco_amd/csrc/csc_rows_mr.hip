// csc_rows_mr.hip -- the register-resident row passes (csc_rows.h) at the mixed-radix widths
// W = 16 N1 = 160, 192, 224, 240, 288, 320, 336, 384, 400, 432, 448, 480: 16 waves x N1 = 10 ... 30
// points per thread, the in-register transform of N1 points from regfft.h (radices 7, 5, 3 then 2).
// The kernels are the templates of csc_rows.hip instantiated with that N1 (translation units of
// their own -- this file for N1 <= 20, csc_rows_mr2.hip for the rest -- so that they compile side by
// side with the power-of-two ones).  Built: the variants of admm.cbpdn.ConvBPDN -- scalar or array
// L1Weight, NonNegCoef, NoBndryCross (sporco/admm/cbpdn.py:267-311, 614-630) -- and of ConvBPDNJoint
// (:785-807; scalar weights), and the proximal row pass of FISTA, at image sizes such as the
// reference's own odd-sized tests (tests/admm/test_cbpdn.py:204-225).
#ifndef SA_MR_PART
#define SA_MR_PART 0
#endif
#define SA_ROWS_MR_TU
#include "csc_rows.hip"

// the lengths this translation unit instantiates
#if SA_MR_PART == 0
#define SA_MR_PART_LENGTHS(X) X(10) X(12) X(14) X(15) X(18) X(20)
#else
#define SA_MR_PART_LENGTHS(X) X(21) X(24) X(25) X(27) X(28) X(30)
#endif

namespace sporco_amd {

namespace {

// MODE 1 (csc_rows.hip): weight array, NoBndryCross and / or the AddMaskSim mask; without an array the kernel reads a
// device-resident 1.0 through zero strides
template <typename A> bool general_options(A &a) {
    const bool g = a.wl1.ptr != nullptr || (a.flags & F_NOBNDRY) || a.ams_bits != nullptr;
    if (g && !a.wl1.ptr) {
        a.wl1 = Weight<float>();
        a.wl1.ptr = device_one();
    }
    return g;
}

template <int N1> void fwd_mr(hipStream_t st, RowsFwdArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<16>(&rows_fwd_kernel<16, false, false, false, 0, N1>);
        set_lds_attr<16>(&rows_fwd_kernel<16, false, true, false, 0, N1>);
        set_lds_attr<16>(&rows_fwd_kernel<16, false, true, false, 1, N1>);
    }
    const dim3 block(16 * 64);
    const size_t lds = rows_lds_bytes(16);
    if (a.v && (a.flags & F_JOINT)) {
        // the V form of ConvBPDNJoint: tiles as the joint epilogue (one image, 32 filters, all channels)
        static PerDeviceOnce jattr;
        if (jattr.first()) set_lds_attr<16>(&rows_fwd_kernel<16, false, true, true, 0, N1>);
        SA_REQUIRE(rows_joint_supported<float>(a.W, a.C, a.K) && a.C * a.N == a.CN && !a.wl1.ptr &&
                       !(a.flags & F_NOBNDRY),
                   "configuration not handled by the joint row pass");
        const dim3 jgrid = rows_grid(a, 16, (int64_t)a.N * (a.K / 32), a.H, 0);
        hipLaunchKernelGGL((rows_fwd_kernel<16, false, true, true, 0, N1>), jgrid, block, lds, st, a);
        return;
    }
    const dim3 grid = rows_grid(a, 16, ceil_div(a.P, 128), a.H, 0);
    if (!a.v) {
        hipLaunchKernelGGL((rows_fwd_kernel<16, false, false, false, 0, N1>), grid, block, lds, st, a);
    } else if (general_options(a)) {
        SA_REQUIRE(a.C * a.N == a.CN, "the derivation needs the channel / image split");
        hipLaunchKernelGGL((rows_fwd_kernel<16, false, true, false, 1, N1>), grid, block, lds, st, a);
    } else {
        hipLaunchKernelGGL((rows_fwd_kernel<16, false, true, false, 0, N1>), grid, block, lds, st, a);
    }
}

template <int N1, bool EMIT, int MODE> void post_mr_mode(hipStream_t st, const RowsPostArgs<float> &a, dim3 grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, MODE, EMIT, false, 0, N1>);
        set_lds_attr<16>(&rows_inv_post_kernel<16, true, MODE, EMIT, false, 0, N1>);
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, MODE, EMIT, false, 1, N1>);
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, MODE, EMIT, false, 2, N1>);
    }
    const dim3 block(16 * 64);
    const size_t lds = rows_lds_bytes(16);
    if (a.v_out) {
        SA_REQUIRE(!a.x, "the V form has no X output");
        if (a.v_in) hipLaunchKernelGGL((rows_inv_post_kernel<16, false, MODE, EMIT, false, 2, N1>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((rows_inv_post_kernel<16, false, MODE, EMIT, false, 1, N1>), grid, block, lds, st, a);
        return;
    }
    SA_REQUIRE(!a.v_in, "a V-form input needs a V-form output");
    if (a.x) hipLaunchKernelGGL((rows_inv_post_kernel<16, true, MODE, EMIT, false, 0, N1>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((rows_inv_post_kernel<16, false, MODE, EMIT, false, 0, N1>), grid, block, lds, st, a);
}
template <int N1, bool EMIT> void post_mr_joint(hipStream_t st, const RowsPostArgs<float> &a, dim3 grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, 0, EMIT, true, 0, N1>);
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, 0, EMIT, true, 1, N1>);
        set_lds_attr<16>(&rows_inv_post_kernel<16, false, 0, EMIT, true, 2, N1>);
    }
    const dim3 block(16 * 64);
    const size_t lds = rows_lds_bytes(16);
    if (a.v_out && a.v_in) hipLaunchKernelGGL((rows_inv_post_kernel<16, false, 0, EMIT, true, 2, N1>), grid, block, lds, st, a);
    else if (a.v_out) hipLaunchKernelGGL((rows_inv_post_kernel<16, false, 0, EMIT, true, 1, N1>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((rows_inv_post_kernel<16, false, 0, EMIT, true, 0, N1>), grid, block, lds, st, a);
}
template <int N1, bool EMIT> void post_mr(hipStream_t st, RowsPostArgs<float> &a, dim3 grid) {
    if (a.flags & F_JOINT) post_mr_joint<N1, EMIT>(st, a, grid);
    else if (general_options(a)) post_mr_mode<N1, EMIT, 1>(st, a, grid);
    else post_mr_mode<N1, EMIT, 0>(st, a, grid);
}

template <int N1> void prox_mr(hipStream_t st, RowsProxArgs<float> &a, dim3 grid) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) {
        set_lds_attr<16>(&rows_inv_prox_fwd_kernel<16, false, N1>);
        set_lds_attr<16>(&rows_inv_prox_fwd_kernel<16, true, N1>);
    }
    // (a weight array, NoBndryCross, or a negative threshold: the variant that makes no assumption)
    const bool general = a.wl1.ptr != nullptr || (a.flags & F_NOBNDRY) || a.thr < 0.f;
    if (general && !a.wl1.ptr) a.wl1.ptr = device_one();
    if (general)
        hipLaunchKernelGGL((rows_inv_prox_fwd_kernel<16, true, N1>), grid, dim3(16 * 64), rows_lds_bytes(16), st, a);
    else
        hipLaunchKernelGGL((rows_inv_prox_fwd_kernel<16, false, N1>), grid, dim3(16 * 64), rows_lds_bytes(16), st, a);
}

}  // namespace

#if SA_MR_PART == 0
#define SA_MR_FN(name) name
void launch_rows_fwd_mr2(hipStream_t st, RowsFwdArgs<float> &a);
void launch_rows_inv_post_mr2(hipStream_t st, RowsPostArgs<float> &a, dim3 grid, bool emit);
void launch_rows_inv_prox_fwd_mr2(hipStream_t st, RowsProxArgs<float> &a, dim3 grid);
#else
#define SA_MR_FN(name) name##2
#endif

#if SA_MR_PART == 0
void launch_rows_fwd_mr(hipStream_t st, const RowsFwdArgs<float> &a_in) {
    RowsFwdArgs<float> a = a_in;
    SA_REQUIRE(rows_mr_width(a.W) && a.K % 2 == 0 && a.H <= 65535, "shape not handled by the mixed-radix row kernels");
    SA_REQUIRE(!a.y_bcast, "mixed-radix widths: no broadcast form");
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: fwd_mr<n>(st, a); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: launch_rows_fwd_mr2(st, a);
    }
    SA_HIP(hipGetLastError());
}

int64_t launch_rows_inv_post_mr(hipStream_t st, const RowsPostArgs<float> &a_in) {
    RowsPostArgs<float> a = a_in;
    SA_REQUIRE(rows_mr_width(a.W) && a.K % 2 == 0 && a.H <= 65535, "shape not handled by the mixed-radix row kernels");
    SA_REQUIRE(!a.t_odd, "mixed-radix widths: no striped spectrum");
    const bool joint = a.flags & F_JOINT;
    if (joint) {
        SA_REQUIRE(!a.v_in || a.v_out, "a V-form input needs a V-form output");
        SA_REQUIRE(rows_joint_supported<float>(a.W, a.C, a.K) && !a.wl1.ptr && !(a.flags & F_NOBNDRY) && !a.x,
                   "configuration not handled by the joint row epilogue");
    }
    // (ConvBPDNJoint tiles by (image, 32 filters): all channels of a pixel in one wave, csc_rows.h)
    const int64_t tx = joint ? (int64_t)a.N * (a.K / 32) : ceil_div(a.P, 128);
    const bool emit = a.t_next != nullptr;
    const dim3 grid = rows_grid(a, 16, tx, a.H, emit);
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: emit ? post_mr<n, true>(st, a, grid) : post_mr<n, false>(st, a, grid); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: launch_rows_inv_post_mr2(st, a, grid, emit);
    }
    SA_HIP(hipGetLastError());
    return tx * a.H;
}

int64_t launch_rows_inv_prox_fwd_mr(hipStream_t st, const RowsProxArgs<float> &a_in) {
    RowsProxArgs<float> a = a_in;
    SA_REQUIRE(rows_mr_width(a.W) && a.K % 2 == 0 && a.H <= 65535, "shape not handled by the mixed-radix row kernels");
    const int64_t tx = ceil_div(a.P, 128);
    const dim3 grid = rows_grid(a, 16, tx, a.H, 0);
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: prox_mr<n>(st, a, grid); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: launch_rows_inv_prox_fwd_mr2(st, a, grid);
    }
    SA_HIP(hipGetLastError());
    return tx * a.H;
}
#else
// the second half of the lengths (csc_rows_mr2.hip): called by the dispatchers above
void launch_rows_fwd_mr2(hipStream_t st, RowsFwdArgs<float> &a) {
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: fwd_mr<n>(st, a); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "width not handled by the mixed-radix row kernels");
    }
}
void launch_rows_inv_post_mr2(hipStream_t st, RowsPostArgs<float> &a, dim3 grid, bool emit) {
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: emit ? post_mr<n, true>(st, a, grid) : post_mr<n, false>(st, a, grid); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "width not handled by the mixed-radix row kernels");
    }
}
void launch_rows_inv_prox_fwd_mr2(hipStream_t st, RowsProxArgs<float> &a, dim3 grid) {
    switch (a.W / 16) {
#define SA_MR_CASE(n) case n: prox_mr<n>(st, a, grid); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "width not handled by the mixed-radix row kernels");
    }
}
#endif

}  // namespace sporco_amd
