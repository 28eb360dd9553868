// csc_pgm_mr.hip -- the column kernels of the fused FISTA iteration and of the tile-major dictionary
// update (csc_pgm.h) at the mixed-radix heights H = 16 N1, N1 = 10 ... 30 (regfft.h SA_MR_LENGTHS): the
// templates of csc_pgm.hip instantiated with that many rows per thread, run-time K <= 64, one
// workgroup per tile.  Two translation units (this file: N1 <= 20; csc_pgm_mr2.hip: the rest).
// Reference: sporco/pgm/cbpdn.py:263-372, sporco/pgm/pgm.py:779-846, sporco/pgm/ccmod.py:295-323.
#ifndef SA_MR_PART
#define SA_MR_PART 0
#endif
#define SA_PGM_MR_TU
#include "csc_pgm.hip"

#if SA_MR_PART == 0
#define SA_MR_PART_LENGTHS(X) X(10) X(12) X(14) X(15) X(18) X(20)
#else
#define SA_MR_PART_LENGTHS(X) X(21) X(24) X(25) X(27) X(28) X(30)
#endif

namespace sporco_amd {

namespace {

template <int N1, bool BT> void grad_mr(hipStream_t st, const PgmColsArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) set_lds(&pgm_grad_ifft_kernel<16, 1, 0, BT, false, false, N1>, pgm_lds_bytes(16, 1));
    hipLaunchKernelGGL((pgm_grad_ifft_kernel<16, 1, 0, BT, false, false, N1>), dim3(pgm_all_tiles(a)), dim3(1024),
                       pgm_lds_bytes(16, 1), st, a);
}
template <int N1, bool STATS, bool PLAIN, bool BT> void mom_mr(hipStream_t st, const PgmColsArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first())
        set_lds(&pgm_fft_momentum_kernel<16, 1, 0, STATS, PLAIN, BT, false, N1>, pgm_lds_bytes(16, 1));
    hipLaunchKernelGGL((pgm_fft_momentum_kernel<16, 1, 0, STATS, PLAIN, BT, false, N1>), dim3(pgm_all_tiles(a), 1u),
                       dim3(1024), pgm_lds_bytes(16, 1), st, a);
}
template <int N1> void grad_mr_eyin(hipStream_t st, const PgmColsArgs<float> &a) {
    static PerDeviceOnce attr_set;
    if (attr_set.first()) set_lds(&pgm_grad_ifft_kernel<16, 1, 0, false, false, true, N1>, pgm_lds_bytes(16, 1));
    hipLaunchKernelGGL((pgm_grad_ifft_kernel<16, 1, 0, false, false, true, N1>), dim3(pgm_all_tiles(a)), dim3(1024),
                       pgm_lds_bytes(16, 1), st, a);
}
template <int N1> void grad_mr_any(hipStream_t st, const PgmColsArgs<float> &a) {
    if (a.ey_in) grad_mr_eyin<N1>(st, a);      // (the masked classes: the residual comes from memory)
    else if (a.ey) grad_mr<N1, true>(st, a);
    else grad_mr<N1, false>(st, a);
}
template <int N1> void mom_mr_any(hipStream_t st, const PgmColsArgs<float> &a, bool plain) {
    if (plain) mom_mr<N1, false, true, false>(st, a);
    else if (a.ey) mom_mr<N1, true, false, true>(st, a);
    else if (a.want_stats) mom_mr<N1, true, false, false>(st, a);
    else mom_mr<N1, false, false, false>(st, a);
}
template <int N1> void ccmod_mr(hipStream_t st, const CcmodTiledArgs<float> &a, unsigned grid) {
    hipLaunchKernelGGL((ccmod_grad_tiled_kernel<16, 0, 0, N1>), dim3(grid), dim3(1024), sizeof(double) * 4 * 16, st, a);
}

}  // namespace

#if SA_MR_PART == 0
void launch_pgm_grad_ifft_mr2(hipStream_t st, const PgmColsArgs<float> &a);
void launch_pgm_fft_momentum_mr2(hipStream_t st, const PgmColsArgs<float> &a, bool plain);
void launch_ccmod_grad_tiled_mr2(hipStream_t st, const CcmodTiledArgs<float> &a, unsigned grid);

int64_t launch_pgm_grad_ifft_mr(hipStream_t st, const PgmColsArgs<float> &a) {
    SA_REQUIRE(fused_mr_height(a.H) && a.K >= 1 && a.K <= 64 && !(a.ey_in && a.ey),
               "shape / mode not handled by the mixed-radix FISTA kernels");
    switch (a.H / 16) {
#define SA_MR_CASE(n) case n: grad_mr_any<n>(st, a); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: launch_pgm_grad_ifft_mr2(st, a);
    }
    SA_HIP(hipGetLastError());
    return (int64_t)(a.W / 2 + 1) * a.CN;
}
int64_t launch_pgm_fft_momentum_mr(hipStream_t st, const PgmColsArgs<float> &a, bool plain) {
    SA_REQUIRE(fused_mr_height(a.H) && a.K >= 1 && a.K <= 64, "shape not handled by the mixed-radix FISTA kernels");
    SA_REQUIRE(plain || !a.ey || a.want_stats, "the backtracking sums need want_stats");
    switch (a.H / 16) {
#define SA_MR_CASE(n) case n: mom_mr_any<n>(st, a, plain); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: launch_pgm_fft_momentum_mr2(st, a, plain);
    }
    SA_HIP(hipGetLastError());
    return (int64_t)(a.W / 2 + 1) * a.CN;
}
int64_t launch_ccmod_grad_tiled_mr(hipStream_t st, const CcmodTiledArgs<float> &a) {
    SA_REQUIRE(fused_mr_height(a.H) && a.K >= 1 && a.K <= 64, "shape not handled by the mixed-radix D-step kernel");
    const unsigned grid = (unsigned)((a.W / 2 + 1) * a.G);
    switch (a.H / 16) {
#define SA_MR_CASE(n) case n: ccmod_mr<n>(st, a, grid); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: launch_ccmod_grad_tiled_mr2(st, a, grid);
    }
    SA_HIP(hipGetLastError());
    return grid;
}
#else
void launch_pgm_grad_ifft_mr2(hipStream_t st, const PgmColsArgs<float> &a) {
    switch (a.H / 16) {
#define SA_MR_CASE(n) case n: grad_mr_any<n>(st, a); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "height not handled by the mixed-radix FISTA kernels");
    }
}
void launch_pgm_fft_momentum_mr2(hipStream_t st, const PgmColsArgs<float> &a, bool plain) {
    switch (a.H / 16) {
#define SA_MR_CASE(n) case n: mom_mr_any<n>(st, a, plain); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "height not handled by the mixed-radix FISTA kernels");
    }
}
void launch_ccmod_grad_tiled_mr2(hipStream_t st, const CcmodTiledArgs<float> &a, unsigned grid) {
    switch (a.H / 16) {
#define SA_MR_CASE(n) case n: ccmod_mr<n>(st, a, grid); break;
    SA_MR_PART_LENGTHS(SA_MR_CASE)
#undef SA_MR_CASE
    default: SA_REQUIRE(false, "height not handled by the mixed-radix D-step kernel");
    }
}
#endif

}  // namespace sporco_amd
