// csc_ctl_dev.h -- device-side pieces of the device-driven ADMM solve that more than one kernel
// uses: the control-block update (csc_kernels.hip: admm_ctl_update_kernel) and the fixed-order
// final reduction of block partials (finalize_kernel), also evaluated inside the one-launch
// solve of small problems (csc_rows.hip: admm_persist_kernel) -- same operations, same order,
// same bits.
#pragma once

#include "csc_kernels.h"
#include <gfx950_intrin.h>

#include "../../include/sporco_amd.h"

namespace sporco_amd {

__device__ __forceinline__ void admm_ctl_derive(AdmmCtl *c) {
#pragma clang fp contract(off)
    // what the iteration kernels read, from (rho, u_scale): the casts of
    // csc_api.hip admm_iter_fused ((T)p.rho, (T)(p.lmbda / p.rho), (T)p.u_scale)
    c->rho_f = (float)c->rho;
    c->thr_f = (float)(c->lmbda / c->rho);
    c->thr21_f = (float)(c->mu21 / c->rho);
    c->u_scale_f = (float)c->u_scale;
    c->stable_run = c->u_scale == 1.0 ? c->stable_run + 1 : 0;
    // (no_speculation: 0 = emit once rho has been stable for two iterations, 1 = never,
    // 2 = always -- small problems, where a wasted emit costs less than a launch that returns)
    c->emit = c->no_speculation == 2 ? 1 : ((c->stable_run >= 2 && !c->no_speculation) ? 1 : 0);
    c->skip_fwd = (c->emitted && c->u_scale == 1.0) ? 1 : 0;
}

// The arithmetic below restates, operation by operation, sporco_amd/admm/cbpdn.py
// residual_norms and sporco_amd/admm/admm.py compute_residuals / rho_scale_factor /
// update_rho (themselves sporco/admm/admm.py:462-486, 549-575) for a solver whose real
// type is T: sums, norms, residuals and tolerances are float64; rho, tau, mu, xi are T
// scalars, products of two of them are formed in T, and a multiplier that was clipped to
// tau is a T value (so 1 / tau is a T division) -- NumPy's scalar promotion rules.
// (one thread; rec == nullptr: the control block alone is advanced -- the other workgroups of
// the one-launch solve, which keep a copy each and advance it identically)
template <typename T>
__device__ __forceinline__ void admm_ctl_update_dev(AdmmCtl *c, const double *sums, AdmmRecord *rec,
                                                    int index) {
    // (no fused multiply-adds: the host code this mirrors rounds every product)
#pragma clang fp contract(off)
    if (c->stop) return;
    const double rho = c->rho;
    if (rec) {
        for (int i = 0; i < 16; ++i) rec->sums[i] = sums[i];
        rec->rho = rho;
        rec->u_scale = c->u_scale;
        rec->k = c->k;
        rec->emit = c->emit;
        rec->skip_fwd = c->skip_fwd;
    }
    double r = 0.0, s = 0.0, epri = 0.0, edua = 0.0;
    double rho_new = rho, u_scale_new = 1.0;
    int stop = 0;
    if (c->need_resid) {
        const double nr = sqrt(sums[SPORCO_AMD_OUT_R2]);
        const double ns = rho * sqrt(sums[SPORCO_AMD_OUT_S2]);
        const double nax = sqrt(sums[SPORCO_AMD_OUT_AX2]), ny = sqrt(sums[SPORCO_AMD_OUT_Y2]);
        double rn = nax >= ny ? nax : ny;
        double sn = rho * sqrt(sums[SPORCO_AMD_OUT_U2]);
        if (c->stdres) {
            r = nr;
            s = ns;
            epri = c->sqrt_nc * c->abstol + rn * c->reltol;
            edua = c->sqrt_nx * c->abstol + sn * c->reltol;
        } else {
            if (rn == 0.0) rn = 1.0;
            if (sn == 0.0) sn = 1.0;
            r = nr / rn;
            s = ns / sn;
            epri = c->sqrt_nc * c->abstol / rn + c->reltol;
            edua = c->sqrt_nx * c->abstol / sn + c->reltol;
        }
        const int k = c->k;
        if (c->autorho && k != 0 && ((k + 1) % c->period) == 0) {
            const T tau = (T)c->tau, mu = (T)c->mu, xi = (T)c->xi;
            double mlt_d = 0.0;     // the multiplier when it is a float64 value ...
            bool mlt_is_t = true;   // ... or tau itself (a T value)
            if (c->autoscaling && !(s == 0.0 || r == 0.0)) {
                const double sx = s * (double)xi;
                mlt_d = sqrt(r > sx ? r / sx : sx / r);
                mlt_is_t = mlt_d > (double)tau;
            }
            double rsf = 1.0;       // float(rsf) of the host code
            if (r > (double)(T)(xi * mu) * s) {
                rsf = mlt_is_t ? (double)tau : mlt_d;
            } else if (s > (double)(T)(mu / xi) * r) {
                rsf = mlt_is_t ? (double)(T)(T(1) / tau) : 1.0 / mlt_d;
            }
            rho_new = (double)(T)((T)rho * (T)rsf);
            u_scale_new = 1.0 / rsf;
        }
        stop = (r < epri && s < edua) ? 1 : 0;
    }
    if (rec) {
        rec->r = r;
        rec->s = s;
        rec->epri = epri;
        rec->edua = edua;
        rec->stop = stop;
        rec->ticks = sa_wall_clock() - c->t0;
    }
    c->rho = rho_new;
    c->u_scale = u_scale_new;
    c->emitted = c->emit;
    c->k = c->k + 1;
    c->stop = stop;
    c->thr_prev_f = c->thr_f;       // (of the iteration just finished)
    c->thr21_prev_f = c->thr21_f;
    admm_ctl_derive(c);
    if (rec) {
        sa_fence_system();
        rec->seq = index + 1;
        sa_fence_system();
    }
}

// The fixed-order reduction of finalize_kernel (csc_kernels.hip) for value i of a group of
// block partials, by the first kFinalizeThreads threads of a workgroup (all threads of the
// workgroup call it: it synchronises).  scratch: kFinalizeThreads doubles.
constexpr int kFinalizeThreads = 256;
__device__ __forceinline__ double finalize_value_dev(const double *partials, int nblocks, int stride,
                                                    int i, double *scratch) {
    const int t = threadIdx.x;
    if (t < kFinalizeThreads) {
        double s = 0.0;
        for (int b = t; b < nblocks; b += kFinalizeThreads) s = s + partials[(int64_t)b * stride + i];
        scratch[t] = s;
    }
    __syncthreads();
    for (int w = kFinalizeThreads / 2; w > 0; w >>= 1) {
        if (t < w) scratch[t] = scratch[t] + scratch[t + w];
        __syncthreads();
    }
    const double r = scratch[0];
    __syncthreads();
    return r;
}

}  // namespace sporco_amd
