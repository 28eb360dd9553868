// csc_fused_body.h -- the column pass of csc_fused.hip (column FFT + Sherman-Morrison + column
// IFFT of one tile in registers) as a device function: the body of fused_cols_kernel, and the
// middle phase of the one-launch solve of small problems (csc_rows.hip: admm_persist_kernel).
// See csc_fused.hip for the method.
#pragma once

#include "csc_fused.h"

#include "regfft.h"

#include <type_traits>
#include <utility>

namespace sporco_amd {

namespace {

using namespace regfft;

constexpr int kExchUnitsMax = 16384;  // f2 units of the largest exchange buffer (128 KiB)
// LDS of one instantiation: its exchange group (LP lines of NW points for every wave) + scratch
constexpr size_t fused_lds_bytes(int NW, int LP) {
    return sizeof(f2) * LP * NW * NW * 64 + sizeof(double) * 2 * 16;   // + block_sum scratch
}

// N1 x NW = H; NW waves; KC: compile-time filter count (64), or 0 for a run-time K <= 64.
// GRAD: the diagonal Sherman-Morrison form of ConvBPDNGradReg (cbpdn.py:1163-1175,
// linalg.py:300-366) with dd = mu wg_k (ghh[f] + ghw[wf]) + rho per element:
//     coef = (Sf - rho sum_k Df yuf / dd) / g1,    g1 = 1 + sum_k |Df|^2 / dd  (table)
//     xf = (rho yuf + conj(Df) coef) / dd,         Df.xf - Sf = -coef
// and a second partial per tile, the weighted sum of wg GHGf |xf|^2 (cbpdn.py:1204-1214).
// KRT (with KC = 64): the kernel owns 64 filters of rows that are a.K > 64 filters long
// (FusedColsArgs::Kv): run-time row stride, all lanes valid, multipliers stored.
// PER_TILE (with KC = 64): the D-side operands (dft, gramt) are those of the tile, not of its
// row frequency (FusedColsArgs::per_tile).
#ifdef SPORCO_AMD_HOSTSIM
#define SA_TS(i)
#else
#define SA_TS(i)                                                                  \
    if constexpr (DBG >= 3) {                                                      \
        unsigned long long t_;                                                     \
        asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_));            \
        ts[i] = t_;                                                                \
    }
#endif
// DBG (measurement builds only, SPORCO_AMD_COLS_DEBUG): 1 = loads and stores only, 2 = no tile
// loads / stores (arithmetic, exchanges and operand loads only).
// PERSIST: the workgroup walks its XCD's tile list (below).  AOFF >= 0: `a` is the kernel's
// argument (at that byte offset of the argument block, sa_args_reload).  AOFF < 0: the arguments
// are read through `afix` instead -- a copy in device memory (admm_persist_kernel: one per
// workgroup, each with its own control block); no stop test and no start-up stagger here then.
// The statements live in csc_fused_body.inc and are included into fused_cols_kernel
// (csc_fused.hip) and into this function.
template <int N1, int NW, int LPARAM, int KC, bool GRAD, bool KRT, bool PER_TILE, int DBG,
          bool PERSIST, int AOFF>
__device__ __forceinline__ void fused_cols_body(const FusedColsArgs<float> &a,
                                                SA_ARGS_PTR_T(FusedColsArgs<float>) afix) {
#include "csc_fused_body.inc"
}

}  // namespace

}  // namespace sporco_amd
