// DPP row-broadcast helpers for 16 x 16 tile recurrences kept in registers (gfx950): row owner = lane & 15, the tile replicated in
// (or four tiles spread over) the four 16-lane rows of a wave.  Used by potrf_diag128_kernel (kernels_diag.hip) and by the unit-lower
// tile inverses at the end of ep_block_kernel (gpcore_ep.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace {

// d -= a(lane C of this lane's 16-lane row) * b: ONE DPP instruction (row_newbcast on a 64-bit VALU op) where the portable form
// costs two v_readlane + an FMA.  A DPP read needs 2 wait states after a VALU write of the same VGPR and the hazard recogniser
// does not look inside inline asm: the FIRST use of a freshly written `a` goes through fnma_bcast_first, which carries the s_nop
// and passes `a` through as an output, so that every later use depends on it and cannot be scheduled ahead of it.
template <int C>
__device__ __forceinline__ void fnma_bcast_first(double &d, double &a, double b) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d), "+v"(a) : "v"(b), "n"(C));
}
template <int C>
__device__ __forceinline__ void fnma_bcast(double &d, double a, double b) {
    asm("v_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(a), "v"(b), "n"(C));
}
// lane C of this lane's 16-lane row, copied to every lane of the row
template <int C>
__device__ __forceinline__ double bcast_row(double a) {
    double d;
    asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(a), "n"(C));
    return d;
}
// trailing columns of step J of the 16 x 16 tile factorisation: row[c] -= row[J] * L(c, J), c = C .. 15
template <int J, int C>
__device__ __forceinline__ void tile_update_cols(double (&row)[16], double lj) {
    if constexpr (C < 16) {
        fnma_bcast<C>(row[C], lj, lj);
        tile_update_cols<J, C + 1>(row, lj);
    }
}
// step J of the inverse, right-looking: s[rr] -= L(rr, J) x[J], rr = RR .. 15, with L(rr, J) = row[J] of lane rr
template <int J, int RR>
__device__ __forceinline__ void tile_inv_update(double (&s)[16], double lj, double xj) {
    if constexpr (RR < 16) {
        fnma_bcast<RR>(s[RR], lj, xj);
        tile_inv_update<J, RR + 1>(s, lj, xj);
    }
}
// X = L^-1 for a UNIT lower 16 x 16 tile, right-looking: x[J] = s[J], then s[rr] -= L(rr, J) x[J] for rr > J.  row[k] = L(lane's row, k)
// for k < the lane's row (anything elsewhere); s starts as the lane's column of the identity (column owner = lane & 15).
template <int J>
__device__ __forceinline__ void tile_unit_inverse(double (&row)[16], double (&s)[16], double (&x)[16]) {
    if constexpr (J < 16) {
        x[J] = s[J];
        if constexpr (J < 15) {
            double lj = row[J];
            fnma_bcast_first<J + 1>(s[J + 1], lj, x[J]);
            tile_inv_update<J, J + 2>(s, lj, x[J]);
        }
        tile_unit_inverse<J + 1>(row, s, x);
    }
}

}  // namespace
