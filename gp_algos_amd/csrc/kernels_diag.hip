// Critical-path kernels of the blocked fp64 Cholesky and triangular solves, MFMA-blocked at 16 (gfx950).
//
//   potrf_diag128 : Cholesky of one 128x128 diagonal block held in LDS.  Per 16-column micro-panel:
//                   the 16x16 diagonal tile is factored and inverted in registers by one wave
//                   (v_readlane broadcasts, no LDS round trips), the panel below is multiplied by the
//                   tile inverse on the matrix cores, and the rest of the block is updated with
//                   v_mfma_f64_16x16x4_f64 straight out of LDS.  Emits the 8 tile inverses ("dinv").
//   trsm_panel128 : X (M x 128) <- X * Lkk^-T.  64 rows per workgroup, X strip in LDS, blocked forward
//                   substitution: GEMM part on MFMA (L fragments from L2), 16x16 solves by the tile
//                   inverses, feeding the accumulator registers directly back as the next B operand.
//   fwd/bwd_step  : one block step of t <- L^-1 t / alpha <- L^-T t: every workgroup re-solves the
//                   128x128 diagonal system in LDS (cheap) and applies its share of the panel update.
//
// Reference semantics: breeze.linalg.cholesky (GpPredictor.scala:120), MatrixUtils.forwardSolve /
// backSolve (MatrixUtils.scala:17-35,115-133).  Only the 16x16 tile inverses are explicit inverses;
// everything larger is substitution, so the backward error stays at the substitution level.
#include "gpcore_internal.h"
#include "dpp_tile.h"
#include "trsm_tile.h"

namespace {

constexpr int NB = GP_NB;   // 128
constexpr int XS = TRSM_XS;
constexpr int PLS = 144;    // LDS column stride of the 128x128 block: 1152 B = 128 (mod 256) -> conflict-free fragments
constexpr int LS1 = NB + 1;
#ifndef POTRF_DPP
#define POTRF_DPP 1   // 16 x 16 tile factorisation with DPP row broadcasts (0: v_readlane / LDS-broadcast form)
#endif
#ifndef POTRF_NR2
#define POTRF_NR2 0   // second Newton step on v_rsq_f64 in the tile pivots: not needed, the sqrt and reciprocal corrections that follow are Newton steps themselves
#endif

__device__ __forceinline__ double rl64(double v, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Column J of the 16 x 16 tile Cholesky (row owner = lane & 15, replicated in the four 16-lane rows of the wave) together
// with step J of the tile inverse X = L^-1 (column owner = lane & 15).  Both are chains of DEPENDENT fp64 operations
// (~16 cycles each on this hardware, measured: the factorisation alone 219 cycles per column, the inverse as much again when it
// ran afterwards), so (i) the pivot chain is as short as it can be made: v_rsq_f64 and two Newton steps give rs = 1/sqrt(a_jj) to
// an ulp or two, the column is scaled by rs directly (no divide, no dependence on the rounded diagonal), and the diagonal
// sqrt(a_jj) = a_jj rs with its correction is computed beside the chain; (ii) the inverse is written right-looking,
// x[J] = s[J] rs and s[rr] -= L(rr, J) x[J], which needs nothing but column J -- its instructions fill the issue slots the pivot
// chain leaves empty; (iii) every cross-lane operand is a DPP row broadcast folded into the FMA (fnma_bcast).
// The diagonal needs no special case: for lane J, row[J] = a_jj, so a_jj rs is sqrt(a_jj) to an ulp or two.
// A non-positive pivot is not replaced: 1/sqrt turns it into NaN or +-inf, which reaches `guard` (the running sum of the rs),
// the caller tests that ONE number, and only then looks for the first offender among the pivots kept in piv[] -- the common
// path carries one add per column instead of a compare-and-select chain.
template <int J>
__device__ __forceinline__ void tile_chol_inv_cols(double (&row)[16], double (&s)[16], double (&x)[16], double (&piv)[16], double &guard) {
    if constexpr (J < 16) {
        const double ajj = bcast_row<J>(row[J]);
        piv[J] = ajj;
        double rs = __builtin_amdgcn_rsq(ajj);
        double e = fma(-(ajj * rs), rs, 1.0);
        rs = fma(0.5 * rs, e, rs);
        const double t = ajj * rs;               // ~ sqrt(a_jj)
        e = fma(-t, rs, 1.0);
        rs = fma(0.5 * rs, e, rs);
        guard += rs;
        double lj = row[J] * rs;                 // L(i, J) for the rows below the diagonal (lanes i > J), sqrt(a_jj) on lane J
        x[J] = s[J] * rs;
        if constexpr (J < 15) {
            fnma_bcast_first<J + 1>(row[J + 1], lj, lj);     // the next pivot column first
            tile_update_cols<J, J + 2>(row, lj);
            tile_inv_update<J, J + 1>(s, lj, x[J]);
        }
        row[J] = lj;
        tile_chol_inv_cols<J + 1>(row, s, x, piv, guard);
    }
}

// 128x128 block global -> LDS with 16-byte loads, 8 in flight per thread (a serial load/wait/ds_write loop
// costs ~45 us for this block; batched it is ~2 us).  LOWER: zero the strict upper triangle on the way in.
template <int STRIDE, bool LOWER, int INFLIGHT = 8>
__device__ __forceinline__ void load_block_lds(double *a, const double *__restrict__ A, int lda, int tid) {
#pragma unroll
    for (int q0 = 0; q0 < 32; q0 += INFLIGHT) {
        double2_t v[INFLIGHT];
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) {
            const int e = tid + 256 * (q0 + q), i = (e & 63) * 2, j = e >> 6;
            v[q] = *reinterpret_cast<const double2_t *>(A + i + (size_t)j * lda);
        }
#pragma unroll
        for (int q = 0; q < INFLIGHT; ++q) {
            const int e = tid + 256 * (q0 + q), i = (e & 63) * 2, j = e >> 6;
            a[i + j * STRIDE] = (!LOWER || i >= j) ? v[q].x : 0.0;
            a[i + 1 + j * STRIDE] = (!LOWER || i + 1 >= j) ? v[q].y : 0.0;
        }
    }
}

#ifdef POTRF_STAMPS   // tools/lab/potrf_lab.hip: cycle stamps of the phases of one launch (never defined in the library build)
__device__ unsigned long long potrf_stamps[256];
#define STAMP(slot) do { if (lane == 0) potrf_stamps[(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

// -------------------------------------------------------------------------------------------------
constexpr int POTRF_WAVES = 8;   // wave 0: the serial chain; the others: panel, update and stores (two waves per SIMD hide each other's LDS / MFMA latency)
// The factorisation proper of potrf_diag128_kernel: the 128 x 128 block is in LDS (a, column stride PLS; the
// part on and below the diagonal is enough), every wave's share of it has landed before the call (the first barrier inside orders
// the waves).  Writes L (zeros above the diagonal) and the eight tile inverses to global memory as it goes.
__device__ __forceinline__ void potrf_block_body(double *a, double *dv, int *flag, double *__restrict__ A, int lda, double *__restrict__ dinv,
                                                 int *info, int base, int tid, int lane, int wave, int fr, int fg) {
    // Two barriers per micro-panel.  Wave 0 owns the serial part (tile Cholesky + tile inverse) and, of the update that follows a
    // panel, only the ONE tile it needs next (the following diagonal tile); the other 3 waves apply the rest of that update and
    // write the finished 16 columns out while wave 0 is already factoring -- everybody meets again at the barrier after the
    // factorisation.
    for (int jb = 0; jb < NB / 16; ++jb) {
        const int c0 = jb * 16;
        if (wave == 0) {
            // ---- 16x16 tile: Cholesky (row owner = lane&15) and inverse (column owner = lane&15) in one pass ----
            double row[16], x[16], sv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) row[c] = a[(c0 + fr) + (c0 + c) * PLS];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) sv[rr] = (rr == fr) ? 1.0 : 0.0;
            double piv[16], guard = 0.0;
            STAMP(8 + 8 * jb + 0);
            tile_chol_inv_cols<0>(row, sv, x, piv, guard);
            STAMP(8 + 8 * jb + 1);
            if (lane < 16) {
                // (what lands above the diagonal of the tile is never read: the stores to global memory mask it)
#pragma unroll
                for (int c = 0; c < 16; ++c) a[(c0 + fr) + (c0 + c) * PLS] = row[c];
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) dv[rr + 16 * fr] = x[rr];
            }
            if (!(fabs(guard) < HUGE_VAL)) {     // a pivot was <= 0 or NaN (rare): which one came first
                int bad = 0;
#pragma unroll
                for (int j = 15; j >= 0; --j) if (!(piv[j] > 0.0)) bad = j + 1;
                if (lane == 0) { atomicCAS(info, 0, base + c0 + bad); *flag = 1; }
            }
            STAMP(8 + 8 * jb + 2);
        }
        __syncthreads();
        if (wave == 0) STAMP(8 + 8 * jb + 3);
        if (*flag) break;
        // ---- panel below the tile:  P <- A_below * invD^T, computed as (invD * A_below^T) so rows stay on the lane ----
        const int T = NB / 16 - 1 - jb;
        if (wave < T) {    // T <= 7 tiles, one per wave
            const int r0 = c0 + 16 + 16 * wave;
            double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double aop = dv[fr + 16 * (4 * ks + fg)];
                const double bop = a[(r0 + fr) + (c0 + 4 * ks + fg) * PLS];
                acc = MFMA(aop, bop, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) a[(r0 + fr) + (c0 + fg + 4 * r) * PLS] = acc[r];
        }
        if (wave == POTRF_WAVES - 1) {   // the tile inverse also goes out to global memory (the panel solves of the other kernels): off wave 0's chain
#pragma unroll
            for (int q = 0; q < 4; ++q) dinv[jb * 256 + lane + 64 * q] = dv[lane + 64 * q];
        }
        if (wave == 0) STAMP(8 + 8 * jb + 4);
        __syncthreads();
        if (wave == 0) STAMP(8 + 8 * jb + 5);
        // ---- rest of the block:  A(ti,tj) -= P_ti P_tj^T  for jb < tj <= ti; wave 0 only the next diagonal tile ----
        const int ntile = T * (T + 1) / 2;
        if (wave == 0) {
            if (ntile > 0) {
                const int ri = c0 + 16;
                double4_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = a[(ri + fr) + (ri + fg + 4 * r) * PLS];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const double bop = a[(ri + fr) + (c0 + 4 * ks + fg) * PLS];
                    acc = MFMA(-bop, bop, acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) a[(ri + fr) + (ri + fg + 4 * r) * PLS] = acc[r];
            }
        } else {
            for (int q = wave; q < ntile; q += POTRF_WAVES - 1) {
                int bi, bj;
                tri_coords(q, bi, bj);
                const int ri = c0 + 16 + 16 * bi, rj = c0 + 16 + 16 * bj;
                double4_t acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = a[(ri + fr) + (rj + fg + 4 * r) * PLS];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const double aop = -a[(rj + fr) + (c0 + 4 * ks + fg) * PLS];
                    const double bop = a[(ri + fr) + (c0 + 4 * ks + fg) * PLS];
                    acc = MFMA(aop, bop, acc);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) a[(ri + fr) + (rj + fg + 4 * r) * PLS] = acc[r];
            }
            // columns c0 .. c0+15 are final (diagonal tile before the first barrier, the panel before the second): out they go,
            // zeros above the diagonal, while wave 0 factors the next tile
            for (int e = tid - 64; e < 16 * 64; e += 64 * (POTRF_WAVES - 1)) {
                const int i = (e & 63) * 2, j = c0 + (e >> 6);
                double2_t v;
                v.x = (i >= j) ? a[i + j * PLS] : 0.0;
                v.y = (i + 1 >= j) ? a[i + 1 + j * PLS] : 0.0;
                *reinterpret_cast<double2_t *>(A + i + (size_t)j * lda) = v;
            }
        }
        if (wave == 0) STAMP(8 + 8 * jb + 6);
        if (wave == 1) STAMP(128 + jb);
    }
}

__global__ __launch_bounds__(64 * POTRF_WAVES) void potrf_diag128_kernel(double *__restrict__ A, int lda, double *__restrict__ dinv,
                                                            int *info, int base, gp_batch bt) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    A += (size_t)blockIdx.x * bt.s0;      // one workgroup per problem of a lockstep batch
    dinv += (size_t)blockIdx.x * bt.s1;
    info += blockIdx.x;
    double *a = sm;                 // NB x PLS
    double *dv = sm + NB * PLS;     // 16 x 16 inverse of the current diagonal tile, (c,k) at c + 16k
    int *flag = reinterpret_cast<int *>(dv + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    if (tid == 0) *flag = 0;
    if (wave == 0) STAMP(0);
    // Load by LDS-DMA: one wave instruction moves one 128-double column (64 lanes x 16 B, contiguous in LDS, the pad follows it),
    // no staging VGPRs, everything in flight at once.  This kernel is a chain link -- nothing else hides its loads, and one CU
    // pulls ~50 GB/s, so (i) only the part on and below the diagonal is fetched (lanes above it are masked off; nothing below
    // reads the strict upper triangle: the tile factorisation writes zeros above the diagonal of the diagonal tiles, panel and
    // update steps only touch tiles on or below the diagonal, the stores mask i < j), and (ii) wave 0 fetches just the first
    // 16 columns and starts factoring while the other three waves bring in the rest.
    {
        const double *src = A + lane * 2;
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (lane * 2 + 1 >= j) __builtin_amdgcn_global_load_lds(src + (size_t)j * lda, a + j * PLS, 16, 0, 0);
        } else {
            for (int j = 16 + (wave - 1); j < NB; j += POTRF_WAVES - 1)
                if (lane * 2 + 1 >= j) __builtin_amdgcn_global_load_lds(src + (size_t)j * lda, a + j * PLS, 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's columns are in LDS
    }
    if (wave == 0) STAMP(1);
    potrf_block_body(a, dv, flag, A, lda, dinv, info, base, tid, lane, wave, fr, fg);
    if (wave == 0) { STAMP(2); STAMP(3); }
}

// -------------------------------------------------------------------------------------------------
template <bool SCALED_COPY>
__global__ __launch_bounds__(256, 2) void trsm_panel128_kernel(double *__restrict__ X, int ldx, const double *__restrict__ L, int ldl,
                                                             const double *__restrict__ dinv, double *__restrict__ sumsq,
                                                             const double *__restrict__ tvec, double *__restrict__ dots, gp_batch bt,
                                                             double *__restrict__ X2, const double *__restrict__ cs2) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // NB x XS
    if (bt.tri && (int)blockIdx.x * 64 >= ((int)blockIdx.y + 1) * NB) return;   // problem g only has (g+1)*128 non-zero rows
    X += (size_t)blockIdx.y * bt.s0;      // blockIdx.y = problem of a lockstep batch
    L += (size_t)blockIdx.y * bt.s1;
    dinv += (size_t)blockIdx.y * bt.s2;
    if (X2) X2 += (size_t)blockIdx.y * bt.s3;
    if (cs2) cs2 += (size_t)blockIdx.y * bt.s4;
    if (tvec) tvec += (size_t)blockIdx.y * bt.s4;
    if (dots) dots += (size_t)blockIdx.y * bt.s5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int row0 = blockIdx.x * 64;
    double *Xg = X + row0;
    // strip load: 64 rows x 128 columns, 16 B per lane, 8 loads in flight
    const int li = (tid & 31) * 2, lc = tid >> 5;   // row pair, column (+8 per step)
#pragma unroll
    for (int q0 = 0; q0 < 16; q0 += 8) {
        double2_t v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const double2_t *>(Xg + li + (size_t)(lc + 8 * (q0 + q)) * ldx);
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<double2_t *>(xs + (lc + 8 * (q0 + q)) * XS + li) = v[q];
    }
    double fa[28], fb[28];
    trsm_load_frags<1>(L, ldl, fr, fg, fa);
    __syncthreads();
    const int sp = wave * 16 + fr;   // this lane's row inside the strip
    double ss = 0.0;
    // L fragments of the next chunk are requested before the current chunk's MFMAs (ping-pong fa/fb)
    trsm_load_frags<2>(L, ldl, fr, fg, fb);
    trsm_chunk<0>(xs, sp, fr, fg, fa, dinv, ss);
    trsm_chunk<1>(xs, sp, fr, fg, fa, dinv, ss);
    trsm_load_frags<3>(L, ldl, fr, fg, fa);
    trsm_chunk<2>(xs, sp, fr, fg, fb, dinv, ss);
    trsm_load_frags<4>(L, ldl, fr, fg, fb);
    trsm_chunk<3>(xs, sp, fr, fg, fa, dinv, ss);
    trsm_load_frags<5>(L, ldl, fr, fg, fa);
    trsm_chunk<4>(xs, sp, fr, fg, fb, dinv, ss);
    trsm_load_frags<6>(L, ldl, fr, fg, fb);
    trsm_chunk<5>(xs, sp, fr, fg, fa, dinv, ss);
    trsm_load_frags<7>(L, ldl, fr, fg, fa);
    trsm_chunk<6>(xs, sp, fr, fg, fb, dinv, ss);
    trsm_chunk<7>(xs, sp, fr, fg, fa, dinv, ss);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q)
        *reinterpret_cast<double2_t *>(Xg + li + (size_t)(lc + 8 * q) * ldx) = *reinterpret_cast<const double2_t *>(xs + (lc + 8 * q) * XS + li);
    if (SCALED_COPY) {   // second copy with column c scaled by cs2[c] (EP: S diag(c), the other operand of the rank-128 update)
        __builtin_amdgcn_sched_barrier(0);   // keep these loads below the solve: hoisted, they push the kernel into scratch
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            double2_t v = *reinterpret_cast<const double2_t *>(xs + (lc + 8 * q) * XS + li);
            const double sc = cs2[lc + 8 * q];
            v.x *= sc, v.y *= sc;
            *reinterpret_cast<double2_t *>(X2 + row0 + li + (size_t)(lc + 8 * q) * ldx) = v;
        }
    }
    if (dots && tid < 64) {   // fused row dot with this block's slice of t = L^-1 y: the posterior mean  V^T (L^-1 y)
        double acc = 0.0;
#pragma unroll 8
        for (int c = 0; c < NB; ++c) acc = fma(xs[c * XS + tid], tvec[c], acc);
        dots[row0 + tid] += acc;
    }
    if (sumsq) {
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (fg == 0) sumsq[row0 + sp] += ss;
    }
}

// -------------------------------------------------------------------------------------------------
// The link between two site blocks of an EP sweep (EpParameterEstimator.scala:52-54), ONE workgroup: the 128 rows of the delayed
// columns that belong to the next block, X <- X Lmat^-T (the row-panel solve above on a 128-row strip, eight waves x 16 rows),
// their scaled copy X2 = X diag(c), the mean update dots += X tvec, and -- from the strip still in LDS -- the one 128 x 128 tile
// of the rank-128 update the next block kernel reads, D(lower) -= X2 X^T.  As two launches (panel solve on 2 workgroups, tile
// update on 36) the second re-read what the first had just written through L2/HBM, 4 us alone and 20+ us while the other
// streams' GEMMs load the memory system; fused, the chain between two block kernels is one launch shorter as well.
constexpr int LKS = 144;   // LDS column stride of the 128-row strip: 1152 B = 128 (mod 256)
__global__ __launch_bounds__(512) void ep_link_kernel(double *__restrict__ X, int ldx, const double *__restrict__ L, const double *__restrict__ dinv,
                                                      const double *__restrict__ tvec, double *__restrict__ dots, double *__restrict__ X2,
                                                      const double *__restrict__ cs2, double *__restrict__ D, int ldd) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // NB x LKS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int li = (tid & 63) * 2, lc = tid >> 6;   // row pair, column (+8 per step)
    {   // strip by LDS-DMA: one wave instruction = one 128-row column (1 KiB), 16 per wave, all in flight, no staging registers
        const double *src = X + lane * 2 + (size_t)wave * ldx;
#pragma unroll
        for (int q = 0; q < NB / 8; ++q) __builtin_amdgcn_global_load_lds(src + (size_t)(8 * q) * ldx, xs + (wave + 8 * q) * LKS, 16, 0, 0);
    }
    // this wave's tiles of D (q = wave, wave + 8, ...; at most 5): fetched now, under the solve -- they do not depend on it
    double4_t dacc[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = wave + 8 * u;
        int I = 0, J = 0;
        if (q < 36) tri_coords(q, I, J);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) dacc[u][rr] = (q < 36) ? D[(16 * I + fr) + (size_t)(16 * J + fg + 4 * rr) * ldd] : 0.0;
    }
    double fa[28], fb[28];
    trsm_load_frags<1>(L, NB, fr, fg, fa);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMA has landed (and the loads above)
    __syncthreads();
    const int sp = wave * 16 + fr;   // this lane's row inside the strip
    double ss = 0.0;
    trsm_load_frags<2>(L, NB, fr, fg, fb);
    trsm_chunk<0, LKS>(xs, sp, fr, fg, fa, dinv, ss);
    trsm_chunk<1, LKS>(xs, sp, fr, fg, fa, dinv, ss);
    trsm_load_frags<3>(L, NB, fr, fg, fa);
    trsm_chunk<2, LKS>(xs, sp, fr, fg, fb, dinv, ss);
    trsm_load_frags<4>(L, NB, fr, fg, fb);
    trsm_chunk<3, LKS>(xs, sp, fr, fg, fa, dinv, ss);
    trsm_load_frags<5>(L, NB, fr, fg, fa);
    trsm_chunk<4, LKS>(xs, sp, fr, fg, fb, dinv, ss);
    trsm_load_frags<6>(L, NB, fr, fg, fb);
    trsm_chunk<5, LKS>(xs, sp, fr, fg, fa, dinv, ss);
    trsm_load_frags<7>(L, NB, fr, fg, fa);
    trsm_chunk<6, LKS>(xs, sp, fr, fg, fb, dinv, ss);
    trsm_chunk<7, LKS>(xs, sp, fr, fg, fa, dinv, ss);
    (void)ss;
    __syncthreads();
    // the tile update first (the next block kernel waits for it), 36 lower 16 x 16 tiles over the eight waves, K = 128
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = wave + 8 * u;
        if (q >= 36) break;
        int I, J;
        tri_coords(q, I, J);
        double4_t acc = dacc[u];
#pragma unroll 8
        for (int ks = 0; ks < 32; ++ks) {
            const int k = 4 * ks + fg;
            const double aop = -(cs2[k] * xs[k * LKS + 16 * J + fr]);
            const double bop = xs[k * LKS + 16 * I + fr];
            acc = MFMA(aop, bop, acc);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            if (16 * I + fr >= 16 * J + fg + 4 * rr) D[(16 * I + fr) + (size_t)(16 * J + fg + 4 * rr) * ldd] = acc[rr];
    }
    // the solved rows, their scaled copy and the mean update
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        double2_t v = *reinterpret_cast<const double2_t *>(xs + (lc + 8 * q) * LKS + li);
        *reinterpret_cast<double2_t *>(X + li + (size_t)(lc + 8 * q) * ldx) = v;
        const double sc = cs2[lc + 8 * q];
        v.x *= sc, v.y *= sc;
        *reinterpret_cast<double2_t *>(X2 + li + (size_t)(lc + 8 * q) * ldx) = v;
    }
    if (tid < NB) {
        double acc = 0.0;
#pragma unroll 8
        for (int c = 0; c < NB; ++c) acc = fma(xs[c * LKS + tid], tvec[c], acc);
        dots[tid] += acc;
    }
}


// -------------------------------------------------------------------------------------------------
// Solve the 128x128 diagonal system in LDS with the tile inverses.  a = L_kk (NB x LS1), v = rhs (NB), in place.
// trans = 0: L x = b (chunks ascending);  trans = 1: L^T x = b (chunks descending).  128 threads take part.
__device__ __forceinline__ void diag_solve_lds(const double *a, const double *dv, double *v, double *tmp, int tid, int trans) {
    for (int s = 0; s < NB / 16; ++s) {
        const int cb = trans ? (NB / 16 - 1 - s) : s;
        const int c0 = cb * 16;
        if (tid < 16) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const double m = trans ? dv[cb * 256 + k + 16 * tid] : dv[cb * 256 + tid + 16 * k];
                acc = fma(m, v[c0 + k], acc);
            }
            tmp[tid] = acc;
        }
        __syncthreads();
        if (tid < 16) v[c0 + tid] = tmp[tid];
        if (tid < NB) {
            if (!trans && tid >= c0 + 16) {
                double acc = v[tid];
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fma(-a[tid + (c0 + k) * LS1], tmp[k], acc);
                v[tid] = acc;
            } else if (trans && tid < c0) {
                double acc = v[tid];
#pragma unroll
                for (int k = 0; k < 16; ++k) acc = fma(-a[(c0 + k) + tid * LS1], tmp[k], acc);
                v[tid] = acc;
            }
        }
        __syncthreads();
    }
}

// forward step k:  x_k = L_kk^-1 t_k  (written to sol);  t[(k+1)*128 + p] -= L21(p,:) x_k  for this workgroup's rows
__global__ __launch_bounds__(256) void fwd_step_kernel(const double *__restrict__ L, int ldl, const double *__restrict__ dinv,
                                                       double *__restrict__ t, double *__restrict__ sol, int k0, int r) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *a = sm, *dv = sm + NB * LS1, *v = dv + 8 * 256, *tmp = v + NB;
    const int tid = threadIdx.x;
    const double *Lkk = L + k0 + (size_t)k0 * ldl;
    load_block_lds<LS1, false>(a, Lkk, ldl, tid);
    for (int e = tid; e < 8 * 256; e += 256) dv[e] = dinv[e];
    if (tid < NB) v[tid] = t[k0 + tid];
    __syncthreads();
    diag_solve_lds(a, dv, v, tmp, tid, 0);
    if (blockIdx.x == 0 && tid < NB) sol[k0 + tid] = v[tid];
    const int p = blockIdx.x * 256 + tid;
    if (p < r) {
        const double *Lp = L + (k0 + NB + p) + (size_t)k0 * ldl;
        double acc = 0.0;
#pragma unroll 8
        for (int c = 0; c < NB; ++c) acc = fma(Lp[(size_t)c * ldl], v[c], acc);
        t[k0 + NB + p] -= acc;
    }
}

// backward step k:  x_k = L_kk^-T t_k  (written to sol);  t[c] -= sum_r L(k0+r, c) x_k[r]  for this workgroup's 64 columns c < k0
__global__ __launch_bounds__(256) void bwd_step_kernel(const double *__restrict__ L, int ldl, const double *__restrict__ dinv,
                                                       double *__restrict__ t, double *__restrict__ sol, int k0) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *a = sm, *dv = sm + NB * LS1, *v = dv + 8 * 256, *tmp = v + NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *Lkk = L + k0 + (size_t)k0 * ldl;
    load_block_lds<LS1, false>(a, Lkk, ldl, tid);
    for (int e = tid; e < 8 * 256; e += 256) dv[e] = dinv[e];
    if (tid < NB) v[tid] = t[k0 + tid];
    __syncthreads();
    diag_solve_lds(a, dv, v, tmp, tid, 1);
    if (blockIdx.x == 0 && tid < NB) sol[k0 + tid] = v[tid];
    const int cbase = blockIdx.x * 64;
    if (cbase >= k0) return;
    const double x0 = v[lane], x1 = v[lane + 64];
    for (int cc = wave; cc < 64; cc += 4) {
        const int c = cbase + cc;
        const double *Lc = L + k0 + (size_t)c * ldl;
        double pr = fma(Lc[lane], x0, Lc[lane + 64] * x1);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pr += __shfl_xor(pr, o);
        if (lane == 0) t[c] -= pr;
    }
}

// inverses of the 16x16 diagonal tiles of a given lower-triangular L (for callers that bring their own factor)
__global__ __launch_bounds__(64) void tile_inverse_kernel(const double *__restrict__ L, int ldl, double *__restrict__ dinv) {
    const int lane = threadIdx.x, fr = lane & 15;
    const int c0 = blockIdx.x * 16;
    double row[16], x[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = L[(c0 + fr) + (size_t)(c0 + c) * ldl];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        double s = (rr == fr) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) s = fma(-rl64(row[k], rr), x[k], s);
        x[rr] = s / rl64(row[rr], rr);
    }
    if (lane < 16) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) dinv[(size_t)blockIdx.x * 256 + rr + 16 * fr] = x[rr];
    }
}

}  // namespace

static constexpr int POTRF_LDS = (NB * PLS + 256 + 8) * (int)sizeof(double);
static constexpr int TRSM_LDS = NB * XS * (int)sizeof(double);
static constexpr int STEP_LDS = (NB * LS1 + 8 * 256 + NB + 16) * (int)sizeof(double);
static constexpr int LINK_LDS = NB * LKS * (int)sizeof(double);

int gpk_init_diag_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_diag128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(trsm_panel128_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(trsm_panel128_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(ep_link_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LINK_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, STEP_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_step_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, STEP_LDS);
    return e == hipSuccess ? 0 : 1;
}

void gpk_potrf_diag128(hipStream_t s, double *A, int lda, double *dinv, int *d_info, int base, gp_batch bt) {
    if (bt.count <= 0) return;
    hipLaunchKernelGGL(potrf_diag128_kernel, dim3(bt.count), dim3(64 * POTRF_WAVES), POTRF_LDS, s, A, lda, dinv, d_info, base, bt);
}
void gpk_trsm_panel128(hipStream_t s, double *X, int M, int ldx, const double *Lkk, int ldl, const double *dinv, double *sumsq,
                       const double *tvec, double *dots, gp_batch bt, double *X2, const double *cs2) {
    if (M <= 0 || bt.count <= 0) return;
    if (X2)
        hipLaunchKernelGGL(trsm_panel128_kernel<true>, dim3(M / 64, bt.count), dim3(256), TRSM_LDS, s, X, ldx, Lkk, ldl, dinv, sumsq, tvec, dots, bt, X2, cs2);
    else
        hipLaunchKernelGGL(trsm_panel128_kernel<false>, dim3(M / 64, bt.count), dim3(256), TRSM_LDS, s, X, ldx, Lkk, ldl, dinv, sumsq, tvec, dots, bt, X2, cs2);
}
void gpk_ep_link(hipStream_t s, double *X, int ldx, const double *Lmat, const double *dinv, const double *tvec, double *dots, double *X2,
                 const double *cs2, double *D, int ldd) {
    hipLaunchKernelGGL(ep_link_kernel, dim3(1), dim3(512), LINK_LDS, s, X, ldx, Lmat, dinv, tvec, dots, X2, cs2, D, ldd);
}
void gpk_fwd_step(hipStream_t s, const double *L, int ldl, const double *dinv_k, double *t, double *sol, int k0, int r) {
    int grid = r > 0 ? (r + 255) / 256 : 1;
    hipLaunchKernelGGL(fwd_step_kernel, dim3(grid), dim3(256), STEP_LDS, s, L, ldl, dinv_k, t, sol, k0, r);
}
void gpk_bwd_step(hipStream_t s, const double *L, int ldl, const double *dinv_k, double *t, double *sol, int k0) {
    int grid = k0 > 0 ? k0 / 64 : 1;
    hipLaunchKernelGGL(bwd_step_kernel, dim3(grid), dim3(256), STEP_LDS, s, L, ldl, dinv_k, t, sol, k0);
}
void gpk_tile_inverses(hipStream_t s, const double *L, int np, int ldl, double *dinv) {
    hipLaunchKernelGGL(tile_inverse_kernel, dim3(np / 16), dim3(64), 0, s, L, ldl, dinv);
}
