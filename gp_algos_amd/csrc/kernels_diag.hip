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

// -------------------------------------------------------------------------------------------------
// The whole blocked factorisation of ONE matrix as ONE persistent launch (round 4): a task list instead of ~250 launches.
//
// What the launch-per-step form cannot do is run throughput work beside the serial chain at a finer grain than a kernel: the far
// trailing update and the next panel's chain share the chip as two kernels, the chain's whole-chip launches queue behind the
// update's workgroups (2-4x slower), and late in the factorisation the chip idles through every diagonal block.
// Here one workgroup per CU (512 threads, the diagonal-block kernel's 150 KB of LDS) takes tasks off a list, in list order, until the
// list is empty:
//   POTRF(k)            the diagonal block (potrf_block_body, as potrf_diag128_kernel); every block but the first with its link to the
//                       block before as the prologue (mega_potrf_link: the solve of its own rows and the update of its tile that those
//                       rows complete -- the in-panel one, or, for a panel's first block, the previous panel's outer one, whose earlier
//                       columns three SUM tasks have added up ahead of time in a scratch tile);
//   TRSM(k, i)          row block i of block column k:  X <- X L_kk^-T  (128 rows: the strip solve of trsm_panel128 / ep_link);
//   UPD(k0, kb, i, j)   tile (i, j) -= X[i, k0..k0+kb) X[j, k0..k0+kb)^T  with K = 128 kb: the in-panel updates (kb = 1) and the outer
//                       updates (kb = 4) of the two-level scheme -- the SAME products in the SAME order per element as the launches of
//                       chol_blocked, so the factor is bit-identical to the one-stream form; a tile on the chain's way (the next
//                       diagonal block's) is cut into 64 x 64 quarters that four workgroups take (q >= 0).
// A task names the (at most ten) tasks whose results it reads or overwrites; `done[t] == epoch` says task t is in memory.  Publish:
// every wave drains its stores, barrier, one lane: agent-scope release, drain, relaxed flag store.  Consume: one lane polls relaxed
// (bounded: under a second, then *err is set and every workgroup falls through to the end -- a loud GP_EHIP, not a hung queue), agent-scope
// acquire, drain, barrier, plain loads (MI355X_MICROARCH.md: the per-XCD L2s are not coherent with each other).
// No deadlock: tasks are claimed in list order by workgroups that are running, the list is a topological order (checked on the
// host), so everything a claimed task waits for was claimed earlier by a workgroup that is resident and will finish.
// The ORDER of the list is a list schedule computed on the host from a cost model (critical path first: chol_mega_plan): claimed in
// that order the dependencies are mostly met on arrival, and what waits is what the model would have idle anyway.
struct chol_task { int type, k0, kb, i, j, q, dep[10]; };   // 64 bytes; type 0 POTRF (k0), 1 TRSM (k0, i), 2 UPD, 3 a place holder, 4 link + POTRF (k0; q = the place holder it announces, j = slot of the started sum or -1), 5 quarter q of a started sum (rows i, K = 128 kb from block column k0, into slot j)
constexpr int MEGA_THREADS = 512;

// tile (128 x 128 at Cp) -= A (128 x K at Ap) B (128 x K at Bp)^T, all with leading dimension ld: gemm_nt_f64_kernel's products (8 waves
// x 64 x 32, BK = 16) on the workgroup's dynamic LDS.  diag: elements above the diagonal are neither read nor written.  quarter q >= 0:
// only the 64 x 64 quarter (row half q & 1, column half q >> 1) is computed and stored, by all eight waves.
// Staging: two buffers of 32 k-columns each, filled by LDS-DMA one slice ahead, one barrier per slice.  (One workgroup per CU, two waves
// per SIMD, and the tasks of 256 workgroups spread over the matrix with no two neighbours on one XCD: the operands come from HBM, not
// L2 -- PMC: 12.9 GB per factorisation, 24 x the matrix.  All the same the loop does not wait for memory: a ring of four 16-column slices
// with the DMA three slices ahead measured the same 77-80 us per K = 512 tile as two buffers one slice ahead
// (profiles/r04_o_fit_mega_ring_only.log), slices of 32 instead of 16 -- half the barriers -- 76.2 against 77.8 us
// (profiles/r04_t_mega_trace_kslice32.log): the tile runs at ~0.85 of the matrix pipe's rate at the clock the chip holds, the
// launch-per-step kernel gets its last 10 % from the second workgroup on the CU.)
// quarter q + 4: the quarter's sum of products alone, written (all 64 x 64 elements, nothing read) to Cp with leading dimension ldc --
// a STARTED sum that mega_potrf_link continues (the first block of an outer panel, below).
__device__ __noinline__ void mega_gemm_tile(const double *__restrict__ Ap, const double *__restrict__ Bp, double *__restrict__ Cp,
                                            int ld, int ldc, int K, bool diag, int quarter) {
    // (each task body is a function of its own: inlined into one kernel the three of them need more than the 256 registers a wave of
    // a 512-thread workgroup can have, and the diagonal block's serial chain is the last place for scratch traffic; the dynamic LDS is
    // named here again rather than passed, so that the compiler keeps LDS addressing)
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x;
    // (arguments of a function that is not inlined arrive in vector registers and as generic pointers: say that they are uniform and
    // that C is global memory -- flat stores would also count on the LDS counter the k loop waits on)
    ld = __builtin_amdgcn_readfirstlane(ld);
    ldc = __builtin_amdgcn_readfirstlane(ldc);
    K = __builtin_amdgcn_readfirstlane(K);
    quarter = __builtin_amdgcn_readfirstlane(quarter);
    const bool raw = quarter >= 4;
    if (raw) quarter -= 4;
    diag = __builtin_amdgcn_readfirstlane((int)diag) != 0;
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble *const Cg = (gdouble *)Cp;
    constexpr int TKm = 32, STR = 144;      // 2 buffers x 2 operands x 32 x 144 doubles = 144 KB
    double *As = sm, *Bs = sm + 2 * TKm * STR;
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const double *Asrc = Ap + lane * 2 + (size_t)wave * ld, *Bsrc = Bp + lane * 2 + (size_t)wave * ld;
    auto stage = [&](int kt) {      // k-slice kt into buffer kt & 1: eight LDS-DMA instructions per wave
        const size_t koff = (size_t)kt * TKm;
        const int buf = kt & 1;
#pragma unroll
        for (int q = 0; q < TKm / 8; ++q) {
            __builtin_amdgcn_global_load_lds(Asrc + (koff + 8 * q) * ld, As + (buf * TKm + wave + 8 * q) * STR, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(Bsrc + (koff + 8 * q) * ld, Bs + (buf * TKm + wave + 8 * q) * STR, 16, 0, 0);
        }
    };
    const int KT = K / TKm;
    // slice kt is in LDS for everybody: this wave's own share has landed and its reads of the slice before are done, then the barrier;
    // the other buffer is free after it and takes slice kt + 1 while the matrix cores run slice kt
    auto arrive = [&](int kt) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < KT) stage(kt + 1);
    };
    stage(0);
    if (quarter >= 0) {
        // A quarter is on the chain's way (the next diagonal block's tile): all eight waves share its 64 x 64 -- wave w: rows 16 (w & 3),
        // columns 32 (w >> 2), two accumulators -- so its k loop is a quarter of a full tile's instead of the same length on two waves.
        // Same products in the same k order per element as the full tile.
        const int wr = 64 * (quarter & 1) + 16 * (wave & 3), wc = 64 * (quarter >> 1) + 32 * (wave >> 2);
        double4_t acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        for (int kt = 0; kt < KT; ++kt) {
            arrive(kt);
            const int cur = kt & 1;
            const double *Ac = As + cur * TKm * STR + wr + fr, *Bc = Bs + cur * TKm * STR + wc + fr;
            // fragments one k-group ahead of the matrix cores (two waves per SIMD, in step with each other: nothing else hides an LDS read)
            double af = Ac[fk * STR], bf0 = Bc[fk * STR], bf1 = Bc[fk * STR + 16];
#pragma unroll
            for (int ks = 0; ks < TKm / 4; ++ks) {
                const double a0 = af, b0 = bf0, b1 = bf1;
                if (ks + 1 < TKm / 4) {
                    af = Ac[((ks + 1) * 4 + fk) * STR];
                    bf0 = Bc[((ks + 1) * 4 + fk) * STR];
                    bf1 = Bc[((ks + 1) * 4 + fk) * STR + 16];
                }
                acc[0] = MFMA(b0, a0, acc[0]);
                acc[1] = MFMA(b1, a0, acc[1]);
            }
        }
        const int m = wr + fr;
        if (raw) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cg[m + (size_t)(wc + nt * 16 + fk + 4 * r) * ldc] = acc[nt][r];
            return;
        }
        double cv[2][4];
        if (!diag) {
            // off the diagonal nothing is masked: eight loads in flight together, eight stores nobody waits for.  (With the mask
            // in the way -- one exec-masked block per element -- the compiler cannot count what is outstanding and puts
            // `s_waitcnt vmcnt(0)` in front of every store: each store then waits for the acknowledgement of the one before it.)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) cv[nt][r] = Cg[m + (size_t)(wc + nt * 16 + fk + 4 * r) * ldc];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cg[m + (size_t)(wc + nt * 16 + fk + 4 * r) * ldc] = fma(1.0, cv[nt][r], -1.0 * acc[nt][r]);
            return;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = wc + nt * 16 + fk + 4 * r;
                cv[nt][r] = (m < n) ? 0.0 : Cg[m + (size_t)n * ldc];
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = wc + nt * 16 + fk + 4 * r;
                const double v = fma(1.0, cv[nt][r], -1.0 * acc[nt][r]);
                if (!(m < n)) Cg[m + (size_t)n * ldc] = v;
            }
        return;
    }
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
    double4_t acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int kt = 0; kt < KT; ++kt) {
        arrive(kt);
        const int cur = kt & 1;
        const double *Ac = As + cur * TKm * STR + wm + fr, *Bc = Bs + cur * TKm * STR + wn + fr;
        double afn[4], bfn[2];
#pragma unroll
        for (int u = 0; u < 4; ++u) afn[u] = Ac[fk * STR + u * 16];
#pragma unroll
        for (int u = 0; u < 2; ++u) bfn[u] = Bc[fk * STR + u * 16];
#pragma unroll
        for (int ks = 0; ks < TKm / 4; ++ks) {
            double af[4], bf[2];
#pragma unroll
            for (int u = 0; u < 4; ++u) af[u] = afn[u];
#pragma unroll
            for (int u = 0; u < 2; ++u) bf[u] = bfn[u];
            if (ks + 1 < TKm / 4) {      // the next k-group's fragments are on their way while these sixteen products issue
#pragma unroll
                for (int u = 0; u < 4; ++u) afn[u] = Ac[((ks + 1) * 4 + fk) * STR + u * 16];
#pragma unroll
                for (int u = 0; u < 2; ++u) bfn[u] = Bc[((ks + 1) * 4 + fk) * STR + u * 16];
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = MFMA(bf[nt], af[mt], acc[nt][mt]);
        }
    }
    // acc[nt][mt][r] = sum over k for (m = wm + 16 mt + fr, n = wn + 16 nt + fk + 4 r): C <- 1.0 * C + (-1.0) * acc, as gemm_nt_f64_kernel
    if (!diag) {
        // unmasked (see the quarter's epilogue): per column half sixteen loads together, then sixteen stores
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            double cv[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) cv[r][mt] = Cg[(wm + mt * 16 + fr) + (size_t)(wn + nt * 16 + fk + 4 * r) * ldc];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    Cg[(wm + mt * 16 + fr) + (size_t)(wn + nt * 16 + fk + 4 * r) * ldc] = fma(1.0, cv[r][mt], -1.0 * acc[nt][mt][r]);
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        double cv[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = wn + nt * 16 + fk + 4 * r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = wm + mt * 16 + fr;
                cv[r][mt] = (m < n) ? 0.0 : Cg[m + (size_t)n * ldc];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = wn + nt * 16 + fk + 4 * r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = wm + mt * 16 + fr;
                const double v = fma(1.0, cv[r][mt], -1.0 * acc[nt][mt][r]);
                if (!(m < n)) Cg[m + (size_t)n * ldc] = v;
            }
        }
    }
}

// X (128 rows x 128 columns at Xp, leading dimension ld) <- X L^-T, L at Lp (same ld), its tile inverses at dv: eight waves x 16 rows,
// the strip in LDS with column stride LKS -- per row the arithmetic of trsm_panel128_kernel (trsm_chunk)
__device__ __noinline__ void mega_trsm_rows(double *__restrict__ Xp, const double *__restrict__ Lp, const double *__restrict__ dv, int ld) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const int li = (tid & 63) * 2, lc = tid >> 6;
    {
        const double *src = Xp + lane * 2 + (size_t)wave * ld;
#pragma unroll
        for (int q = 0; q < NB / 8; ++q) __builtin_amdgcn_global_load_lds(src + (size_t)(8 * q) * ld, xs + (wave + 8 * q) * LKS, 16, 0, 0);
    }
    double fa[28], fb[28];
    trsm_load_frags<1>(Lp, ld, fr, fg, fa);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const int sp = wave * 16 + fr;
    double ss = 0.0;
    trsm_load_frags<2>(Lp, ld, fr, fg, fb);
    trsm_chunk<0, LKS>(xs, sp, fr, fg, fa, dv, ss);
    trsm_chunk<1, LKS>(xs, sp, fr, fg, fa, dv, ss);
    trsm_load_frags<3>(Lp, ld, fr, fg, fa);
    trsm_chunk<2, LKS>(xs, sp, fr, fg, fb, dv, ss);
    trsm_load_frags<4>(Lp, ld, fr, fg, fb);
    trsm_chunk<3, LKS>(xs, sp, fr, fg, fa, dv, ss);
    trsm_load_frags<5>(Lp, ld, fr, fg, fa);
    trsm_chunk<4, LKS>(xs, sp, fr, fg, fb, dv, ss);
    trsm_load_frags<6>(Lp, ld, fr, fg, fb);
    trsm_chunk<5, LKS>(xs, sp, fr, fg, fa, dv, ss);
    trsm_load_frags<7>(Lp, ld, fr, fg, fa);
    trsm_chunk<6, LKS>(xs, sp, fr, fg, fb, dv, ss);
    trsm_chunk<7, LKS>(xs, sp, fr, fg, fa, dv, ss);
    (void)ss;
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < 16; ++q)
        *reinterpret_cast<double2_t *>(Xp + li + (size_t)(lc + 8 * q) * ld) = *reinterpret_cast<const double2_t *>(xs + (lc + 8 * q) * LKS + li);
}

// the diagonal block, exactly as potrf_diag128_kernel does it (masked LDS-DMA load of the lower part, potrf_block_body)
__device__ __noinline__ void mega_potrf_block(double *__restrict__ Akk, int lda, double *__restrict__ dinvk, int *info, int base) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    double *a = sm, *dv = sm + NB * PLS;
    int *flag = reinterpret_cast<int *>(dv + 256);
    if (tid == 0) *flag = 0;
    const double *src = Akk + lane * 2;
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (lane * 2 + 1 >= j) __builtin_amdgcn_global_load_lds(src + (size_t)j * lda, a + j * PLS, 16, 0, 0);
    } else {
        for (int j = 16 + (wave - 1); j < NB; j += POTRF_WAVES - 1)
            if (lane * 2 + 1 >= j) __builtin_amdgcn_global_load_lds(src + (size_t)j * lda, a + j * PLS, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    potrf_block_body(a, dv, flag, Akk, lda, dinvk, info, base, tid, lane, wave, fr, fg);
}

// Diagonal block k INSIDE an outer panel (k > the panel's first block), with its link to the step before as the prologue: this block's
// 128 rows of block column k-1 are solved here, X = A[blk k, blk k-1] L_{k-1,k-1}^-T (what TRSM(k-1, k) would do: the strip solve of
// mega_trsm_rows), stored and ANNOUNCED (done[aux] = epoch: the tasks that multiply by these rows start now), the block's last in-panel
// update D = A[blk k, blk k] - X X^T is formed from the strip still in LDS (36 lower 16 x 16 tiles over the eight waves, K = 128,
// accumulated from zero and subtracted at the end -- element for element the arithmetic of the GEMM tile it replaces) and dropped into
// the LDS image the factorisation starts from.  On the chain this is one task and ~20 us where panel solve, hand-over, quarter
// update and hand-over were ~45.  (Round 3's potrf_link128_kernel did the same as a launch; there the rest of the panel had to wait
// on a helper stream, here it is just more tasks.)
// The FIRST block of an outer panel is linked the same way (round 4, last change): there the update this task completes is the previous
// panel's OUTER update of the tile, K = 512 -- its first 384 columns are summed ahead of time by three type-5 quarter tasks into a
// 128 x 128 scratch tile (acc0; they only need rows that were solved a step earlier), this task starts from that sum instead of from
// zero and adds the last block column's 128 products: the same sequence of matrix-core accumulations per element as the K = 512
// quarter tiles it replaces (k ascending from the panel's first column, C - sum at the end), so the factor does not change by a bit,
// and the chain across a panel boundary is this one task instead of panel solve -> hand-over -> K = 512 quarter -> hand-over -> block.
__device__ __noinline__ void mega_potrf_link(double *__restrict__ A, int lda, double *__restrict__ dinv, int *info, int base, int *__restrict__ done_aux, int epoch,
                                             const double *__restrict__ acc0) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *a = sm;                 // NB x PLS: first the strip X, then the block D
    double *dv = sm + NB * PLS;
    int *flag = reinterpret_cast<int *>(dv + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    if (tid == 0) *flag = 0;
    double *X = A - (size_t)NB * lda;                       // rows of this block, columns of the block before
    const double *Lp = A - NB - (size_t)NB * lda;           // the diagonal block before (factored) ...
    const double *dp = dinv - 8 * 256;                      // ... and its tile inverses
    {
        const double *src = X + lane * 2 + (size_t)wave * lda;
#pragma unroll
        for (int q = 0; q < NB / 8; ++q) __builtin_amdgcn_global_load_lds(src + (size_t)(8 * q) * lda, a + (wave + 8 * q) * PLS, 16, 0, 0);
    }
    // this wave's tiles of D (q = wave, wave + 8, ...; at most 5): fetched now, under the solve
    double4_t dold[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = wave + 8 * u;
        int I = 0, J = 0;
        if (q < 36) tri_coords(q, I, J);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) dold[u][rr] = (q < 36) ? A[(16 * I + fr) + (size_t)(16 * J + fg + 4 * rr) * lda] : 0.0;
    }
    double fa[28], fb[28];
    trsm_load_frags<1>(Lp, lda, fr, fg, fa);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    const int sp = wave * 16 + fr;
    double ss = 0.0;
    trsm_load_frags<2>(Lp, lda, fr, fg, fb);
    trsm_chunk<0, PLS>(a, sp, fr, fg, fa, dp, ss);
    trsm_chunk<1, PLS>(a, sp, fr, fg, fa, dp, ss);
    trsm_load_frags<3>(Lp, lda, fr, fg, fa);
    trsm_chunk<2, PLS>(a, sp, fr, fg, fb, dp, ss);
    trsm_load_frags<4>(Lp, lda, fr, fg, fb);
    trsm_chunk<3, PLS>(a, sp, fr, fg, fa, dp, ss);
    trsm_load_frags<5>(Lp, lda, fr, fg, fa);
    trsm_chunk<4, PLS>(a, sp, fr, fg, fb, dp, ss);
    trsm_load_frags<6>(Lp, lda, fr, fg, fb);
    trsm_chunk<5, PLS>(a, sp, fr, fg, fa, dp, ss);
    trsm_load_frags<7>(Lp, lda, fr, fg, fa);
    trsm_chunk<6, PLS>(a, sp, fr, fg, fb, dp, ss);
    trsm_chunk<7, PLS>(a, sp, fr, fg, fa, dp, ss);
    (void)ss;
    __syncthreads();
    // the started sums of this wave's tiles (the L fragments' registers are free now), or zeros
    double4_t acc_init[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = wave + 8 * u;
        int I = 0, J = 0;
        if (q < 36) tri_coords(q, I, J);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc_init[u][rr] = (acc0 && q < 36) ? acc0[(16 * I + fr) + (16 * J + fg + 4 * rr) * NB] : 0.0;
    }
    {   // the solved rows go back in place now: the stores drain while the matrix cores run the tile update
        const int li = lane * 2, lc = wave;
#pragma unroll 8
        for (int q = 0; q < NB / 8; ++q)
            *reinterpret_cast<double2_t *>(X + li + (size_t)(lc + 8 * q) * lda) = *reinterpret_cast<const double2_t *>(a + (lc + 8 * q) * PLS + li);
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = wave + 8 * u;
        if (q < 36) {
            int I, J;
            tri_coords(q, I, J);
            double4_t acc = acc_init[u];
#pragma unroll 8
            for (int ks = 0; ks < 32; ++ks) {
                const int k = 4 * ks + fg;
                acc = MFMA(a[k * PLS + 16 * J + fr], a[k * PLS + 16 * I + fr], acc);     // acc[rr]: (m = 16 I + fr, n = 16 J + fg + 4 rr)
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) dold[u][rr] = fma(1.0, dold[u][rr], -1.0 * acc[rr]);      // from here on: the updated tile
        }
        if (u == 0) {
            // every wave's stores of X have had a tile's time to drain: announce the solved rows now, not after the whole update
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 64) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(done_aux, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    __syncthreads();                      // nobody reads the strip any more
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int q = wave + 8 * u;
        if (q >= 36) break;
        int I, J;
        tri_coords(q, I, J);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a[(16 * I + fr) + (16 * J + fg + 4 * rr) * PLS] = dold[u][rr];
    }
    // (the first barrier of the factorisation orders these writes before anybody reads them)
    potrf_block_body(a, dv, flag, A, lda, dinv, info, base, tid, lane, wave, fr, fg);
}

__global__ __launch_bounds__(MEGA_THREADS, 1) void chol_mega_kernel(double *__restrict__ A, int lda, double *__restrict__ dinv, int *info,
                                                                 const chol_task *__restrict__ tasks, int ntasks, int *__restrict__ done,
                                                                 int *__restrict__ counter, int epoch, int *__restrict__ err, int mode,
                                                                 unsigned long long *__restrict__ stamps, double *__restrict__ sums) {
    __shared__ int s_task;
    const int tid = threadIdx.x;
    // Control flow is kept UNIFORM: the loop runs on a scalar task index, and every lane-0 block sits between two barriers of its own.
    // (A first version claimed at the top of the loop and published at the bottom, both under `if (tid == 0)` with no barrier between
    // them across the back edge: hipcc merged the two blocks into one divergent region and structurised the loop so that lane 0 left
    // it alone -- the other 511 threads went round again on the stale task index, for ever.)
    if (tid == 0) s_task = atomicAdd(counter, 1);
    __syncthreads();
    int t = __builtin_amdgcn_readfirstlane(s_task);
    __syncthreads();
    while (t < ntasks) {
        const chol_task tk = tasks[t];
        if (tk.type == 3) {
            // a place in the list for a result that a fused task announces half-way (mega_potrf_link): nothing to wait for, run or publish
            if (tid == 0) s_task = atomicAdd(counter, 1);
            __syncthreads();
            t = __builtin_amdgcn_readfirstlane(s_task);
            __syncthreads();
            continue;
        }
        if (stamps && tid == 0) stamps[4 * (size_t)t] = __builtin_amdgcn_s_memrealtime();      // lab (GPCORE_MEGA_TRACE): claimed
        if (tid == 0) {
            for (int d = 0; d < 10; ++d) {
                const int dep = tk.dep[d];
                if (dep < 0) continue;
                int it = 0;
                while (__hip_atomic_load(done + dep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (++it > 400000) { atomicExch(err, 1 + t); break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (stamps && tid == 0) stamps[4 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();  // dependencies met, acquire done
        if (tk.type == 0) {
            if (mode & 1)
                mega_potrf_block(A + (size_t)tk.k0 * NB * ((size_t)lda + 1), lda, dinv + (size_t)tk.k0 * NB * 16, info, tk.k0 * NB);
        } else if (tk.type == 1) {
            if (mode & 2)
                mega_trsm_rows(A + (size_t)tk.i * NB + (size_t)tk.k0 * NB * lda, A + (size_t)tk.k0 * NB * ((size_t)lda + 1),
                               dinv + (size_t)tk.k0 * NB * 16, lda);
        } else if (tk.type == 4) {
            if (mode & 1)
                mega_potrf_link(A + (size_t)tk.k0 * NB * ((size_t)lda + 1), lda, dinv + (size_t)tk.k0 * NB * 16, info, tk.k0 * NB, done + tk.q, epoch,
                                tk.j >= 0 ? sums + (size_t)tk.j * NB * NB : nullptr);
        } else if (tk.type == 5) {
            if (mode & 4)
                mega_gemm_tile(A + (size_t)tk.i * NB + (size_t)tk.k0 * NB * lda, A + (size_t)tk.i * NB + (size_t)tk.k0 * NB * lda,
                               sums + (size_t)tk.j * NB * NB, lda, NB, tk.kb * NB, false, tk.q + 4);
        } else if (mode & 4)
            mega_gemm_tile(A + (size_t)tk.i * NB + (size_t)tk.k0 * NB * lda, A + (size_t)tk.j * NB + (size_t)tk.k0 * NB * lda,
                           A + (size_t)tk.i * NB + (size_t)tk.j * NB * lda, lda, lda, tk.kb * NB, tk.i == tk.j, tk.q);
        // publish: every wave's stores acknowledged, then ONE lane releases at agent scope, sets the flag -- and claims the next task
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (stamps) stamps[4 * (size_t)t + 2] = __builtin_amdgcn_s_memrealtime();                 // body done, stores acknowledged
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(done + t, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stamps) stamps[4 * (size_t)t + 3] = __builtin_amdgcn_s_memrealtime() | ((unsigned long long)blockIdx.x << 48);   // published; who ran it
            s_task = atomicAdd(counter, 1);
        }
        __syncthreads();
        t = __builtin_amdgcn_readfirstlane(s_task);
        __syncthreads();
    }
}

}  // namespace

static constexpr int POTRF_LDS = (NB * PLS + 256 + 8) * (int)sizeof(double);
static constexpr int TRSM_LDS = NB * XS * (int)sizeof(double);
static constexpr int STEP_LDS = (NB * LS1 + 8 * 256 + NB + 16) * (int)sizeof(double);
static constexpr int LINK_LDS = NB * LKS * (int)sizeof(double);
static constexpr int MEGA_LDS = POTRF_LDS > LINK_LDS ? POTRF_LDS : LINK_LDS;     // the diagonal block's image / the 128-row strip / 144 KB of GEMM staging (two buffers of 32 k-columns)

int gpk_init_diag_kernels() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_diag128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, POTRF_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(trsm_panel128_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(trsm_panel128_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, TRSM_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(ep_link_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LINK_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(chol_mega_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MEGA_LDS);
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
// tasks: ntasks x 16 ints (chol_task), done: ntasks ints, counter / err: one int each (counter zeroed by the caller in stream order)
void gpk_chol_mega(hipStream_t s, int num_cu, double *A, int lda, double *dinv, int *d_info, const int *tasks, int ntasks, int *done, int *counter,
                   int epoch, int *err, double *sums, unsigned long long *stamps) {
    const int grid = ntasks < num_cu ? ntasks : num_cu;
    static const int mode = [] { const char *e = getenv("GPCORE_MEGA_MODE"); return e ? atoi(e) : 7; }();   // lab: which task bodies run
    hipLaunchKernelGGL(chol_mega_kernel, dim3(grid), dim3(MEGA_THREADS), MEGA_LDS, s, A, lda, dinv, d_info, reinterpret_cast<const chol_task *>(tasks),
                       ntasks, done, counter, epoch, err, mode, stamps, sums);
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
