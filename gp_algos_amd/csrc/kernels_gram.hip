// ARD-RBF Gram-matrix kernels (HBM-write-bound; one exp per entry).
//   k(x, y, same) = sf*sf*exp(-0.5 * sum_k ((x_k-y_k) * 1/(l_k*l_k)) * (x_k-y_k)) + (same ? sn*sn : 0)
// restating GaussianRbfKernel.apply (utils/KernelRequisites.scala:66-72,109-113) as driven by
// MatrixUtils.buildKernelMatrix (utils/MatrixUtils.scala:44-70).
//
// Layout: X is column-major n x d ("feature-major": d runs of n doubles), so lane i loading
// X[i + k*ldx] is a fully coalesced 512-B wave access.  One 256-thread workgroup produces a 64x64
// tile: each thread owns one row point (its d features live in registers) and 16 column points
// (read as LDS broadcasts).  The symmetric builder evaluates tiles on/below the diagonal only, like
// the reference's j <= i loop, and mirrors through an LDS transpose so both stores are coalesced.
#include "gpcore_internal.h"
#include <atomic>

namespace {

constexpr int GT = 64;     // tile edge
constexpr int DC = 8;      // feature chunk held in registers
constexpr int GP_DMAX = 64;

struct GramParams {
    double sf2, sn2, extra;
    double inv_ls2[GP_DMAX];
    double inv_ls[GP_DMAX];    // 1 / l_k (the matrix-core form scales the features once)
    // derivative mode (GaussianRbfKernel.derAfterHyperParam, KernelRequisites.scala:76-86): 0 = the kernel itself,
    // 1 = d/d sf (2 sf e), 2 = d/d l_k (sf^2 e (x_k - y_k)^2 l_k^-3, k = dk), 3 = d/d sn (same ? 2 sn : 0)
    int dmode, dk;
    double dcoef;   // 2 sf | sf^2 l_k^-3 | 2 sn
    double lnsf2;   // ln(sf^2): the unit kernel carries it inside the exponent
};

__device__ __forceinline__ void tile_lower(int t, int &bi, int &bj) {
    int b = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (b * (b + 1) / 2 > t) --b;
    while ((b + 1) * (b + 2) / 2 <= t) ++b;
    bi = b;
    bj = t - b * (b + 1) / 2;
}

template <bool SYM, bool DER = false>
__global__ __launch_bounds__(256) void gram_rbf_kernel(const double *__restrict__ Xr, int nr, int ldxr,
                                                       const double *__restrict__ Xc, int nc, int ldxc, int d,
                                                       GramParams prm, double *__restrict__ K, int ldk, int full,
                                                       int nbr, const int *__restrict__ far_flag = nullptr, int epoch = 0) {
    // launched behind the unit kernel as its other half: works only when the scan found a point beyond GRAM_NORM_LIMIT (see there)
    if (far_flag && *far_flag != epoch) return;
    __shared__ double xcs[DC][GT];
    __shared__ double tile[SYM ? GT * (GT + 1) : 1];
    int bi, bj;
    if (SYM) tile_lower(blockIdx.x, bi, bj);
    else { bi = blockIdx.x % nbr; bj = blockIdx.x / nbr; }
    const int tid = threadIdx.x, ti = tid & 63, tq = tid >> 6;
    const int gi = bi * GT + ti;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;

    for (int kc = 0; kc < d; kc += DC) {
        if (kc) __syncthreads();
#pragma unroll
        for (int e = tid; e < DC * GT; e += 256) {
            int kk = e >> 6, jj = e & 63, gj = bj * GT + jj;
            xcs[kk][jj] = (gj < nc && kc + kk < d) ? Xc[gj + (size_t)(kc + kk) * ldxc] : 0.0;
        }
        double xi[DC], inv[DC];
#pragma unroll
        for (int kk = 0; kk < DC; ++kk) {
            bool ok = (gi < nr) && (kc + kk < d);
            xi[kk] = ok ? Xr[gi + (size_t)(kc + kk) * ldxr] : 0.0;
            inv[kk] = (kc + kk < d) ? prm.inv_ls2[kc + kk] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int jj = tq + 4 * q;
#pragma unroll
            for (int kk = 0; kk < DC; ++kk) {
                double diff = xi[kk] - xcs[kk][jj];
                acc[q] = fma(diff * inv[kk], diff, acc[q]);
            }
        }
    }

    const bool diag_tile = SYM && (bi == bj);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int jj = tq + 4 * q, gj = bj * GT + jj;
        double v = prm.sf2 * exp(-0.5 * acc[q]);
        if (SYM && gi == gj) v = (prm.sf2 + prm.sn2) + prm.extra;  // exp(-0) == 1: sf*sf*1 + sn*sn (+ sigmaNoise)
        if (DER) {   // compile-time: the plain Gram kernels carry none of this
            const double e = exp(-0.5 * acc[q]);
            if (prm.dmode == 1) v = prm.dcoef * e;
            else if (prm.dmode == 2) {
                const double diff = (gi < nr && gj < nc) ? Xr[gi + (size_t)prm.dk * ldxr] - Xc[gj + (size_t)prm.dk * ldxc] : 0.0;
                v = (prm.dcoef * e) * (diff * diff);
            } else v = (SYM && gi == gj) ? prm.dcoef : 0.0;
        }
        if (gi < nr && gj < nc && !(diag_tile && !full && gi < gj)) K[gi + (size_t)gj * ldk] = v;
        if (SYM) tile[ti * (GT + 1) + jj] = v;
    }
    if (SYM && full && !diag_tile) {
        __syncthreads();
        // mirrored tile: K(bj*GT + a, bi*GT + b) = tile[b][a]; lanes run over a (contiguous rows)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int b = tq + 4 * q, a = ti;
            const int r = bj * GT + a, c = bi * GT + b;
            if (r < nc && c < nr) K[r + (size_t)c * ldk] = tile[b * (GT + 1) + a];
        }
    }
}

// The same Gram matrices with the X X^T term of the squared distance on the matrix cores (north star: "MFMA on the X.X^T term
// of the squared-distance kernel"):  with z = (x - c) / l  (c = the first training point, a common shift that leaves every
// difference unchanged and keeps the norms small),
//     r^2_ij = |z_i|^2 + |z_j|^2 - 2 z_i . z_j,        K_ij = sf^2 exp(-r^2_ij / 2) (+ sn^2 on the diagonal, written exactly).
// The per-pair form above spends 3 d + ~30 fp64 VALU instructions per entry (d = 8: ~55) and is VALU-bound at 3.7 TB/s of
// stores; here the d-dependent part is 2 d / 4 MFMA k-steps in the matrix pipe, which runs beside the VALU, and what is left
// per entry (norm adds, clamp, exp, scale) fits under the HBM store time.  Rounding: the error of r^2 is ABSOLUTE, a few ulp
// of |z|^2, i.e. a RELATIVE error of the same size in K = exp(-r^2/2): ~1e-15 for the configurations here, inside the 1e-13
// the parity tests state (GPCORE_GRAM_MFMA=0 selects the per-pair form, which follows the reference's own operation order).
// One 256-thread workgroup per 64 x 64 tile: scaled features staged in LDS in chunks of 16 ([k][row], row stride 80 doubles
// -> conflict-free fragment reads), wave w owns the tile's 16-column block w and its 4 row blocks.  MFMA operand roles put the
// ROW index on lane & 15, so every store instruction writes 4 full 128-byte lines of column-major K.
typedef double gram_d4 __attribute__((ext_vector_type(4)));
constexpr int ZS = 80;      // LDS row stride (doubles): 640 B = 128 (mod 256)
constexpr int ZC = 16;      // features per staged chunk

template <bool SYM>
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double *__restrict__ Xr, int nr, int ldxr, const double *__restrict__ Xc,
                                                        int nc, int ldxc, int d, GramParams prm, const double *__restrict__ center, int ldcen,
                                                        double *__restrict__ K, int ldk, int full, int nbr) {
    __shared__ __attribute__((aligned(16))) double zr[ZC * ZS], zc[ZC * ZS];
    __shared__ double nrm[2][GT];
    __shared__ double tile[SYM ? GT * (GT + 1) : 1];
    int bi, bj;
    if (SYM) tile_lower(blockIdx.x, bi, bj);
    else { bi = blockIdx.x % nbr; bj = blockIdx.x / nbr; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const int i0 = bi * GT, j0 = bj * GT;
    gram_d4 acc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[it] = (gram_d4){0.0, 0.0, 0.0, 0.0};
    double nacc = 0.0;   // threads 0-63: |z|^2 of row i0 + tid; threads 64-127: of column j0 + tid - 64
    // rows / columns past the end are clamped to the last point: their products are computed and never stored
    const double *xr = Xr + min(i0 + lane, nr - 1), *xc = Xc + min(j0 + lane, nc - 1);
    for (int kc = 0; kc < d; kc += ZC) {
        if (kc) __syncthreads();
        // stage the chunk: thread -> (point = lane, feature = wave + 4 q)
#pragma unroll
        for (int q = 0; q < ZC / 4; ++q) {
            const int kk = wave + 4 * q, k = kc + kk;
            double vr = 0.0, vc = 0.0;
            if (k < d) {
                const double cen = center[(size_t)k * ldcen], il = prm.inv_ls[k];
                vr = (xr[(size_t)k * ldxr] - cen) * il;
                vc = (xc[(size_t)k * ldxc] - cen) * il;
            }
            zr[kk * ZS + lane] = vr;
            zc[kk * ZS + lane] = vc;
        }
        __syncthreads();
        if (tid < 2 * GT) {
            const double *z = (tid < GT) ? zr : zc;
#pragma unroll
            for (int kk = 0; kk < ZC; ++kk) nacc = fma(z[kk * ZS + lane], z[kk * ZS + lane], nacc);
        }
        const int ksteps = (min(d - kc, ZC) + 3) / 4;
        for (int ks = 0; ks < ksteps; ++ks) {
            const double a = zc[(4 * ks + fk) * ZS + 16 * wave + fr];       // column point j0 + 16 wave + fr, feature 4 ks + fk
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const double b = zr[(4 * ks + fk) * ZS + 16 * it + fr];     // row point i0 + 16 it + fr
                acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[it], 0, 0, 0);   // reg r: (j = fk + 4 r, i = fr)
            }
        }
    }
    if (tid < 2 * GT) nrm[tid >> 6][lane] = nacc;
    __syncthreads();
    const bool diag_tile = SYM && (bi == bj);
    // interior off-diagonal tiles (nearly all of them) store without per-element tests
    const bool plain = !diag_tile && (i0 + GT <= nr) && (j0 + GT <= nc);
    double *Kp = K + (i0 + fr) + (size_t)(j0 + 16 * wave + fk) * ldk;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int il = 16 * it + fr, gi = i0 + il;
        const double ni = nrm[0][il];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jl = 16 * wave + fk + 4 * r, gj = j0 + jl;
            const double r2 = fmax(fma(-2.0, acc[it][r], ni + nrm[1][jl]), 0.0);
            double v = prm.sf2 * exp(-0.5 * r2);
            if (plain) {
                Kp[16 * it + (size_t)(4 * r) * ldk] = v;
            } else {
                if (SYM && gi == gj) v = (prm.sf2 + prm.sn2) + prm.extra;      // exp(-0) == 1: sf*sf*1 + sn*sn (+ sigmaNoise), exact
                if (gi < nr && gj < nc && !(diag_tile && !full && gi < gj)) Kp[16 * it + (size_t)(4 * r) * ldk] = v;
            }
            if (SYM) tile[il * (GT + 1) + jl] = v;
        }
    }
    if (SYM && full && !diag_tile) {
        __syncthreads();
        const int ti = tid & 63, tq = tid >> 6;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int b = tq + 4 * q, a = ti;
            const int r = j0 + a, c = i0 + b;
            if (r < nc && c < nr) K[r + (size_t)c * ldk] = tile[b * (GT + 1) + a];
        }
    }
}

// The form used for d <= 14: the WHOLE exponent on the matrix cores, one wave per job, no workgroup barrier, no LDS tile.
//     z~_i = (z_i, -|z_i|^2 / 2, 1),   z~_j = (z_j, 1, ln sf^2 - |z_j|^2 / 2)   ->   z~_i . z~_j = ln sf^2 - r_ij^2 / 2,   K_ij = exp(.)
// (z = (x - c) / l as above; d + 2 <= 16 features = KS <= 4 k-steps of v_mfma_f64_16x16x4_f64).  What the earlier forms lost was not
// arithmetic but the life cycle of a tile (tools/lab/gram_aug_lab.hip, profiles/r03_g_gram_lab.log): a tile's operand loads queue
// behind the CU's outstanding stores and come back after microseconds (cross-Gram 1.20 ms with them, 0.97 ms with operands made
// up from indices), and fp64 MFMA and fp64 VALU instructions do not overlap on this chip (exp alone 350 us + MFMA alone 275 us =
// 638 us together), so every instruction that is not exp or MFMA counts.  Hence:
//   * a UNIT is 64 rows x 16 columns of a row strip; a wave takes `upw` consecutive units of the strip-major order (lower forms:
//     strip b has 4 (b + 1) units), keeps the strip's four row operand blocks in registers for all of them, and asks for the NEXT
//     64-column tile's features before it issues the current tile's 64 stores -- the loads are back when the tile is done;
//   * operands go from global memory (X is L2-resident) into MFMA layout directly: lane (fr, fk) holds feature 4 ks + fk of point
//     fr; the norms come from two cross-lane adds; the accumulator IS the exponent (no norm adds, no scaling after exp);
//   * MFMA operand roles put the ROW index on lane & 15: each store instruction writes 4 full 128-byte lines.  When the mirrored
//     half is wanted the column points are fed in the order 4 (fr & 3) + (fr >> 2), which leaves each lane with 4 CONSECUTIVE
//     columns -> the mirror is one 32-byte store per lane and block (no transpose through LDS).
// Rounding: the error of the exponent is a few ulp of |z|^2 (absolute), i.e. the RELATIVE error of K grows with the spread of the data
// in length scales.  So every call is three launches: gram_far_kernel scans the points (|z|^2 against GRAM_NORM_LIMIT, 2-3 us), this
// kernel builds the matrix when none is far (relative error below 5e-14), and gram_rbf_kernel -- the per-pair sum in the reference's
// own order -- builds it otherwise; the half that is not needed returns at its first instruction.  No host round trip, and wide
// data costs what it cost in rounds 1-2.  (A per-tile choice inside this kernel was tried first: one wave per 64 x 64 tile on the
// vector unit runs at a third of gram_rbf_kernel's speed -- profiles/r03_h_gram_scales.log.)
// Measured at n = 8192, m = 65536, d = 8, as called (scan + kernel + empty launch): lower 84 -> 61 us (4.4 TB/s; 57 us the kernel alone),
// mirrored 124 -> 123 us, cross 1.12 -> 0.84 ms (5.1 TB/s); by feature count profiles/r03_l_gram_by_d.log.
constexpr double GRAM_NORM_LIMIT = 64.0;
constexpr int GU_DMAX = 14;

// exp(t) for FINITE t in [-900, 709.7] and nothing else: argument reduction by n = rint(t log2 e) against ln 2 in two pieces, the
// degree-11 minimax polynomial the device library's exp uses (same coefficients, same order: bit-identical results in range), one
// ldexp.  No overflow / underflow / NaN selects (5 instructions of 24): the caller's arguments are ln sf^2 - r^2 / 2 with r^2 <= 256
// on this path and ln sf^2 finite (the launcher sends anything else to the per-pair kernel); ldexp itself flushes 2^-1075 and below.
__device__ __forceinline__ double exp_bounded(double t) {
    const double n = __builtin_rint(t * 0x1.71547652b82fep+0);
    double f = fma(n, -0x1.62e42fefa39efp-1, t);
    f = fma(n, -0x1.abc9e3b39803fp-56, f);
    double p = fma(f, 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22);
    p = fma(f, p, 0x1.71dee623fde64p-19);
    p = fma(f, p, 0x1.a01997c89e6b0p-16);
    p = fma(f, p, 0x1.a01a014761f6ep-13);
    p = fma(f, p, 0x1.6c16c1852b7b0p-10);
    p = fma(f, p, 0x1.1111111122322p-7);
    p = fma(f, p, 0x1.55555555502a1p-5);
    p = fma(f, p, 0x1.5555555555511p-3);
    p = fma(f, p, 0x1.000000000000bp-1);
    p = fma(f, p, 1.0);
    p = fma(f, p, 1.0);
    return ldexp(p, (int)n);
}
// min without the canonicalising v_max the compiler puts in front of fmin (b is uniform)
__device__ __forceinline__ double min_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

template <int KS, int KR, bool ROW>
__device__ __forceinline__ void unit_operand(const double (&raw)[KR], int d, const double (&cen)[KR], const double (&il)[KR], int fk, double lnsf2,
                                              double (&z)[KS]) {
    double part = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) z[ks] = 0.0;
#pragma unroll
    for (int ks = 0; ks < KR; ++ks) {
        const double v = (4 * ks + fk < d) ? (raw[ks] - cen[ks]) * il[ks] : 0.0;
        z[ks] = v;
        part = fma(v, v, part);
    }
    part += __shfl_xor(part, 16);
    part += __shfl_xor(part, 32);
    const double hn = ROW ? -0.5 * part : fma(-0.5, part, lnsf2);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int k = 4 * ks + fk;
        if (k == d) z[ks] = ROW ? hn : 1.0;
        if (k == d + 1) z[ks] = ROW ? 1.0 : hn;
    }
}

// The scan in front of the unit kernel: one thread per point, |z|^2 = sum_k ((x_k - c_k) / l_k)^2 against GRAM_NORM_LIMIT (NaN counts
// as far).  A far point writes this call's epoch into the flag; nobody ever has to clear it (epochs are unique per process).
__global__ __launch_bounds__(256) void gram_far_kernel(const double *__restrict__ Xa, int na, int ldxa, const double *__restrict__ Xb, int nb, int ldxb,
                                                       int d, GramParams prm, const double *__restrict__ cenp, int ldcen, int *__restrict__ far_flag,
                                                       int epoch) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= na + nb) return;
    const double *xp = i < na ? Xa + i : Xb + (i - na);
    const int ld = i < na ? ldxa : ldxb;
    double z2 = 0.0;
    for (int k = 0; k < d; ++k) {
        const double v = (xp[(size_t)k * ld] - cenp[(size_t)k * ldcen]) * prm.inv_ls[k];
        z2 = fma(v, v, z2);
    }
    if (!(z2 <= GRAM_NORM_LIMIT)) *far_flag = epoch;
}

// MODE 0: cross-Gram (all tiles); 1: symmetric, lower triangle only; 2: symmetric, mirrored
template <int KS, int KR, int MODE>
__global__ __launch_bounds__(64) void gram_unit_kernel(const double *__restrict__ Xr, int nr, int ldxr, const double *__restrict__ Xc, int nc, int ldxc,
                                                       int d, GramParams prm, const double *__restrict__ cenp, int ldcen, double *__restrict__ K,
                                                       int ldk, int nbc, int upw, long total, const int *__restrict__ far_flag, int epoch) {
    if (*far_flag == epoch) return;      // a point beyond GRAM_NORM_LIMIT: the per-pair kernel launched behind this one builds the matrix
    const int lane = threadIdx.x, fr = lane & 15, fk = lane >> 4;
    long u0 = (long)blockIdx.x * upw;
    const long u1 = u0 + upw < total ? u0 + upw : total;
    const int cperm = (MODE == 2) ? 4 * (fr & 3) + (fr >> 2) : fr;
    double cen[KR], il[KR];
#pragma unroll
    for (int ks = 0; ks < KR; ++ks) {
        const int k = min(4 * ks + fk, d - 1);
        cen[ks] = cenp[(size_t)k * ldcen];
        il[ks] = prm.inv_ls[k];
    }
    const double lnsf2 = prm.lnsf2, dval = (prm.sf2 + prm.sn2) + prm.extra;
    constexpr int JS = (MODE == 2) ? 1 : 4;      // column step between a lane's four accumulator registers
    // lane parts of the store addresses as 32-bit byte offsets (ldk < 2^25, launcher) on top of a uniform base per block
    const unsigned voff = ((unsigned)fr + (unsigned)((MODE == 2) ? 4 * fk : fk) * (unsigned)ldk) * 8u;    // primary store
    const unsigned moff = ((unsigned)(4 * fk) + (unsigned)fr * (unsigned)ldk) * 8u;                        // mirrored store
    double zr[4][KS], zc[4][KS], raw[4][KR];
    while (u0 < u1) {
        // the strip u0 lies in, and this wave's range [q0, q1) of its 16-column blocks
        int bi;
        long ub;
        if (MODE == 0) {
            bi = (int)(u0 / (4 * nbc));
            ub = (long)bi * 4 * nbc;
        } else {
            bi = (int)((sqrt(2.0 * (double)u0 + 1.0) - 1.0) * 0.5);
            while (2L * bi * (bi + 1) > u0) --bi;
            while (2L * (bi + 1) * (bi + 2) <= u0) ++bi;
            bi = __builtin_amdgcn_readfirstlane(bi);      // uniform by construction; the sqrt left it in a vector register
            ub = 2L * bi * (bi + 1);
        }
        const int slen = (MODE == 0) ? 4 * nbc : 4 * (bi + 1);
        const int q0 = (int)(u0 - ub), q1 = (int)((u1 - ub < slen) ? u1 - ub : slen);
        const int i0 = bi * 64;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double *xp = Xr + min(i0 + 16 * b + fr, nr - 1);
#pragma unroll
            for (int ks = 0; ks < KR; ++ks) raw[b][ks] = xp[(size_t)min(4 * ks + fk, d - 1) * ldxr];
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) unit_operand<KS, KR, true>(raw[b], d, cen, il, fk, 0.0, zr[b]);
        const int c0 = q0 >> 2, c1 = (q1 + 3) >> 2;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double *xp = Xc + min(c0 * 64 + 16 * b + cperm, nc - 1);
#pragma unroll
            for (int ks = 0; ks < KR; ++ks) raw[b][ks] = xp[(size_t)min(4 * ks + fk, d - 1) * ldxc];
        }
        for (int c = c0; c < c1; ++c) {
#pragma unroll
            for (int b = 0; b < 4; ++b) unit_operand<KS, KR, false>(raw[b], d, cen, il, fk, lnsf2, zc[b]);
            if (c + 1 < c1) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double *xp = Xc + min((c + 1) * 64 + 16 * b + cperm, nc - 1);
#pragma unroll
                    for (int ks = 0; ks < KR; ++ks) raw[b][ks] = xp[(size_t)min(4 * ks + fk, d - 1) * ldxc];
                }
            }
            const int j0 = c * 64;
            const bool diag = MODE && bi == c;
            const int cbs = max(q0 - 4 * c, 0), cbe = min(q1 - 4 * c, 4);
            const bool plain = !diag && i0 + 64 <= nr && j0 + 64 <= nc && (MODE != 2 || (ldk & 3) == 0);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                if (cb < cbs || cb >= cbe) continue;
                gram_d4 acc[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) acc[it] = (gram_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int it = 0; it < 4; ++it) acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(zc[cb][ks], zr[it][ks], acc[it], 0, 0, 0);
                // register r of lane (fr, fk): row i0 + 16 it + fr, column j0 + 16 cb + (MODE 2 ? 4 fk + r : fk + 4 r)
                char *cbase = reinterpret_cast<char *>(K + i0 + (size_t)(j0 + 16 * cb) * ldk);
                char *mbase = reinterpret_cast<char *>(K + (j0 + 16 * cb) + (size_t)i0 * ldk);
                if (plain) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        gram_d4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = exp_bounded(min_raw(acc[it][r], lnsf2));
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            *reinterpret_cast<double *>(cbase + ((size_t)(16 * it) + (size_t)(JS * r) * ldk) * 8 + voff) = v[r];
                        if (MODE == 2) *reinterpret_cast<gram_d4 *>(mbase + (size_t)(16 * it) * ldk * 8 + moff) = v;
                    }
                } else {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        if (MODE == 1 && diag && it < cb) continue;     // block above the diagonal
                        const int gi = i0 + 16 * it + fr;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int gj = j0 + 16 * cb + ((MODE == 2) ? 4 * fk : fk) + JS * r;
                            double v = exp(fmin(acc[it][r], lnsf2));
                            if (MODE && gi == gj) v = dval;       // exp(-0) == 1: sf*sf*1 + sn*sn (+ sigmaNoise), exact
                            if (gi < nr && gj < nc && !(diag && gi < gj)) K[gi + (size_t)gj * ldk] = v;
                            if (MODE == 2 && gi < nr && gj < nc && (!diag || gi > gj)) K[gj + (size_t)gi * ldk] = v;
                        }
                    }
                }
            }
        }
        u0 = ub + q1;
    }
}

__global__ void pad_identity_kernel(double *A, int n, int np, int lda) {
    // zero rows [n,np) x cols [0,np) and rows [0,n) x cols [n,np); ones on the pad diagonal
    const int pad = np - n;
    size_t total = (size_t)pad * np + (size_t)n * pad;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int i, j;
        if (e < (size_t)pad * np) { i = n + (int)(e % pad); j = (int)(e / pad); }
        else { size_t f = e - (size_t)pad * np; i = (int)(f % n); j = n + (int)(f / n); }
        A[i + (size_t)j * lda] = (i == j) ? 1.0 : 0.0;
    }
}

__global__ void zero_upper_kernel(double *A, int n, int lda) {
    // grid rows (blockIdx.y, strided: gridDim.y is capped at 65535) walk the columns j; threads sweep the rows i < j (coalesced)
    for (int j = blockIdx.y; j < n; j += gridDim.y)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < j; i += gridDim.x * blockDim.x) A[i + (size_t)j * lda] = 0.0;
}

__global__ void fill_kernel(double *p, size_t count, double v) {
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) p[e] = v;
}

__global__ void copy2d_kernel(double *dst, int ldd, const double *src, int lds, int rows, int cols) {
    size_t total = (size_t)rows * cols;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(e % rows), j = (int)(e / rows);
        dst[i + (size_t)j * ldd] = src[i + (size_t)j * lds];
    }
}

__global__ void copy_strided_kernel(double *dst, size_t ds, const double *src, size_t ss, int count) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < count) dst[(size_t)j * ds] = src[(size_t)j * ss];
}

GramParams make_params(const double *theta, int d, double extra) {
    GramParams p;
    p.sf2 = theta[0] * theta[0];
    p.sn2 = theta[d + 1] * theta[d + 1];
    p.extra = extra;
    p.dmode = 0, p.dk = 0, p.dcoef = 0.0;
    p.lnsf2 = std::log(p.sf2);
    for (int k = 0; k < GP_DMAX; ++k) p.inv_ls2[k] = p.inv_ls[k] = 0.0;
    for (int k = 0; k < d; ++k) {
        p.inv_ls2[k] = 1.0 / (theta[1 + k] * theta[1 + k]);
        p.inv_ls[k] = 1.0 / fabs(theta[1 + k]);
    }
    return p;
}

}  // namespace

namespace {

// Centre of the matrix-core forms' z = (x - c) / l: the CENTROID of the training points (round 4; rounds 1-3 took the first training
// point).  Any centre is exact in exact arithmetic -- it only decides how large |z|^2 gets, i.e. the rounding of the exponent and
// whether the scan sends the call to the per-pair kernel -- and the centroid halves the norms of data that is spread evenly: config
// C3's short-length-scale settings (s = 0.5: |z|^2 up to 100 from a corner point, 50 from the middle) stay on the unit kernel.
// One block per feature, fixed summation order (thread t takes rows t, t + 256, ...; LDS tree): the same X gives the same centre.
__global__ __launch_bounds__(256) void centroid_kernel(const double *__restrict__ X, int n, int ldx, double *__restrict__ c) {
    __shared__ double red[256];
    const double *col = X + (size_t)blockIdx.x * ldx;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += col[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) c[blockIdx.x] = red[0] / (double)n;
}

}  // namespace

void gpk_centroid(hipStream_t s, const double *X, int n, int d, int ldx, double *c) {
    if (n <= 0 || d <= 0) return;
    hipLaunchKernelGGL(centroid_kernel, dim3(d), dim3(256), 0, s, X, n, ldx, c);
}

// Which form builds the Gram matrices: 2 = the unit kernel (whole exponent on the matrix cores, d <= 14: every BASELINE configuration),
// 1 = the LDS-staged matrix-core form (any d; the default from d = 16, where it beats the per-pair form 1.95 vs 2.68 ms at d = 32),
// 0 = the per-pair form in the reference's own operation order.  GPCORE_GRAM_MFMA=0 forces the per-pair form, =1 a matrix-core form
// at every d, =2 the LDS-staged one (read per call: the tests run all of them in one process).
static int gram_form(int d, const GramParams &p, int ldk) {
    // the unit kernel wants a finite ln sf^2 (sf = 0, inf, NaN: per-pair form, which multiplies by sf^2) and 32-bit store offsets
    const bool unit_ok = d <= GU_DMAX && std::isfinite(p.lnsf2) && ldk < (1 << 25);
    const char *e = getenv("GPCORE_GRAM_MFMA");
    if (e) {
        const int v = atoi(e);
        if (v == 0) return 0;
        return (v == 1 && d <= GU_DMAX) ? (unit_ok ? 2 : 0) : 1;
    }
    return d <= GU_DMAX ? (unit_ok ? 2 : 0) : (d >= 16 ? 1 : 0);
}

// `upw` units per wave: about 1.5 waves per resident slot (2 waves per SIMD at this register count), at most 32 -- wave counts that
// are a multiple of the 2048 slots run in lock step (all load, then all store) and measure up to 25 % slower
static int next_epoch() {
    static std::atomic<int> e{0};
    int v = e.fetch_add(1) + 1;
    return v <= 0 ? (e = 1, 1) : v;     // wraps after 2^31 calls
}

template <int MODE>
static void launch_unit(hipStream_t s, const double *Xr, int nr, int ldxr, const double *Xc, int nc, int ldxc, int d, const GramParams &p,
                        const double *cen, int ldcen, double *K, int ldk, int *far_flag, int epoch) {
    // the scan: row points and, for the cross form, column points
    hipLaunchKernelGGL(gram_far_kernel, dim3((nr + (MODE ? 0 : nc) + 255) / 256), dim3(256), 0, s, Xr, nr, ldxr, Xc, MODE ? 0 : nc, ldxc, d, p, cen, ldcen,
                       far_flag, epoch);
    const int nbr = (nr + 63) / 64, nbc = (nc + 63) / 64;
    const long total = MODE ? 2L * nbr * (nbr + 1) : 4L * nbr * nbc;
    long upw = (total + 3071) / 3072;
    if (upw > 32) upw = 32;
    if (const char *e = getenv("GPCORE_GRAM_UPW")) upw = atoi(e) > 0 ? atoi(e) : upw;
    const unsigned grid = (unsigned)((total + upw - 1) / upw);
#define GU_LAUNCH(KS, KR) hipLaunchKernelGGL((gram_unit_kernel<KS, KR, MODE>), dim3(grid), dim3(64), 0, s, Xr, nr, ldxr, Xc, nc, ldxc, d, p, cen, ldcen, K, ldk, nbc, (int)upw, total, far_flag, epoch)
    if (d <= 2) GU_LAUNCH(1, 1);
    else if (d <= 4) GU_LAUNCH(2, 1);
    else if (d <= 6) GU_LAUNCH(2, 2);
    else if (d <= 8) GU_LAUNCH(3, 2);
    else if (d <= 10) GU_LAUNCH(3, 3);
    else if (d <= 12) GU_LAUNCH(4, 3);
    else GU_LAUNCH(4, 4);
#undef GU_LAUNCH
}

void gpk_gram_sym(hipStream_t s, const double *X, int n, int d, int ldx, const double *theta, double *K, int ldk, int full, double extra_diag, int *far_flag,
                  const double *center) {
    const double *cen = center ? center : X;      // d doubles (gpk_centroid) or, without one, the first point
    const int ldcen = center ? 1 : ldx;
    GramParams p = make_params(theta, d, extra_diag);
    int nb = (n + GT - 1) / GT;
    const int form = far_flag ? gram_form(d, p, ldk) : 0;
    if (form == 2) {
        const int epoch = next_epoch();
        if (full) launch_unit<2>(s, X, n, ldx, X, n, ldx, d, p, cen, ldcen, K, ldk, far_flag, epoch);
        else launch_unit<1>(s, X, n, ldx, X, n, ldx, d, p, cen, ldcen, K, ldk, far_flag, epoch);
        hipLaunchKernelGGL(gram_rbf_kernel<true>, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, X, n, ldx, X, n, ldx, d, p, K, ldk, full, nb, far_flag, epoch);
        return;
    }
    if (form == 1) {
        hipLaunchKernelGGL(gram_mfma_kernel<true>, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, X, n, ldx, X, n, ldx, d, p, cen, ldcen, K, ldk, full, nb);
        return;
    }
    hipLaunchKernelGGL(gram_rbf_kernel<true>, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, X, n, ldx, X, n, ldx, d, p, K, ldk, full, nb);
}

// d K / d theta_pos (pos 1-based, vector order sf, l_1..l_d, sn), full symmetric
void gpk_dgram_sym(hipStream_t s, const double *X, int n, int d, int ldx, const double *theta, int pos, double *D, int ldd) {
    GramParams p = make_params(theta, d, 0.0);
    if (pos == 1) { p.dmode = 1; p.dcoef = 2.0 * theta[0]; }
    else if (pos < d + 2) { p.dmode = 2; p.dk = pos - 2; p.dcoef = (theta[0] * theta[0]) / (theta[pos - 1] * theta[pos - 1] * theta[pos - 1]); }
    else { p.dmode = 3; p.dcoef = 2.0 * theta[d + 1]; }
    int nb = (n + GT - 1) / GT;
    hipLaunchKernelGGL((gram_rbf_kernel<true, true>), dim3(nb * (nb + 1) / 2), dim3(256), 0, s, X, n, ldx, X, n, ldx, d, p, D, ldd, 1, nb);
}

void gpk_gram_cross(hipStream_t s, const double *Xs, int m, int ldxs, const double *X, int n, int ldx, int d, const double *theta, double *Ks, int ldks, int *far_flag,
                    const double *center) {
    const double *cen = center ? center : X;      // the TRAINING points' centre for both operands
    const int ldcen = center ? 1 : ldx;
    GramParams p = make_params(theta, d, 0.0);
    int nbr = (m + GT - 1) / GT, nbc = (n + GT - 1) / GT;
    const int form = far_flag ? gram_form(d, p, ldks) : 0;
    if (form == 2) {
        const int epoch = next_epoch();
        launch_unit<0>(s, Xs, m, ldxs, X, n, ldx, d, p, cen, ldcen, Ks, ldks, far_flag, epoch);
        hipLaunchKernelGGL(gram_rbf_kernel<false>, dim3(nbr * nbc), dim3(256), 0, s, Xs, m, ldxs, X, n, ldx, d, p, Ks, ldks, 1, nbr, far_flag, epoch);
        return;
    }
    if (form == 1) {
        hipLaunchKernelGGL(gram_mfma_kernel<false>, dim3(nbr * nbc), dim3(256), 0, s, Xs, m, ldxs, X, n, ldx, d, p, cen, ldcen, Ks, ldks, 1, nbr);
        return;
    }
    hipLaunchKernelGGL(gram_rbf_kernel<false>, dim3(nbr * nbc), dim3(256), 0, s, Xs, m, ldxs, X, n, ldx, d, p, Ks, ldks, 1, nbr);
}

void gpk_pad_identity(hipStream_t s, double *A, int n, int np, int lda) {
    if (np == n) return;
    hipLaunchKernelGGL(pad_identity_kernel, dim3(256), dim3(256), 0, s, A, n, np, lda);
}

void gpk_zero_upper(hipStream_t s, double *A, int n, int lda) {
    if (n < 2) return;
    hipLaunchKernelGGL(zero_upper_kernel, dim3(8, n < 65535 ? n : 65535), dim3(256), 0, s, A, n, lda);
}

void gpk_fill(hipStream_t s, double *p, size_t count, double v) {
    if (!count) return;
    hipLaunchKernelGGL(fill_kernel, dim3(512), dim3(256), 0, s, p, count, v);
}

void gpk_copy_2d(hipStream_t s, double *dst, int ldd, const double *src, int lds, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(copy2d_kernel, dim3(1024), dim3(256), 0, s, dst, ldd, src, lds, rows, cols);
}

void gpk_copy_strided(hipStream_t s, double *dst, size_t dst_stride, const double *src, size_t src_stride, int count) {
    if (count <= 0) return;
    hipLaunchKernelGGL(copy_strided_kernel, dim3((count + 255) / 256), dim3(256), 0, s, dst, dst_stride, src, src_stride, count);
}
