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
                                                       int nbr) {
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
    for (int k = 0; k < GP_DMAX; ++k) p.inv_ls2[k] = p.inv_ls[k] = 0.0;
    for (int k = 0; k < d; ++k) {
        p.inv_ls2[k] = 1.0 / (theta[1 + k] * theta[1 + k]);
        p.inv_ls[k] = 1.0 / fabs(theta[1 + k]);
    }
    return p;
}

}  // namespace

// Which form builds the Gram matrices.  Measured on MI355X (tools/gram_perf.py, n = 8192, m = 65536): at d = 8 the per-pair form
// wins (cross-Gram 1.21 vs 1.24 ms, lower Gram 84 vs 104 us: both are bound by VALU issue and by the load -> barrier -> store
// life cycle of a tile, not by the store stream -- a store-only kernel with the same tile pattern reaches 5.4 TB/s, memset
// 6.4 TB/s), at d = 32 the matrix-core form wins (1.95 vs 2.68 ms), so the default switches at d >= 16.
// GPCORE_GRAM_MFMA=1 / 0 forces one form (read per call: the tests run both in one process).
static bool gram_mfma(int d) {
    const char *e = getenv("GPCORE_GRAM_MFMA");
    return e ? atoi(e) != 0 : d >= 16;
}

void gpk_gram_sym(hipStream_t s, const double *X, int n, int d, int ldx, const double *theta, double *K, int ldk, int full, double extra_diag) {
    GramParams p = make_params(theta, d, extra_diag);
    int nb = (n + GT - 1) / GT;
    if (gram_mfma(d)) {
        hipLaunchKernelGGL(gram_mfma_kernel<true>, dim3(nb * (nb + 1) / 2), dim3(256), 0, s, X, n, ldx, X, n, ldx, d, p, X, ldx, K, ldk, full, nb);
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

void gpk_gram_cross(hipStream_t s, const double *Xs, int m, int ldxs, const double *X, int n, int ldx, int d, const double *theta, double *Ks, int ldks) {
    GramParams p = make_params(theta, d, 0.0);
    int nbr = (m + GT - 1) / GT, nbc = (n + GT - 1) / GT;
    if (gram_mfma(d)) {   // centre = first TRAINING point for both operands
        hipLaunchKernelGGL(gram_mfma_kernel<false>, dim3(nbr * nbc), dim3(256), 0, s, Xs, m, ldxs, X, n, ldx, d, p, X, ldx, Ks, ldks, 1, nbr);
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
