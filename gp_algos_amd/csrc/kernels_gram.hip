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
    for (int k = 0; k < GP_DMAX; ++k) p.inv_ls2[k] = 0.0;
    for (int k = 0; k < d; ++k) p.inv_ls2[k] = 1.0 / (theta[1 + k] * theta[1 + k]);
    return p;
}

}  // namespace

void gpk_gram_sym(hipStream_t s, const double *X, int n, int d, int ldx, const double *theta, double *K, int ldk, int full, double extra_diag) {
    GramParams p = make_params(theta, d, extra_diag);
    int nb = (n + GT - 1) / GT;
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
