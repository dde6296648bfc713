// Batched small-n posterior workloads: the callers of GpPredictor.computePosterior with ONE test point
// per call and a few hundred training points --
//   GP-UCB   gp/optimization/GPOptimizer.scala:47-109          (refit per appended point, c L-BFGS runs over the UCB surface)
//   GP-UKF   dynamicalsystems/filtering/GPUnscentedKalmanFilter.scala:63-147   (one GP per state / observation dimension,
//            2 D + 1 sigma points per time step, mean(0) and sigma(0,0) of each)
// -- restructured as batches: G models over ONE set of training inputs stay resident with L^-1 next to L (what
// GPOptimizer.maximizeUCB builds with invTriangular per call, :85), so every (model, test point) pair is one workgroup of ONE
// launch: k*, mean = k* . alpha, v = L^-1 k* (a triangular matrix-vector product, no substitution chain), var = k** - |v|^2,
// and for the UCB objective its input gradient through GaussianRbfKernel.gradient (utils/KernelRequisites.scala:95-107).
// A point is APPENDED by a rank-1 extension of L, L^-1 and alpha (O(n^2)) instead of the reference's O(n^3) refit per
// iteration (GPOptimizer.scala:51).
#include "gpcore_internal.h"

#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

struct gp_small {
    gp_ctx *ctx = nullptr;
    int n = 0, cap = 0, d = 0, G = 0;
    double sigma_noise = NAN;
    double *dX = nullptr;       // cap x d, ld = cap
    double *dY = nullptr;       // G x cap (targets of model g at dY + g * cap); absent (all zero) when built from factors
    double *dL = nullptr;       // G x (cap x cap), ld = cap, lower
    double *dLinv = nullptr;    // G x (cap x cap), ld = cap, lower
    double *dalpha = nullptr;   // G x cap
    double *dtheta = nullptr;   // G x (d + 2)
    bool has_y = false;
    std::vector<double> thetas;
};

namespace {

constexpr int SM_THREADS = 256;

__device__ __forceinline__ double block_sum(double v, double *red, int tid) {   // fixed-order tree, result broadcast
    red[tid] = v;
    __syncthreads();
    for (int s = SM_THREADS / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// k(x, X_j, false) in the reference's operation order (KernelRequisites.scala:66-72,109-113)
__device__ __forceinline__ double rbf_pair(const double *xs, const double *X, int ldx, int j, int d, const double *th) {
    double acc = 0.0;
    for (int k = 0; k < d; ++k) {
        const double diff = xs[k] - X[j + (size_t)k * ldx];
        const double ls = th[1 + k];
        acc = acc + (diff * (1.0 / (ls * ls))) * diff;
    }
    return th[0] * th[0] * exp(-0.5 * acc);
}

// One workgroup per (test point, model).  UCB: also value = mean + kappa sqrt(var) and its gradient w.r.t. the test point.
//   mean[g * m + i], var[g * m + i];  UCB (one model `g0`): value[i], grad[i * d + k]
template <bool UCB>
__global__ __launch_bounds__(SM_THREADS) void small_posterior_kernel(int n, int cap, int d, int g0, const double *__restrict__ X,
                                                                     const double *__restrict__ Linv, const double *__restrict__ alpha,
                                                                     const double *__restrict__ theta, const double *__restrict__ Xs, int m,
                                                                     int ldxs, double kappa, double *__restrict__ mean, double *__restrict__ var,
                                                                     double *__restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *ks = sm, *v = sm + cap, *w = v + cap, *xs = w + (UCB ? cap : 0), *red = xs + 64;
    const int tid = threadIdx.x, i = blockIdx.x, g = UCB ? g0 : blockIdx.y;
    const double *th = theta + (size_t)g * (d + 2), *Li = Linv + (size_t)g * cap * cap, *al = alpha + (size_t)g * cap;
    if (tid < d) xs[tid] = Xs[i + (size_t)tid * ldxs];
    __syncthreads();
    double pm = 0.0;
    for (int j = tid; j < n; j += SM_THREADS) {
        const double kv = rbf_pair(xs, X, cap, j, d, th);
        ks[j] = kv;
        pm = fma(kv, al[j], pm);
    }
    const double mu = block_sum(pm, red, tid);        // also orders ks[] before the product below
    double ps = 0.0;
    for (int r = tid; r < n; r += SM_THREADS) {       // v = L^-1 k*: row r, columns 0..r (coalesced over r)
        double acc = 0.0;
        for (int c = 0; c <= r; ++c) acc = fma(Li[r + (size_t)c * cap], ks[c], acc);
        v[r] = acc;
        ps = fma(acc, acc, ps);
    }
    const double vv = block_sum(ps, red, tid);
    const double sf = th[0], sn = th[d + 1];
    const double sig = (sf * sf + sn * sn) - vv;      // buildKernelMatrix(kernel, testData) carries sn^2 on its diagonal (:56)
    if (!UCB) {
        if (tid == 0) { mean[(size_t)g * m + i] = mu; var[(size_t)g * m + i] = sig; }
        return;
    }
    // w = L^-T v: column c of L^-1 against v, one wave per column, lanes striding the rows
    const int lane = tid & 63, wave = tid >> 6;
    for (int c = wave; c < n; c += SM_THREADS / 64) {
        double acc = 0.0;
        for (int r = c + lane; r < n; r += 64) acc = fma(Li[r + (size_t)c * cap], v[r], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) w[c] = acc;
    }
    __syncthreads();
    // D(j, k) = d k(x*, X_j) / d x*_k = -k*_j (x*_k - X_jk) / l_k^2   (gradient(afterFirstArg = true), :95-107);
    // d mean / d x*_k = sum_j D(j,k) alpha_j;  d var / d x*_k = -2 sum_j D(j,k) w_j  (GPOptimizer.scala:97-101, derAfterVarFirst = 0)
    const double sd = sqrt(sig), coeff = kappa / (2.0 * sd);
    for (int k = wave; k < d; k += SM_THREADS / 64) {
        const double ls = th[1 + k], inv = 1.0 / (ls * ls), xk = xs[k];
        double dm = 0.0, dv = 0.0;
        for (int j = lane; j < n; j += 64) {
            const double D = ((xk - X[j + (size_t)k * cap]) * inv) * (-ks[j]);
            dm = fma(D, al[j], dm);
            dv = fma(D, w[j], dv);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { dm += __shfl_xor(dm, o); dv += __shfl_xor(dv, o); }
        if (lane == 0) grad[(size_t)i * d + k] = dm + (-2.0 * dv) * coeff;
    }
    if (tid == 0) { mean[i] = mu + kappa * sd; if (var) var[i] = sig; }
}

// L^-1 of a lower-triangular n x n matrix, column by column: workgroup (j, g) solves L x = e_j by column-oriented forward
// substitution (MatrixUtils.invTriangular solves against the identity the same way, utils/MatrixUtils.scala:106-113).
__global__ __launch_bounds__(SM_THREADS) void small_trinv_kernel(int n, int cap, const double *__restrict__ L, double *__restrict__ Linv) {
    extern __shared__ __attribute__((aligned(16))) double x[];
    const int tid = threadIdx.x, j = blockIdx.x;
    const double *Lg = L + (size_t)blockIdx.y * cap * cap;
    double *Og = Linv + (size_t)blockIdx.y * cap * cap;
    for (int r = tid; r < n; r += SM_THREADS) x[r] = (r == j) ? 1.0 : 0.0;
    __syncthreads();
    for (int c = j; c < n; ++c) {
        const double xc = x[c] / Lg[c + (size_t)c * cap];
        __syncthreads();
        if (tid == 0) x[c] = xc;
        for (int r = c + 1 + tid; r < n; r += SM_THREADS) x[r] = fma(-Lg[r + (size_t)c * cap], xc, x[r]);
        __syncthreads();
    }
    for (int r = tid; r < cap; r += SM_THREADS) Og[r + (size_t)j * cap] = (r < n && r >= j) ? x[r] : 0.0;
}

// Rank-1 extension by the training point already stored in row n of X (one workgroup per model):
//   k = k(X[0:n], x_new), l = L^-1 k, lambda = sqrt(k(x_new, x_new) + sn^2 (+ sigmaNoise) - |l|^2)
//   L    <- [[L, 0], [l^T, lambda]]
//   L^-1 <- [[L^-1, 0], [-(l^T L^-1) / lambda, 1 / lambda]]
//   alpha = L^-T (L^-1 y) over the n + 1 points (targets in Y)
// info[g] = n + 1 (1-based failing pivot, as breeze's cholesky would fail on the extended matrix) when lambda^2 <= 0.
__global__ __launch_bounds__(SM_THREADS) void small_append_kernel(int n, int cap, int d, double extra, const double *__restrict__ X,
                                                                  const double *__restrict__ Y, double *__restrict__ L, double *__restrict__ Linv,
                                                                  double *__restrict__ alpha, const double *__restrict__ theta, int *__restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *ks = sm, *l = sm + cap, *t = l + cap, *xs = t + cap, *red = xs + 64;
    const int tid = threadIdx.x, g = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const double *th = theta + (size_t)g * (d + 2);
    double *Lg = L + (size_t)g * cap * cap, *Li = Linv + (size_t)g * cap * cap, *al = alpha + (size_t)g * cap;
    const double *y = Y + (size_t)g * cap;
    if (tid < d) xs[tid] = X[n + (size_t)tid * cap];
    __syncthreads();
    for (int j = tid; j < n; j += SM_THREADS) ks[j] = rbf_pair(xs, X, cap, j, d, th);
    __syncthreads();
    double ps = 0.0;
    for (int r = tid; r < n; r += SM_THREADS) {
        double acc = 0.0;
        for (int c = 0; c <= r; ++c) acc = fma(Li[r + (size_t)c * cap], ks[c], acc);
        l[r] = acc;
        ps = fma(acc, acc, ps);
    }
    const double ll = block_sum(ps, red, tid);
    const double sf = th[0], sn = th[d + 1];
    const double lam2 = ((sf * sf + sn * sn) + extra) - ll;
    if (!(lam2 > 0.0)) { if (tid == 0) info[g] = n + 1; return; }
    const double lam = sqrt(lam2), ilam = 1.0 / lam;
    for (int c = tid; c < n; c += SM_THREADS) Lg[n + (size_t)c * cap] = l[c];
    if (tid == 0) { Lg[n + (size_t)n * cap] = lam; Li[n + (size_t)n * cap] = ilam; }
    for (int c = wave; c < n; c += SM_THREADS / 64) {          // row n of L^-1: -(l^T L^-1)_c / lambda
        double acc = 0.0;
        for (int r = c + lane; r < n; r += 64) acc = fma(l[r], Li[r + (size_t)c * cap], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) Li[n + (size_t)c * cap] = -acc * ilam;
    }
    __threadfence_block();   // row n of L^-1 is read back by other threads of this workgroup
    __syncthreads();
    const int n1 = n + 1;
    for (int r = tid; r < n1; r += SM_THREADS) {               // t = L^-1 y
        double acc = 0.0;
        for (int c = 0; c <= r; ++c) acc = fma(Li[r + (size_t)c * cap], y[c], acc);
        t[r] = acc;
    }
    __syncthreads();
    for (int c = wave; c < n1; c += SM_THREADS / 64) {         // alpha = L^-T t
        double acc = 0.0;
        for (int r = c + lane; r < n1; r += 64) acc = fma(Li[r + (size_t)c * cap], t[r], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) al[c] = acc;
    }
}

size_t small_lds(int cap, int arrays) { return sizeof(double) * ((size_t)arrays * cap + 64 + SM_THREADS + 8); }

gp_status small_alloc(gp_ctx *ctx, int n, int d, int G, int capacity, gp_small **out) {
    gp_small *s = new (std::nothrow) gp_small();
    if (!s) return GP_ENOMEM;
    s->ctx = ctx, s->n = n, s->d = d, s->G = G;
    s->cap = std::max(n, capacity);
    const size_t cap = s->cap, cc = cap * cap;
    hipError_t e = hipMalloc(&s->dX, sizeof(double) * cap * d);
    if (e == hipSuccess) e = hipMalloc(&s->dY, sizeof(double) * cap * G);
    if (e == hipSuccess) e = hipMalloc(&s->dL, sizeof(double) * cc * G);
    if (e == hipSuccess) e = hipMalloc(&s->dLinv, sizeof(double) * cc * G);
    if (e == hipSuccess) e = hipMalloc(&s->dalpha, sizeof(double) * cap * G);
    if (e == hipSuccess) e = hipMalloc(&s->dtheta, sizeof(double) * (size_t)(d + 2) * G);
    if (e == hipSuccess) e = hipMemsetAsync(s->dX, 0, sizeof(double) * cap * d, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->dY, 0, sizeof(double) * cap * G, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->dL, 0, sizeof(double) * cc * G, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->dLinv, 0, sizeof(double) * cc * G, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(s->dalpha, 0, sizeof(double) * cap * G, ctx->stream);
    if (e != hipSuccess) {
        GP_SET_ERR(ctx, "small-model allocation (n=%d, capacity=%d, G=%d) failed: %s", n, s->cap, G, hipGetErrorString(e));
        gp_small_destroy(s);
        return GP_ENOMEM;
    }
    *out = s;
    return GP_OK;
}

gp_status small_set_lds(gp_ctx *ctx, int cap) {
    static int done_for = 0;
    if (cap <= done_for) return GP_OK;
    GP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(small_posterior_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds(GP_SMALL_MAX_N, 2)));
    GP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(small_posterior_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds(GP_SMALL_MAX_N, 3)));
    GP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(small_append_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)small_lds(GP_SMALL_MAX_N, 3)));
    done_for = GP_SMALL_MAX_N;
    return GP_OK;
}

void small_invert(gp_small *s) {
    hipLaunchKernelGGL(small_trinv_kernel, dim3(s->n, s->G), dim3(SM_THREADS), sizeof(double) * s->cap, s->ctx->stream, s->n, s->cap, s->dL, s->dLinv);
}

}  // namespace

extern "C" {

gp_status gp_small_from_factors(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *thetas, int G, const double *Ls, int ldl,
                                const double *alphas, int capacity, gp_small **out) {
    if (!ctx || !out) return GP_EINVAL;
    *out = nullptr;
    GP_REQUIRE(ctx, X && thetas && Ls && alphas, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && G >= 1 && ldx >= n && ldl >= n && std::max(n, capacity) <= GP_SMALL_MAX_N,
               "need 1 <= n <= capacity <= GP_SMALL_MAX_N, 1 <= d <= 64, G >= 1");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_TRY(small_set_lds(ctx, std::max(n, capacity)));
    gp_small *s = nullptr;
    GP_TRY(small_alloc(ctx, n, d, G, capacity, &s));
    const int cap = s->cap, P = d + 2;
    s->thetas.assign(thetas, thetas + (size_t)G * P);
    gp_status st = gpi_upload_2d(ctx, s->dX, cap, X, ldx, n, d);
    if (st == GP_OK) st = gpi_upload_2d(ctx, s->dtheta, P, thetas, P, P, G);
    for (int g = 0; g < G && st == GP_OK; ++g) {
        st = gpi_upload_2d(ctx, s->dL + (size_t)g * cap * cap, cap, Ls + (size_t)g * ldl * n, ldl, n, n);
        if (st == GP_OK) st = gpi_upload_2d(ctx, s->dalpha + (size_t)g * cap, cap, alphas + (size_t)g * n, n, n, 1);
        if (st == GP_OK) gpk_zero_upper(ctx->stream, s->dL + (size_t)g * cap * cap, n, cap);
    }
    if (st != GP_OK) { gp_small_destroy(s); return st; }
    small_invert(s);
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = s;
    return GP_OK;
}

gp_status gp_small_fit(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *Y, int ldy, int G, const double *thetas,
                       double sigma_noise, int capacity, gp_small **out, int *info) {
    if (!ctx || !out) return GP_EINVAL;
    *out = nullptr;
    if (info) *info = 0;
    GP_REQUIRE(ctx, X && Y && thetas, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && G >= 1 && ldx >= n && ldy >= n && std::max(n, capacity) <= GP_SMALL_MAX_N,
               "need 1 <= n <= capacity <= GP_SMALL_MAX_N, 1 <= d <= 64, G >= 1");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_TRY(small_set_lds(ctx, std::max(n, capacity)));
    gp_small *s = nullptr;
    GP_TRY(small_alloc(ctx, n, d, G, capacity, &s));
    const int cap = s->cap, P = d + 2;
    s->thetas.assign(thetas, thetas + (size_t)G * P);
    s->sigma_noise = sigma_noise;
    s->has_y = true;
    gp_status st = gpi_upload_2d(ctx, s->dX, cap, X, ldx, n, d);
    if (st == GP_OK) st = gpi_upload_2d(ctx, s->dtheta, P, thetas, P, P, G);
    if (st == GP_OK) st = gpi_upload_2d(ctx, s->dY, cap, Y, ldy, n, G);
    // one blocked fit per output dimension on the shared inputs (GPUnscentedKalmanFilter.learnInputOutput :116-129); the
    // factor, alpha and L^-1 are then kept in the compact capacity-strided layout the batched kernels read
    for (int g = 0; g < G && st == GP_OK; ++g) {
        gp_model *m = nullptr;
        int h = 0;
        st = gp_fit_rbf_dev(ctx, s->dX, n, d, cap, s->dY + (size_t)g * cap, thetas + (size_t)g * P, sigma_noise, &m, &h);
        if (st != GP_OK) { if (info) *info = h; break; }
        gpk_copy_2d(ctx->stream, s->dL + (size_t)g * cap * cap, cap, m->dL, m->ldl, n, n);
        gpk_zero_upper(ctx->stream, s->dL + (size_t)g * cap * cap, n, cap);
        st = gpi_model_alpha(m, s->dalpha + (size_t)g * cap);
        gp_model_destroy(m);
    }
    if (st != GP_OK) { gp_small_destroy(s); return st; }
    small_invert(s);
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = s;
    return GP_OK;
}

void gp_small_destroy(gp_small *s) {
    if (!s) return;
    if (s->ctx) { (void)hipSetDevice(s->ctx->device); (void)hipStreamSynchronize(s->ctx->stream); }
    void *ptrs[] = {s->dX, s->dY, s->dL, s->dLinv, s->dalpha, s->dtheta};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete s;
}

gp_status gp_small_size(const gp_small *s, int *n, int *capacity, int *G) {
    if (!s) return GP_EINVAL;
    if (n) *n = s->n;
    if (capacity) *capacity = s->cap;
    if (G) *G = s->G;
    return GP_OK;
}

gp_status gp_small_get(gp_small *s, int g, int what, double *out, int ld) {
    if (!s || !out) return GP_EINVAL;
    gp_ctx *ctx = s->ctx;
    GP_REQUIRE(ctx, g >= 0 && g < s->G, "model index out of range");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t cc = (size_t)s->cap * s->cap;
    switch (what) {
        case GP_SMALL_GET_L: GP_REQUIRE(ctx, ld >= s->n, "ld < n"); return gpi_download_2d(ctx, out, ld, s->dL + g * cc, s->cap, s->n, s->n);
        case GP_SMALL_GET_LINV: GP_REQUIRE(ctx, ld >= s->n, "ld < n"); return gpi_download_2d(ctx, out, ld, s->dLinv + g * cc, s->cap, s->n, s->n);
        case GP_SMALL_GET_ALPHA: return gpi_download_2d(ctx, out, s->n, s->dalpha + (size_t)g * s->cap, s->cap, s->n, 1);
        default: GP_SET_ERR(ctx, "unknown selector %d", what); return GP_EINVAL;
    }
}

gp_status gp_small_posterior(gp_small *s, const double *Xs, int m, int ldxs, double *mean, double *var) {
    if (!s) return GP_EINVAL;
    gp_ctx *ctx = s->ctx;
    GP_REQUIRE(ctx, Xs && mean && var && m >= 0 && ldxs >= m, "bad arguments");
    if (m == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dXs, *dout;
    GP_TRY(gpi_ws_get(ctx, WS_A, sizeof(double) * (size_t)m * s->d, &dXs));
    GP_TRY(gpi_ws_get(ctx, WS_C, sizeof(double) * (size_t)2 * m * s->G, &dout));
    GP_TRY(gpi_upload_2d(ctx, dXs, m, Xs, ldxs, m, s->d));
    hipLaunchKernelGGL(small_posterior_kernel<false>, dim3(m, s->G), dim3(SM_THREADS), small_lds(s->cap, 2), ctx->stream, s->n, s->cap, s->d, 0,
                       s->dX, s->dLinv, s->dalpha, s->dtheta, dXs, m, m, 0.0, dout, dout + (size_t)m * s->G, nullptr);
    GP_LAUNCH_CHECK(ctx);
    GP_TRY(gpi_download_2d(ctx, mean, m * s->G, dout, m * s->G, m * s->G, 1));
    return gpi_download_2d(ctx, var, m * s->G, dout + (size_t)m * s->G, m * s->G, m * s->G, 1);
}

gp_status gp_small_ucb(gp_small *s, int g, const double *Xs, int m, int ldxs, double kappa, double *value, double *grad) {
    if (!s) return GP_EINVAL;
    gp_ctx *ctx = s->ctx;
    GP_REQUIRE(ctx, Xs && value && grad && m >= 0 && ldxs >= m && g >= 0 && g < s->G, "bad arguments");
    if (m == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dXs, *dout;
    GP_TRY(gpi_ws_get(ctx, WS_A, sizeof(double) * (size_t)m * s->d, &dXs));
    GP_TRY(gpi_ws_get(ctx, WS_C, sizeof(double) * (size_t)m * (1 + s->d), &dout));
    GP_TRY(gpi_upload_2d(ctx, dXs, m, Xs, ldxs, m, s->d));
    hipLaunchKernelGGL(small_posterior_kernel<true>, dim3(m), dim3(SM_THREADS), small_lds(s->cap, 3), ctx->stream, s->n, s->cap, s->d, g, s->dX,
                       s->dLinv, s->dalpha, s->dtheta, dXs, m, m, kappa, dout, nullptr, dout + m);
    GP_LAUNCH_CHECK(ctx);
    GP_TRY(gpi_download_2d(ctx, value, m, dout, m, m, 1));
    return gpi_download_2d(ctx, grad, m * s->d, dout + m, m * s->d, m * s->d, 1);
}

gp_status gp_small_append(gp_small *s, const double *x_new, const double *y_new, int *info) {
    if (!s) return GP_EINVAL;
    gp_ctx *ctx = s->ctx;
    if (info) *info = 0;
    GP_REQUIRE(ctx, x_new && y_new, "null pointer");
    GP_REQUIRE(ctx, s->has_y, "models built from caller-held factors carry no targets: alpha cannot be extended");
    GP_REQUIRE(ctx, s->n < s->cap, "capacity exhausted: create the batch with a larger capacity");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    const int n = s->n, cap = s->cap;
    GP_TRY(gpi_upload_2d(ctx, s->dX + n, cap, x_new, 1, 1, s->d));          // row n of X
    GP_TRY(gpi_upload_2d(ctx, s->dY + n, cap, y_new, 1, 1, s->G));          // entry n of every model's targets
    int *dinfo = ctx->d_info;
    std::vector<int> hinfo(s->G, 0);
    double *dscr;
    GP_TRY(gpi_ws_get(ctx, WS_C, sizeof(int) * (size_t)s->G + 16, &dscr));
    dinfo = reinterpret_cast<int *>(dscr);
    GP_HIP(ctx, hipMemsetAsync(dinfo, 0, sizeof(int) * s->G, ctx->stream));
    hipLaunchKernelGGL(small_append_kernel, dim3(s->G), dim3(SM_THREADS), small_lds(cap, 3), ctx->stream, n, cap, s->d,
                       std::isnan(s->sigma_noise) ? 0.0 : s->sigma_noise, s->dX, s->dY, s->dL, s->dLinv, s->dalpha, s->dtheta, dinfo);
    GP_LAUNCH_CHECK(ctx);
    GP_HIP(ctx, hipMemcpyAsync(hinfo.data(), dinfo, sizeof(int) * s->G, hipMemcpyDeviceToHost, ctx->stream));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int g = 0; g < s->G; ++g)
        if (hinfo[g]) {
            if (info) *info = hinfo[g];
            GP_SET_ERR(ctx, "model %d: the extended matrix is not positive definite at pivot %d", g, hinfo[g]);
            return GP_ENOTPD;   // nothing was committed for any model: n is unchanged
        }
    s->n = n + 1;
    return GP_OK;
}

// GPOptimizer.maximizeUCB (gp/optimization/GPOptimizer.scala:82-109) for `c` starting points at once: the reference runs
// gradientOptimizer.maximize (BreezeLbfgsOptimizer, m = 4: optimization/Optimization.scala:30-63) once per start, every
// objective evaluation one computePosterior with a single test point; here the c L-BFGS runs advance in LOCKSTEP and every
// iteration evaluates all their trial steps (c x NC points) in ONE launch.  Returns the best point any evaluation saw
// (the rule of Optimization.scala:44-46,52-55 applied across all starts) and its UCB value.
gp_status gp_small_maximize_ucb(gp_small *s, int g, const double *starts, int c, int lds, double kappa, int max_iter, int history,
                                double *best_x, double *best_val, int *evals_out) {
    if (!s) return GP_EINVAL;
    gp_ctx *ctx = s->ctx;
    GP_REQUIRE(ctx, starts && best_x && c >= 1 && lds >= c && g >= 0 && g < s->G && max_iter >= 0 && history >= 1, "bad arguments");
    const int d = s->d;
    constexpr int NC = 4;   // trial steps per run and iteration: 1, 1/2, 1/4, 1/8 of the quasi-Newton step
    struct run_t {
        std::vector<double> x, gr, dir;
        double f = 0.0, slope = 0.0;
        std::vector<std::vector<double>> S, Y;
        std::vector<double> rho;
        bool done = false;
    };
    std::vector<run_t> runs(c);
    std::vector<double> pts((size_t)c * NC * d), val((size_t)c * NC), grd((size_t)c * NC * d);
    int evals = 0;
    double bf = -INFINITY;
    std::vector<double> bx(d, 0.0);
    auto eval = [&](int count) -> gp_status {   // pts: count points as rows of a (count x d) column-major matrix
        GP_TRY(gp_small_ucb(s, g, pts.data(), count, count, kappa, val.data(), grd.data()));
        evals += count;
        return GP_OK;
    };
    auto note = [&](double v, const double *x) { if (std::isfinite(v) && v > bf) { bf = v; std::copy(x, x + d, bx.begin()); } };
    for (int r = 0; r < c; ++r)
        for (int k = 0; k < d; ++k) pts[r + (size_t)k * c] = starts[r + (size_t)k * lds];
    GP_TRY(eval(c));
    for (int r = 0; r < c; ++r) {
        run_t &R = runs[r];
        R.x.resize(d), R.gr.resize(d), R.dir.resize(d);
        for (int k = 0; k < d; ++k) { R.x[k] = starts[r + (size_t)k * lds]; R.gr[k] = -grd[(size_t)r * d + k]; }   // minimise f = -UCB
        R.f = -val[r];
        if (!std::isfinite(R.f)) R.done = true; else note(val[r], R.x.data());
    }
    auto dot = [&](const std::vector<double> &a, const std::vector<double> &b) { double t = 0.0; for (int k = 0; k < d; ++k) t += a[k] * b[k]; return t; };
    for (int it = 0; it < max_iter; ++it) {
        int active = 0;
        for (int r = 0; r < c; ++r) {
            run_t &R = runs[r];
            if (R.done) continue;
            const double gn = std::sqrt(dot(R.gr, R.gr));
            if (gn <= 1e-10 * std::max(1.0, std::fabs(R.f))) { R.done = true; continue; }
            std::vector<double> q = R.gr, al(R.S.size());
            for (int h = (int)R.S.size() - 1; h >= 0; --h) { al[h] = R.rho[h] * dot(R.S[h], q); for (int k = 0; k < d; ++k) q[k] -= al[h] * R.Y[h][k]; }
            const double sc = R.S.empty() ? 1.0 / gn : dot(R.S.back(), R.Y.back()) / dot(R.Y.back(), R.Y.back());
            for (int k = 0; k < d; ++k) q[k] *= sc;
            for (size_t h = 0; h < R.S.size(); ++h) { const double be = R.rho[h] * dot(R.Y[h], q); for (int k = 0; k < d; ++k) q[k] += R.S[h][k] * (al[h] - be); }
            for (int k = 0; k < d; ++k) R.dir[k] = -q[k];
            R.slope = dot(R.gr, R.dir);
            if (!(R.slope < 0.0)) { R.S.clear(), R.Y.clear(), R.rho.clear(); for (int k = 0; k < d; ++k) R.dir[k] = -R.gr[k] / gn; R.slope = -gn; }
            ++active;
        }
        if (!active) break;
        // trial points of every active run: row index = slot * NC + t
        std::vector<int> slot_of;
        for (int r = 0; r < c; ++r) if (!runs[r].done) slot_of.push_back(r);
        const int count = (int)slot_of.size() * NC;
        for (size_t sidx = 0; sidx < slot_of.size(); ++sidx)
            for (int t = 0; t < NC; ++t)
                for (int k = 0; k < d; ++k)
                    pts[(sidx * NC + t) + (size_t)k * count] = runs[slot_of[sidx]].x[k] + std::ldexp(1.0, -t) * runs[slot_of[sidx]].dir[k];
        GP_TRY(eval(count));
        for (size_t sidx = 0; sidx < slot_of.size(); ++sidx) {
            run_t &R = runs[slot_of[sidx]];
            int chosen = -1;
            for (int t = 0; t < NC; ++t) {
                const size_t e = sidx * NC + t;
                std::vector<double> xt(d);
                for (int k = 0; k < d; ++k) xt[k] = pts[e + (size_t)k * count];
                note(val[e], xt.data());
                if (chosen < 0 && std::isfinite(val[e]) && -val[e] <= R.f + 1e-4 * std::ldexp(1.0, -t) * R.slope) chosen = t;
            }
            if (chosen < 0) { R.done = true; continue; }
            const size_t e = sidx * NC + chosen;
            std::vector<double> sv(d), yv(d);
            for (int k = 0; k < d; ++k) {
                sv[k] = std::ldexp(1.0, -chosen) * R.dir[k];
                yv[k] = -grd[e * d + k] - R.gr[k];
                R.x[k] += sv[k];
                R.gr[k] = -grd[e * d + k];
            }
            const double fnew = -val[e], sy = dot(sv, yv);
            if (sy > 1e-12 * std::sqrt(dot(sv, sv) * dot(yv, yv))) {
                if ((int)R.S.size() == history) { R.S.erase(R.S.begin()); R.Y.erase(R.Y.begin()); R.rho.erase(R.rho.begin()); }
                R.S.push_back(sv), R.Y.push_back(yv), R.rho.push_back(1.0 / sy);
            }
            if (std::fabs(R.f - fnew) <= 1e-12 * std::max(1.0, std::fabs(R.f))) R.done = true;
            R.f = fnew;
        }
    }
    std::copy(bx.begin(), bx.end(), best_x);
    if (best_val) *best_val = bf;
    if (evals_out) *evals_out = evals;
    return GP_OK;
}

}  // extern "C"
