// The composite CO2 kernel of gp/regression/Co2Prediction.scala:29-137 on the device: one-dimensional inputs,
// 11 hyper-parameters hp1..hp11 (1-based, Co2HyperParams.getAtPosition :24):
//   k1 = hp1^2 exp(-r^2 / (2 hp2^2))                                   long-term trend (squared exponential)
//   k2 = hp3^2 exp(-r^2 / (2 hp4^2) - 2 sin(pi r)^2 / hp5^2)           decaying periodic term
//   k3 = hp6^2 (1 + r^2 / (2 hp8 hp7^2))^(-hp8)                        rational quadratic
//   k4 = hp9^2 exp(-r^2 / (2 hp10^2)) + hp11^2 [same index]            short-term term + noise
// with r = x1 - x2, evaluated in the reference's operation order (:39-56), and the eleven derAfterHyperParam branches (:69-137).
// Element-wise kernels: thread per entry, i fastest (coalesced column-major stores); the symmetric builders evaluate i >= j once
// and write both (i, j) and (j, i), like MatrixUtils.buildKernelMatrix / buildMatrixWithFunc (utils/MatrixUtils.scala:57-84).
#include "gpcore_internal.h"

namespace {

struct Co2Params { double hp[11]; };

__device__ __forceinline__ double co2_apply(double x1, double x2, bool same, const Co2Params &p) {
    const double hp1 = p.hp[0], hp2 = p.hp[1], hp3 = p.hp[2], hp4 = p.hp[3], hp5 = p.hp[4], hp6 = p.hp[5], hp7 = p.hp[6], hp8 = p.hp[7],
                 hp9 = p.hp[8], hp10 = p.hp[9], hp11 = p.hp[10];
    const double xDiff = x1 - x2, xDiffSq = (x1 - x2) * (x1 - x2);
    const double k1Val = hp1 * hp1 * exp(-xDiffSq / (2 * hp2 * hp2));
    const double sinVal = sin(M_PI * xDiff);
    const double k2Val = hp3 * hp3 * exp((-xDiffSq / (2 * hp4 * hp4)) - 2 * sinVal * sinVal / (hp5 * hp5));
    const double k3Pow1 = 1 + xDiffSq / (2 * hp8 * hp7 * hp7);
    const double k3Val = hp6 * hp6 * pow(k3Pow1, -hp8);
    const double k4Val = hp9 * hp9 * exp(-xDiffSq / (2 * hp10 * hp10));
    const double indNoise = same ? hp11 * hp11 : 0.0;
    return k1Val + k2Val + k3Val + k4Val + indNoise;
}

// derAfterHyperParam(num), num 1-based (:69-137)
__device__ __forceinline__ double co2_der(double x1, double x2, bool same, int num, const Co2Params &p) {
    const double hp1 = p.hp[0], hp2 = p.hp[1], hp3 = p.hp[2], hp4 = p.hp[3], hp5 = p.hp[4], hp6 = p.hp[5], hp7 = p.hp[6], hp8 = p.hp[7],
                 hp9 = p.hp[8], hp10 = p.hp[9], hp11 = p.hp[10];
    const double xDiff = x1 - x2, sqDiff = (x1 - x2) * (x1 - x2);
    if (num < 3) {
        if (num == 1) return 2 * hp1 * exp(-sqDiff / (2 * hp2 * hp2));
        return hp1 * hp1 * exp(-sqDiff / (2 * hp2 * hp2)) * sqDiff * pow(hp2, -3.0);
    }
    if (num < 6) {
        const double sinVal = sin(M_PI * xDiff);
        const double k2Val = hp3 * hp3 * exp(-sqDiff / (2 * hp4 * hp4) - 2 * sinVal * sinVal / (hp5 * hp5));
        if (num == 3) return 2 * k2Val / hp3;
        if (num == 4) return k2Val * sqDiff * pow(hp4, -3.0);
        return k2Val * 4 * sinVal * sinVal * pow(hp5, -3.0);
    }
    if (num < 9) {
        const double k3Pow1 = 1 + sqDiff / (2 * hp8 * hp7 * hp7);
        if (num == 6) return 2 * hp6 * pow(k3Pow1, -hp8);
        if (num == 7) return hp6 * hp6 * pow(k3Pow1, -hp8 - 1) * sqDiff * pow(hp7, -3.0);
        const double firstTerm = exp(-hp8 * log(k3Pow1));
        const double secondTerm = -log(k3Pow1) + (hp8 * sqDiff / (2 * hp7 * hp7 * hp8 * hp8 * k3Pow1));
        return hp6 * hp6 * firstTerm * secondTerm;
    }
    const double k4Val = hp9 * hp9 * exp(-sqDiff / (2 * hp10 * hp10));
    if (num == 9) return 2 * k4Val / hp9;
    if (num == 10) return k4Val * sqDiff * pow(hp10, -3.0);
    return same ? 2 * hp11 : 0.0;
}

// pos = 0: the kernel; pos = 1..11: its derivative.  sym: rows and columns are the same points (noise flag i == j, i >= j evaluated,
// mirrored when `full`); otherwise the cross matrix (never adds noise, MatrixUtils.scala:44-55).
__global__ __launch_bounds__(256) void co2_gram_kernel(const double *__restrict__ xr, int nr, const double *__restrict__ xc, int nc, Co2Params prm,
                                                       int pos, double *__restrict__ K, int ldk, int sym, int full, double extra) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nr) return;
    const double xi = xr[i];
    for (int j = blockIdx.y; j < nc; j += gridDim.y) {
        if (sym && j > i) continue;
        const bool same = sym && (i == j);
        double v = pos ? co2_der(xi, xc[j], same, pos, prm) : co2_apply(xi, xc[j], same, prm);
        if (same && !pos) v += extra;
        K[i + (size_t)j * ldk] = v;
        if (sym && full && j < i) K[j + (size_t)i * ldk] = v;
    }
}

// out[0] = 0.5 * sum_ij (alpha_i alpha_j - Kinv_ij) D_ij over the full symmetric matrices, from their LOWER triangles
// (GpPredictor.scala:76: 0.5 * trace((alphaSq - inversedK) * derAfterKernelHyperParams)), fixed summation order.
__global__ __launch_bounds__(256) void co2_trace_kernel(int n, const double *__restrict__ alpha, const double *__restrict__ Kinv, int ldk,
                                                        const double *__restrict__ D, int ldd, double *__restrict__ partial) {
    __shared__ double red[256];
    const int j = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    const double aj = alpha[j];
    for (int i = j + tid; i < n; i += 256) {
        const double w = alpha[i] * aj - Kinv[i + (size_t)j * ldk];
        acc = fma((i == j) ? w : 2.0 * w, D[i + (size_t)j * ldd], acc);
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) partial[j] = red[0];
}
__global__ __launch_bounds__(256) void co2_trace_finish_kernel(int n, const double *__restrict__ partial, double *__restrict__ out) {
    __shared__ double red[256];
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int j = tid; j < n; j += 256) acc += partial[j];
    red[tid] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) out[0] = 0.5 * red[0];
}

Co2Params make_co2(const double *theta) {
    Co2Params p;
    for (int k = 0; k < 11; ++k) p.hp[k] = theta[k];
    return p;
}

}  // namespace

void gpk_co2_gram(hipStream_t s, const double *xr, int nr, const double *xc, int nc, const double *theta, int pos, double *K, int ldk, int sym,
                  int full, double extra) {
    if (nr <= 0 || nc <= 0) return;
    const int gy = nc < 1024 ? nc : 1024;
    hipLaunchKernelGGL(co2_gram_kernel, dim3((nr + 255) / 256, gy), dim3(256), 0, s, xr, nr, xc, nc, make_co2(theta), pos, K, ldk, sym, full, extra);
}

void gpk_co2_trace(hipStream_t s, int n, const double *alpha, const double *Kinv, int ldk, const double *D, int ldd, double *partial, double *out) {
    hipLaunchKernelGGL(co2_trace_kernel, dim3(n), dim3(256), 0, s, n, alpha, Kinv, ldk, D, ldd, partial);
    hipLaunchKernelGGL(co2_trace_finish_kernel, dim3(1), dim3(256), 0, s, n, partial, out);
}

// k(x, x, sameIndex = true): the diagonal of buildKernelMatrix(kernel, testData), for the variance
double gpk_co2_kss(const double *theta) {
    return theta[0] * theta[0] + theta[2] * theta[2] + theta[5] * theta[5] * std::pow(1.0, -theta[7]) + theta[8] * theta[8] + theta[10] * theta[10];
}
