// Tuning harness (not part of the product): store-pattern and compute variants of the cross-Gram tile kernel.
// build: hipcc --offload-arch=gfx950 -O3 -o gram_lab gram_lab.hip ; run: ./gram_lab [m] [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// V0: pure store, 64x64 tile, lane = row (8 B/lane, 512 B per wave store), column-major tile order (bi fastest)
__global__ __launch_bounds__(256) void v0(double *K, int ldk, int nbr) {
    const int bi = blockIdx.x % nbr, bj = blockIdx.x / nbr, ti = threadIdx.x & 63, tq = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 16; ++q) K[(bi * 64 + ti) + (size_t)(bj * 64 + tq + 4 * q) * ldk] = 1.5 + q;
}
// V1: pure store, MFMA layout (16 rows x 4 cols per wave store: 4 x 128 B)
__global__ __launch_bounds__(256) void v1(double *K, int ldk, int nbr) {
    const int bi = blockIdx.x % nbr, bj = blockIdx.x / nbr, lane = threadIdx.x & 63, w = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) K[(bi * 64 + 16 * it + fr) + (size_t)(bj * 64 + 16 * w + fk + 4 * r) * ldk] = 1.5 + r;
}
// V2: pure store, 16 B/lane: lane = row pair (32 lanes x 2 rows = 64 rows), 2 columns per wave store
__global__ __launch_bounds__(256) void v2(double *K, int ldk, int nbr) {
    const int bi = blockIdx.x % nbr, bj = blockIdx.x / nbr, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int rp = (lane & 31) * 2, ch = lane >> 5;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        d2 v = {1.5 + q, 2.5};
        *reinterpret_cast<d2 *>(K + (bi * 64 + rp) + (size_t)(bj * 64 + 16 * w + 2 * q + ch) * ldk) = v;
    }
}
// V3: 128-row x 32-col tile per WG: lane = row pair over 128 rows (64 lanes x 2), 16 B/lane, 1 KB contiguous per wave store
__global__ __launch_bounds__(256) void v3(double *K, int ldk, int nbr) {
    const int bi = blockIdx.x % nbr, bj = blockIdx.x / nbr, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        d2 v = {1.5 + q, 2.5};
        *reinterpret_cast<d2 *>(K + (bi * 128 + 2 * lane) + (size_t)(bj * 32 + 8 * w + q) * ldk) = v;
    }
}
// V4: like V0 plus 16 exp per lane (VALU load of the real kernel without the distance part)
__global__ __launch_bounds__(256) void v4(double *K, int ldk, int nbr, const double *x) {
    const int bi = blockIdx.x % nbr, bj = blockIdx.x / nbr, ti = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const double a = x[bi * 64 + ti];
#pragma unroll
    for (int q = 0; q < 16; ++q) K[(bi * 64 + ti) + (size_t)(bj * 64 + tq + 4 * q) * ldk] = 2.25 * exp(-0.5 * (a + q));
}
// V5: persistent-ish: each WG walks 4 consecutive tiles of a column (fewer, longer WGs), V0 pattern + exp
__global__ __launch_bounds__(256) void v5(double *K, int ldk, int nbr, const double *x) {
    const int bj = blockIdx.x / (nbr / 4), b0 = (blockIdx.x % (nbr / 4)) * 4, ti = threadIdx.x & 63, tq = threadIdx.x >> 6;
    for (int bi = b0; bi < b0 + 4; ++bi) {
        const double a = x[bi * 64 + ti];
#pragma unroll
        for (int q = 0; q < 16; ++q) K[(bi * 64 + ti) + (size_t)(bj * 64 + tq + 4 * q) * ldk] = 2.25 * exp(-0.5 * (a + q));
    }
}

struct GP { double sf2, sn2, extra; double inv_ls[64]; };
constexpr int GT = 64, ZS = 80, ZC = 16;
// VAR: 0 = full kernel, 1 = no exp, 2 = skip staging (operands constant), 3 = skip MFMA+norms (exp of a constant-ish)
template <int VAR>
__global__ __launch_bounds__(256) void gm(const double *__restrict__ Xr, int nr, int ldxr, const double *__restrict__ Xc, int nc, int ldxc, int d,
                                          GP prm, const double *__restrict__ center, int ldcen, double *__restrict__ K, int ldk, int nbr) {
    __shared__ __attribute__((aligned(16))) double zr[ZC * ZS], zc[ZC * ZS];
    __shared__ double nrm[2][GT];
    const int bi = blockIdx.x % nbr, bj = blockIdx.x / nbr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const int i0 = bi * GT, j0 = bj * GT;
    d4 acc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[it] = (d4){0.0, 0.0, 0.0, 0.0};
    double nacc = 0.0;
    for (int kc = 0; kc < d; kc += ZC) {
        if (kc) __syncthreads();
        if (VAR != 2) {
#pragma unroll
        for (int q = 0; q < ZC / 4; ++q) {
            const int kk = (tid >> 6) + 4 * q, k = kc + kk, pt = tid & 63;
            double vr = 0.0, vc = 0.0;
            if (k < d) {
                const double cen = center[(size_t)k * ldcen], il = prm.inv_ls[k];
                if (i0 + pt < nr) vr = (Xr[(i0 + pt) + (size_t)k * ldxr] - cen) * il;
                if (j0 + pt < nc) vc = (Xc[(j0 + pt) + (size_t)k * ldxc] - cen) * il;
            }
            zr[kk * ZS + pt] = vr;
            zc[kk * ZS + pt] = vc;
        }
        }
        __syncthreads();
        if (VAR != 3) {
        if (tid < 2 * GT) {
            const double *z = (tid < GT) ? zr : zc;
            const int pt = tid & 63;
#pragma unroll
            for (int kk = 0; kk < ZC; ++kk) nacc = fma(z[kk * ZS + pt], z[kk * ZS + pt], nacc);
        }
        const int ksteps = (min(d - kc, ZC) + 3) / 4;
        for (int ks = 0; ks < ksteps; ++ks) {
            const double a = zc[(4 * ks + fk) * ZS + 16 * wave + fr];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const double b = zr[(4 * ks + fk) * ZS + 16 * it + fr];
                acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[it], 0, 0, 0);
            }
        }
        }
    }
    if (tid < 2 * GT) nrm[tid >> 6][tid & 63] = nacc;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int il = 16 * it + fr, gi = i0 + il;
        const double ni = nrm[0][il];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jl = 16 * wave + fk + 4 * r, gj = j0 + jl;
            const double r2 = fmax(fma(-2.0, acc[it][r], ni + nrm[1][jl]), 0.0);
            double v = (VAR == 1) ? r2 : prm.sf2 * exp(-0.5 * r2);
            if (gi < nr && gj < nc) K[gi + (size_t)gj * ldk] = v;
        }
    }
}

// gm2: d <= 16 single chunk; norms folded into the staging threads; interior tiles take a path without bounds checks;
// pointer arithmetic hoisted; LDS dots -> lane = row epilogue (512 B contiguous per wave store)
template <int EPI>
__global__ __launch_bounds__(256) void gm2(const double *__restrict__ Xr, int nr, int ldxr, const double *__restrict__ Xc, int nc, int ldxc, int d,
                                           GP prm, const double *__restrict__ center, int ldcen, double *__restrict__ K, int ldk, int nbr) {
    __shared__ __attribute__((aligned(16))) double zr[ZC * ZS], zc[ZC * ZS];
    __shared__ double nrm[2][GT];
    __shared__ double dots[EPI == 1 ? GT * (GT + 1) : 1];
    const int bi = blockIdx.x % nbr, bj = blockIdx.x / nbr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const int i0 = bi * GT, j0 = bj * GT;
    {
        const int pt = tid & 63, kq = tid >> 6;
#pragma unroll
        for (int q = 0; q < ZC / 4; ++q) {
            const int k = kq + 4 * q;
            double vr = 0.0, vc = 0.0;
            if (k < d) {
                const double cen = center[(size_t)k * ldcen], il = prm.inv_ls[k];
                vr = (Xr[min(i0 + pt, nr - 1) + (size_t)k * ldxr] - cen) * il;
                vc = (Xc[min(j0 + pt, nc - 1) + (size_t)k * ldxc] - cen) * il;
            }
            zr[k * ZS + pt] = vr;
            zc[k * ZS + pt] = vc;
        }
    }
    __syncthreads();
    if (tid < 2 * GT) {
        const double *z = (tid < GT) ? zr : zc;
        double nacc = 0.0;
#pragma unroll
        for (int kk = 0; kk < ZC; ++kk) nacc = fma(z[kk * ZS + lane], z[kk * ZS + lane], nacc);
        nrm[tid >> 6][lane] = nacc;
    }
    d4 acc[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) acc[it] = (d4){0.0, 0.0, 0.0, 0.0};
    const int ksteps = (d + 3) / 4;
    for (int ks = 0; ks < ksteps; ++ks) {
        const double a = zc[(4 * ks + fk) * ZS + 16 * wave + fr];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const double b = zr[(4 * ks + fk) * ZS + 16 * it + fr];
            acc[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[it], 0, 0, 0);
        }
    }
    const bool interior = (i0 + GT <= nr) && (j0 + GT <= nc);
    if (EPI == 0) {
        __syncthreads();
        double *Kp = K + (i0 + fr) + (size_t)(j0 + 16 * wave + fk) * ldk;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const double ni = nrm[0][16 * it + fr];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double r2 = fmax(fma(-2.0, acc[it][r], ni + nrm[1][16 * wave + fk + 4 * r]), 0.0);
                const double v = prm.sf2 * exp(-0.5 * r2);
                if (interior || (i0 + 16 * it + fr < nr && j0 + 16 * wave + fk + 4 * r < nc)) Kp[16 * it + (size_t)(4 * r) * ldk] = v;
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) dots[(16 * wave + fk + 4 * r) * (GT + 1) + 16 * it + fr] = acc[it][r];
        __syncthreads();
        const int ti = lane, tq = wave;
        const double ni = nrm[0][ti];
        double *Kp = K + (i0 + ti) + (size_t)(j0 + tq) * ldk;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int jl = tq + 4 * q;
            const double r2 = fmax(fma(-2.0, dots[jl * (GT + 1) + ti], ni + nrm[1][jl]), 0.0);
            const double v = prm.sf2 * exp(-0.5 * r2);
            if (interior || (i0 + ti < nr && j0 + jl < nc)) Kp[(size_t)(4 * q) * ldk] = v;
        }
    }
}
int main(int argc, char **argv) {
    const int m = argc > 1 ? atoi(argv[1]) : 65536, n = argc > 2 ? atoi(argv[2]) : 8192;
    double *K, *x;
    CK(hipMalloc(&K, sizeof(double) * (size_t)m * n));
    CK(hipMalloc(&x, sizeof(double) * m));
    CK(hipMemset(x, 0, sizeof(double) * m));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gb = 8.0 * m * n / 1e9;
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-46s %8.1f us  %.2f TB/s\n", name, ms * 1e3, gb / ms);
    };
    const int nbr = m / 64, nbc = n / 64;
    timeit("memset", [&] { CK(hipMemsetAsync(K, 0, sizeof(double) * (size_t)m * n, 0)); });
    timeit("v0 store 64x64 lane=row 8B", [&] { hipLaunchKernelGGL(v0, dim3(nbr * nbc), dim3(256), 0, 0, K, m, nbr); });
    timeit("v1 store 64x64 mfma layout 4x128B", [&] { hipLaunchKernelGGL(v1, dim3(nbr * nbc), dim3(256), 0, 0, K, m, nbr); });
    timeit("v2 store 64x64 16B/lane", [&] { hipLaunchKernelGGL(v2, dim3(nbr * nbc), dim3(256), 0, 0, K, m, nbr); });
    timeit("v3 store 128x32 16B/lane 1KB/wave", [&] { hipLaunchKernelGGL(v3, dim3((m / 128) * (n / 32)), dim3(256), 0, 0, K, m, m / 128); });
    timeit("v4 v0 + 16 exp/lane", [&] { hipLaunchKernelGGL(v4, dim3(nbr * nbc), dim3(256), 0, 0, K, m, nbr, x); });
    timeit("v5 v4, 4 tiles per WG", [&] { hipLaunchKernelGGL(v5, dim3(nbr * nbc / 4), dim3(256), 0, 0, K, m, nbr, x); });
    const int d = 8;
    double *X, *Xs;
    CK(hipMalloc(&X, sizeof(double) * (size_t)n * d)); CK(hipMalloc(&Xs, sizeof(double) * (size_t)m * d));
    std::vector<double> h((size_t)m * d);
    for (size_t i = 0; i < h.size(); ++i) h[i] = -2.0 + 4.0 * ((i * 2654435761u) % 100003) / 100003.0;
    CK(hipMemcpy(Xs, h.data(), sizeof(double) * (size_t)m * d, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, h.data(), sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice));
    GP prm; prm.sf2 = 2.25; prm.sn2 = 0.01; prm.extra = 0; for (int k = 0; k < 64; ++k) prm.inv_ls[k] = 1.0 / (1.0 + 0.25 * (k % 8));
    timeit("gm<0> full mfma gram", [&] { hipLaunchKernelGGL(gm<0>, dim3(nbr * nbc), dim3(256), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr); });
    timeit("gm2<0> mfma-layout epilogue, fast path", [&] { hipLaunchKernelGGL(gm2<0>, dim3(nbr * nbc), dim3(256), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr); });
    timeit("gm2<1> LDS dots, lane=row epilogue", [&] { hipLaunchKernelGGL(gm2<1>, dim3(nbr * nbc), dim3(256), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr); });
    timeit("gm<1> no exp", [&] { hipLaunchKernelGGL(gm<1>, dim3(nbr * nbc), dim3(256), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr); });
    timeit("gm<2> no staging loads", [&] { hipLaunchKernelGGL(gm<2>, dim3(nbr * nbc), dim3(256), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr); });
    timeit("gm<3> no mfma/norm", [&] { hipLaunchKernelGGL(gm<3>, dim3(nbr * nbc), dim3(256), 0, 0, Xs, m, m, X, n, n, d, prm, X, n, K, m, nbr); });
    return 0;
}
