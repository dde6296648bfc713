// Lab for the short-K GEMM (round 4): the library's own kernels (kernels_gemm.hip is included as it stands) on the launch shapes
// of the blocked Cholesky / T = L^-T / EP updates, single problems and lockstep batches; every variant is checked bit for bit
// against gemm_nt_f64_kernel before it is timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/lab/gemm_lab tools/lab/gemm_lab.hip
//   tools/lab/gemm_lab            (prints one line per shape and variant)
#include "../../gp_algos_amd/csrc/kernels_gemm.hip"
#include <chrono>
#include <string>

// stubs for the few launchers kernels_gemm.hip declares but other translation units define: none are called here
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(double *p, size_t n, unsigned long long seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 0.125;
    }
}
__global__ void diff_kernel(const double *a, const double *b, size_t n, unsigned long long *ndiff) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) {
        const unsigned long long x = __double_as_longlong(a[i]), y = __double_as_longlong(b[i]);
        c += (x != y);
    }
    if (c) atomicAdd(ndiff, c);
}

struct shape { const char *name; int n, K0, K, extra, lower, count; int wcols; };   // wcols > 0: rectangular in-panel / right update of `wcols` columns

int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int ncu = pr.multiProcessorCount;
    printf("device %s, %d CUs\n", pr.gcnArchName, ncu);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::vector<shape> shapes = {
        {"chol outer K=512  n=8192 first (r=7680)", 8192, 0, 512, 128, 1, 1, 0},
        {"chol outer K=512  n=8192 r=5632", 8192, 2048, 512, 128, 1, 1, 0},
        {"chol outer K=512  n=8192 r=3584", 8192, 4096, 512, 128, 1, 1, 0},
        {"chol outer K=512  n=8192 r=2048", 8192, 5632, 512, 128, 1, 1, 0},
        {"chol outer K=512  n=4096 r=3584 single", 4096, 0, 512, 128, 1, 1, 0},
        {"chol outer K=512  n=4096 r=3584 x32", 4096, 0, 512, 128, 1, 32, 0},
        {"chol outer K=512  n=4096 r=2048 x32", 4096, 1536, 512, 128, 1, 32, 0},
        {"in-panel  K=128  n=4096 r=3968 w=384 x32", 4096, 0, 128, 128, 1, 32, 384},
        {"in-panel  K=128  n=4096 r=2432 w=384 x32", 4096, 1536, 128, 128, 1, 32, 384},
        {"EP Sigma  K=256  n=4096 lower single", 4096, 0, 256, 0, 1, 1, 0},
        {"EP Sigma  K=1024 n=4096 lower x12", 4096, 0, 1024, 0, 1, 12, 0},
        {"rect      K=512  M=2048 N=2048 x32 (T update)", 4096, 1536, 512, 0, 0, 32, 2048},
        {"rect      K=128  M=2048 N=384 x32 (T in-panel)", 4096, 1536, 128, 0, 0, 32, 384},
    };
    for (const shape &sh : shapes) {
        const int lda = sh.n + 128;
        const size_t per = (size_t)lda * sh.n;                  // one matrix
        const size_t total = per * sh.count;
        double *m0, *m1, *init;
        CK(hipMalloc(&m0, total * 8));
        CK(hipMalloc(&m1, total * 8));
        CK(hipMalloc(&init, total * 8));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s, init, total, 12345ull);
        const int c1 = sh.K0 + sh.K;
        int M, N;
        const double *A0, *B0;
        double *C0;
        auto ptrs = [&](double *base) {
            if (sh.lower) {
                M = sh.n + sh.extra - c1;
                N = sh.wcols ? sh.wcols : sh.n - c1;
                A0 = base + c1 + (size_t)sh.K0 * lda;
                B0 = A0;
                C0 = base + c1 + (size_t)c1 * lda;
            } else {   // rectangular: C[0:M, c1:c1+N] -= T[0:M, K0:c1] * L[c1:c1+N, K0:c1]^T, both operands inside the same buffer
                M = sh.wcols == 2048 ? 2048 : 2048;
                N = sh.wcols;
                A0 = base + (size_t)sh.K0 * lda;
                B0 = base + c1 + (size_t)sh.K0 * lda;
                C0 = base + (size_t)c1 * lda + (sh.n / 2);   // rows n/2.. : disjoint from the A rows used (0..M) only if M <= n/2
            }
        };
        gp_batch bt;
        bt.count = sh.count;
        bt.s0 = bt.s1 = bt.s2 = per;
        double flops;
        for (int variant = 0; variant < 2; ++variant) {
            double *buf = variant == 0 ? m0 : m1;
            CK(hipMemcpyAsync(buf, init, total * 8, hipMemcpyDeviceToDevice, s));
            ptrs(buf);
            const double ntile = sh.lower ? ((double)(N / 128) * (M / 128) - (double)(N / 128) * ((N / 128) - 1) / 2.0) : (double)(M / 128) * (N / 128);
            flops = ntile * sh.count * 2.0 * 128 * 128 * sh.K;
            auto launch = [&]() {
                if (variant == 0) gpk_gemm_nt(s, M, N, sh.K, -1.0, A0, lda, B0, lda, 1.0, C0, lda, sh.lower, 0, bt);
                else gpk_gemm_nt_pt(s, M, N, sh.K, -1.0, A0, lda, B0, lda, 1.0, C0, lda, sh.lower, bt, nullptr, 0, ncu);
            };
            launch();                                   // the checked launch
            CK(hipStreamSynchronize(s));
            if (variant == 1) {
                unsigned long long *nd, h = 0;
                CK(hipMalloc(&nd, 8));
                CK(hipMemset(nd, 0, 8));
                hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, s, m0, m1, total, nd);
                CK(hipMemcpy(&h, nd, 8, hipMemcpyDeviceToHost));
                CK(hipFree(nd));
                printf("    bitwise differences vs gemm_nt_f64_kernel: %llu of %zu\n", h, total);
            }
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            // in the timed loop the operands keep their values (C drifts: harmless)
            CK(hipEventRecord(e0, s));
            for (int it = 0; it < iters; ++it) launch();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipGetLastError());
            printf("%-52s %-22s M=%5d N=%5d K=%4d tiles=%7.0f  %9.1f us  %6.2f TFLOP/s\n", sh.name, variant == 0 ? "gemm_nt_f64 (2 WG/CU)" : "gemm_pt (persistent)",
                   M, N, sh.K, ntile * sh.count, ms / iters * 1e3, flops / (ms / iters * 1e-3) / 1e12);
            fflush(stdout);
            CK(hipEventDestroy(e0));
            CK(hipEventDestroy(e1));
            if (variant == 0) {   // restore for the bitwise comparison: rerun the single checked launch from the initial state
                CK(hipMemcpyAsync(buf, init, total * 8, hipMemcpyDeviceToDevice, s));
                launch();
                CK(hipStreamSynchronize(s));
            }
        }
        CK(hipFree(m0));
        CK(hipFree(m1));
        CK(hipFree(init));
    }
    return 0;
}
