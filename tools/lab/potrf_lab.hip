// Phase timing of potrf_diag128_kernel: the library's kernel source compiled with -DPOTRF_STAMPS, one 128 x 128 SPD block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPOTRF_STAMPS -I include -I gp_algos_amd/csrc tools/lab/potrf_lab.hip -o tools/lab/potrf_lab
#include "../../gp_algos_amd/csrc/kernels_diag.hip"
#include <cstdio>
#include <vector>

int main() {
    const int n = 128;
    std::vector<double> A((size_t)n * n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) A[i + (size_t)j * n] = (i == j ? 4.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
    double *dA, *dinv;
    int *dinfo;
    hipMalloc(&dA, sizeof(double) * n * n);
    hipMalloc(&dinv, sizeof(double) * n * 16);
    hipMalloc(&dinfo, sizeof(int) * 4);
    hipMemset(dinfo, 0, sizeof(int) * 4);
    if (gpk_init_diag_kernels() != 0) { printf("init failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        gpk_potrf_diag128(0, dA, n, dinv, dinfo, 0, gp_batch());
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    unsigned long long st[256];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(potrf_stamps), sizeof(st));
    const double total = (double)(st[3] - st[0]);
    printf("event time %.2f us; stamped span %.0f ticks (%.3f us per tick if the span is the whole launch)\n", best * 1e3, total, best * 1e3 / total);
    printf("load %.0f  loop %.0f  store %.0f\n", (double)(st[1] - st[0]), (double)(st[2] - st[1]), (double)(st[3] - st[2]));
    printf("jb:  chol   inv+st  bar1   panel  bar2   upd(w0)  | loop iteration   wave1 end-of-iteration lag\n");
    for (int jb = 0; jb < 8; ++jb) {
        const unsigned long long *q = st + 8 + 8 * jb;
        const unsigned long long next = jb < 7 ? st[8 + 8 * (jb + 1)] : st[2];
        printf("%d: %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f   | %6.0f   %6.0f\n", jb, (double)(q[1] - q[0]), (double)(q[2] - q[1]), (double)(q[3] - q[2]),
               (double)(q[4] - q[3]), (double)(q[5] - q[4]), (double)(q[6] - q[5]), (double)(next - q[0]), (double)st[128 + jb] - (double)q[6]);
    }
    int info = 0;
    hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost);
    printf("info %d\n", info);
    return 0;
}
