// fp64 MFMA GEMM for gfx950 (MI355X):  C = beta*C + alpha * A * B^T   ("NT", everything column-major).
//
// This one kernel carries every O(n^3) term of the reference's hot path:
//   - Cholesky trailing update  A22 -= P P^T            (breeze.linalg.cholesky, GpPredictor.scala:120)
//   - posterior solve           Vt_i -= Vt_{<i} L_i,<i^T  (forwardSolve(L, K*^T), GpPredictor.scala:34)
//   - full predictive covariance Gram(X*) - Vt Vt^T      (GpPredictor.scala:36)
//   - K^-1 = L^-T L^-1 for the LML gradient              (GpPredictor.scala:66-67)
//   - EP  Sigma = K - V^T V                               (EpParameterEstimator.scala:60)
//
// Design (CDNA4): 128x128 output tile per 256-thread workgroup (4 waves, 2x2, 64x64 per wave =
// 4x4 v_mfma_f64_16x16x4_f64 accumulators = 128 acc VGPRs), BK = 16, LDS double-buffered and
// register-staged.  Both operands are "row-contiguous" in memory (column-major, NT form), so one
// wave-load of 64 x 16 B covers a full 1 KiB tile column: perfectly coalesced.  LDS image is
// [k][row] with a row stride of 144 doubles (1152 B = 128 mod 256 B), which makes the
// ds_read_b64 fragment reads (lanes 0-15: 16 rows at k, lanes 16-31: the same rows at k+1)
// bank-conflict free.  The MFMA is issued with the B-fragment as srcA and the A-fragment as srcB
// so each lane's accumulator holds 4 consecutive-n values at one m: stores are 128-B row segments
// of column-major C.
#include "gpcore_internal.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {

constexpr int TM = 128, TN = 128, TK = GP_BK;
constexpr int LDS_STRIDE = TM + 16;  // doubles

__device__ __forceinline__ void tile_coords_lower(int t, int nbm, int nbn, int &bi, int &bj) {
    // Lower trapezoid of an (nbm x nbn) tile grid, nbm >= nbn, enumerated in SUPER-COLUMNS of G = 4 tile columns:
    // inside a super-column tiles run row by row (bi ascending, its <= 4 columns innermost).  Tiles that are
    // consecutive in this order share their A row panel and cycle through the same 4 B panels, so the working set
    // of one XCD (4 B panels + a streaming A panel, ~2.6 MB at K = 512) stays in its 4 MB L2 instead of
    // re-fetching 2 x 0.5 MB of panels per tile.
    constexpr int G = 4;
    for (int c0 = 0; c0 < nbn; c0 += G) {
        const int w = (nbn - c0 < G) ? nbn - c0 : G;
        const int head = w * (w + 1) / 2;                  // rows c0 .. c0+w-1 hold 1 .. w tiles
        const int total = head + (nbm - c0 - w) * w;       // then full rows of w tiles
        if (t < total) {
            int r, c;
            if (t < head) {
                r = 0;
                while ((r + 1) * (r + 2) / 2 <= t) ++r;
                c = t - r * (r + 1) / 2;
            } else {
                const int u = t - head;
                r = w + u / w;
                c = u % w;
            }
            bi = c0 + r;
            bj = c0 + c;
            return;
        }
        t -= total;
    }
    bi = bj = 0;   // not reached for a valid launch
}

template <int LOWER, int HAS_BETA>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(int M, int N, int K, double alpha, const double *__restrict__ A, int lda,
                                                           const double *__restrict__ B, int ldb, double beta,
                                                           double *__restrict__ C, int ldc, int ktri) {
    __shared__ __attribute__((aligned(16))) double smem[2 * 2 * TK * LDS_STRIDE];
    double *As = smem;                          // [2][TK][LDS_STRIDE]
    double *Bs = smem + 2 * TK * LDS_STRIDE;    // [2][TK][LDS_STRIDE]

    int bi, bj;
    if (LOWER) {
        // XCD-aware: block ids are dealt round-robin over the 8 XCDs; hand each XCD a contiguous run of the order above
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        tile_coords_lower(swz, M / TM, N / TN, bi, bj);
    } else {
        // XCD-aware: consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous
        // run of tiles that share the same B panel (bj) so the panel stays in that XCD's L2.
        int nbm = M / TM;
        int nwg = gridDim.x;
        int id = blockIdx.x;
        int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        bi = swz % nbm;
        bj = swz / nbm;
    }
    const int row0 = bi * TM, col0 = bj * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;

    // ktri: A(i,k) == 0 for k < i (e.g. L^-T): the product over k can start at this tile's row block
    const int kbeg = ktri ? row0 : 0;
    const double *Ag = A + row0 + (size_t)kbeg * lda;
    const double *Bg = B + col0 + (size_t)kbeg * ldb;

    // staging map: 4 x (16-byte) loads per thread per operand.  e = tid + 256*q: row pair e&63, k = e>>6
    const int ld_r = (tid & 63) * 2;
    const int ld_k = tid >> 6;  // + 4*q

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double2_t ra[4], rb[4];
    const int KT = (K - kbeg) / TK;

    if (KT > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *reinterpret_cast<const double2_t *>(Ag + ld_r + (size_t)(ld_k + 4 * q) * lda);
            rb[q] = *reinterpret_cast<const double2_t *>(Bg + ld_r + (size_t)(ld_k + 4 * q) * ldb);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<double2_t *>(As + (ld_k + 4 * q) * LDS_STRIDE + ld_r) = ra[q];
            *reinterpret_cast<double2_t *>(Bs + (ld_k + 4 * q) * LDS_STRIDE + ld_r) = rb[q];
        }
    }
    __syncthreads();

    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) {
            const size_t koff = (size_t)(kt + 1) * TK;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ra[q] = *reinterpret_cast<const double2_t *>(Ag + ld_r + (koff + ld_k + 4 * q) * lda);
                rb[q] = *reinterpret_cast<const double2_t *>(Bg + ld_r + (koff + ld_k + 4 * q) * ldb);
            }
        }
        const double *Ac = As + cur * TK * LDS_STRIDE + wm + fr;
        const double *Bc = Bs + cur * TK * LDS_STRIDE + wn + fr;
#pragma unroll
        for (int ks = 0; ks < TK / 4; ++ks) {
            double af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = Ac[(ks * 4 + fk) * LDS_STRIDE + t * 16];
                bf[t] = Bc[(ks * 4 + fk) * LDS_STRIDE + t * 16];
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        if (kt + 1 < KT) {
            const int nxt = cur ^ 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                *reinterpret_cast<double2_t *>(As + nxt * TK * LDS_STRIDE + (ld_k + 4 * q) * LDS_STRIDE + ld_r) = ra[q];
                *reinterpret_cast<double2_t *>(Bs + nxt * TK * LDS_STRIDE + (ld_k + 4 * q) * LDS_STRIDE + ld_r) = rb[q];
            }
        }
        __syncthreads();
    }

    // epilogue: acc[nt][mt][r] = D[n = wn + nt*16 + fk + 4r][m = wm + mt*16 + fr].
    // C is read in batches of 16 independent loads per 16-column group, then written: a load/wait/store chain per
    // element (what a naive `v += beta * *cp` compiles to) serialises 64 memory round trips per tile.
    const bool diag_tile = LOWER && (bi == bj);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        double cv[4][4];
        if (HAS_BETA) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = col0 + wn + nt * 16 + fk + 4 * r;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int m = row0 + wm + mt * 16 + fr;
                    cv[r][mt] = (diag_tile && m < n) ? 0.0 : C[m + (size_t)n * ldc];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = col0 + wn + nt * 16 + fk + 4 * r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = row0 + wm + mt * 16 + fr;
                double v = alpha * acc[nt][mt][r];
                if (HAS_BETA) v = fma(beta, cv[r][mt], v);
                if (!(diag_tile && m < n)) C[m + (size_t)n * ldc] = v;
            }
        }
    }
}

// register-only MFMA loop: measures the achievable fp64 matrix-core rate (roofline denominator) with 16
// independent accumulators per wave, and stamps shader-clock / real-time counters around the loop so the
// clock the chip holds under fp64 MFMA load and the cycles per v_mfma_f64_16x16x4_f64 can be read off.
__global__ __launch_bounds__(256) void mfma_probe_kernel(double *out, unsigned long long *stamps, int iters) {
    double4_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        // inline asm with "+v": keeps the accumulators in VGPRs (with the builtin hipcc parks them in AGPRs and
        // copies all 128 registers back and forth every iteration, which is what a first version of this probe timed)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = c1 - c0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

}  // namespace

void gpk_gemm_nt(hipStream_t s, int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb,
                 double beta, double *C, int ldc, int lower, int ktri) {
    if (M <= 0 || N <= 0) return;
    const bool hb = beta != 0.0;
    if (lower) {
        int nbm = M / TM, nbn = N / TN;   // trapezoid: M >= N
        int ntiles = nbn * nbm - nbn * (nbn - 1) / 2;
        if (hb) hipLaunchKernelGGL((gemm_nt_f64_kernel<1, 1>), dim3(ntiles), dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ktri);
        else hipLaunchKernelGGL((gemm_nt_f64_kernel<1, 0>), dim3(ntiles), dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ktri);
    } else {
        int ntiles = (M / TM) * (N / TN);
        if (hb) hipLaunchKernelGGL((gemm_nt_f64_kernel<0, 1>), dim3(ntiles), dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ktri);
        else hipLaunchKernelGGL((gemm_nt_f64_kernel<0, 0>), dim3(ntiles), dim3(256), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ktri);
    }
}

double gpk_probe_mfma(hipStream_t s, int num_cu, int waves_per_simd, double *clock_mhz, double *cycles_per_mfma) {
    const int iters = 4000;
    const int blocks = num_cu * waves_per_simd;   // 4 waves per block = one per SIMD
    double *d = nullptr;
    unsigned long long *st = nullptr;
    if (hipMalloc(&d, sizeof(double) * blocks * 256) != hipSuccess) return -1.0;
    if (hipMalloc(&st, sizeof(unsigned long long) * blocks * 8) != hipSuccess) { (void)hipFree(d); return -1.0; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, s, d, st, 200);  // warm-up
    (void)hipEventRecord(e0, s);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, s, d, st, iters);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
    double cyc = 0.0, rt = 0.0;
    for (int w = 0; w < blocks * 4; ++w) { cyc += (double)h[2 * w]; rt += (double)h[2 * w + 1]; }
    if (clock_mhz) *clock_mhz = rt > 0 ? cyc / rt * 100.0 : 0.0;               // s_memrealtime ticks at 100 MHz
    if (cycles_per_mfma) *cycles_per_mfma = cyc / (blocks * 4.0) / ((double)iters * 16.0) / waves_per_simd;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    (void)hipFree(st);
    double flops = (double)blocks * 4 /*waves*/ * (double)iters * 16 * (2.0 * 16 * 16 * 4);
    return flops / (ms * 1e-3) / 1e12;
}
