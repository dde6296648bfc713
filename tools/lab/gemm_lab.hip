// Lab for the short-K GEMM (round 4): the library's gemm_nt_f64_kernel (kernels_gemm.hip is included as it stands) against a
// PERSISTENT-TILE form of the same product on the launch shapes of the blocked Cholesky / T = L^-T / EP updates, single problems and
// lockstep batches; the variant is checked bit for bit against gemm_nt_f64_kernel before it is timed.
// Result (profiles/r04_a_gemm_lab.log): bit-identical everywhere and SLOWER or equal everywhere -- K = 512 in a lockstep batch 60.3
// against 64.5 TFLOP/s (0.77 against 0.82 of the fp64 matrix peak), single problems 46-52 against 50-52, K = 128 equal (41-45: those
// launches are bound by the read-modify-write of C, 16 flop per HBM byte).  Two 8-wave workgroups per CU already overlap one tile's
// epilogue with the other's k loop; one workgroup per CU with 2 waves per SIMD pays more in its k loop (2.22 us per 16-deep k-step
// against 1.71 at the pipe rate) than the deferred epilogue returns.  The kernel therefore lives HERE, not in the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/lab/gemm_lab tools/lab/gemm_lab.hip
//   tools/lab/gemm_lab [iters]
#include "../../gp_algos_amd/csrc/kernels_gemm.hip"
#include <chrono>
#include <string>

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// Persistent-tile form of the same product for SHORT K (round 4):  C = beta*Cin + alpha * A * B^T, K = 128 ... 1024.
//
// With K = 128 ... 512 a 128 x 128 tile is 14 ... 55 us of MFMA issue between a prologue (first operand stage: one L2 round trip)
// and an epilogue (read-modify-write of the 128 KB tile of C: two dependent HBM round trips) that keep the matrix pipe idle --
// gemm_nt_f64_kernel hides them only as far as the OTHER workgroup of the CU happens to be in its k loop.  Here ONE workgroup per
// CU (8 waves x 64 x 32, 2 waves per SIMD, up to 256 registers each) walks a list of tiles and never leaves the k loop between them:
//   * the last k-step of tile t already stages the first operand tile of tile t+1 (the LDS ring just keeps turning);
//   * tile t's finished accumulators move to a second register set ("held") and its epilogue is TRICKLED through the first five
//     k-steps of tile t+1: k-step j asks for a quarter of the C tile (8 loads per lane), k-step j+1 combines and stores it.  The
//     loads have a whole k-step (>= 1.7 us) to come back and are waited for with a COUNTED s_waitcnt (vector-memory operations
//     complete in issue order: stores of the previous quarter, then this k-step's LDS-DMA, then the loads -- vmcnt(8) retires
//     everything but the loads) in front of a raw s_barrier, so neither the DMA nor the barrier ever waits for HBM;
//   * only the very last tile of a workgroup has a conventional epilogue.
// Tile order and XCD mapping: the tiles of the launch (all problems of a lockstep batch) are cut into 8 contiguous runs, one per
// XCD; the W workgroups of an XCD take its run round-robin (tiles t, t+W, ...), so at any moment the XCD works on W consecutive
// tiles, which share their operand panels in that XCD's L2.  Results are bit-identical to gemm_nt_f64_kernel's (same k order per
// element, same fma in the epilogue).
template <int LOWER, int HAS_BETA>
__global__ __launch_bounds__(512, 1) void gemm_pt_kernel(int M, int N, int K, double alpha, const double *__restrict__ A, int lda,
                                                         const double *__restrict__ B, int ldb, double beta, double *__restrict__ C, int ldc,
                                                         gp_batch bt, const double *__restrict__ Cin, int ldcin, int tiles_per_problem, int total_tiles) {
    __shared__ __attribute__((aligned(16))) double smem[2 * 2 * TK * LDS_STRIDE];
    double *As = smem;                          // [2][TK][LDS_STRIDE]
    double *Bs = smem + 2 * TK * LDS_STRIDE;    // [2][TK][LDS_STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
    const int fr = lane & 15, fk = lane >> 4;
    // this workgroup's tiles: run [r0, r1) of its XCD, entries r0 + w, r0 + w + W, ...
    const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
    const int W = ((int)gridDim.x + 7 - xcd) >> 3;
    const int q = total_tiles >> 3, rm = total_tiles & 7;
    const int r0 = xcd * q + (xcd < rm ? xcd : rm), r1 = r0 + q + (xcd < rm ? 1 : 0);
    int t = r0 + w;
    if (t >= r1) return;
    const int nbm = M / TM, nbn = N / TN, KT = K / TK;
    if (!Cin) { Cin = C; ldcin = ldc; bt.s3 = bt.s2; }

    // a tile, as this lane sees it: LDS-DMA sources of k-row `wave`, its first element of C / Cin (m = bi*128 + wm + fr, n = bj*128 + wn + fk),
    // and dbias = 0 on a diagonal tile of a lower product (elements with m < n are neither read nor written), a large number elsewhere
    const double *cur_a, *cur_b, *nxt_a = nullptr, *nxt_b = nullptr, *hd_ci = nullptr, *cur_ci, *nxt_ci = nullptr;
    double *cur_c, *nxt_c = nullptr, *hd_c = nullptr;
    int cur_db, nxt_db = 0;
    const int dm = wm + fr - wn - fk;     // element (mt, dn) of this lane lies above the diagonal of a diagonal tile iff dm + 16 mt < dn
    auto locate = [&](int g, const double *&pa, const double *&pb, double *&pc, const double *&pci, int &db) {
        const int prob = g / tiles_per_problem, tt = g - prob * tiles_per_problem;
        int bi, bj;
        if (LOWER) tile_coords_lower(tt, nbm, nbn, bi, bj);
        else { bi = tt % nbm; bj = tt / nbm; }
        pa = A + (size_t)prob * bt.s0 + (size_t)bi * TM + lane * 2 + (size_t)wave * lda;
        pb = B + (size_t)prob * bt.s1 + (size_t)bj * TN + lane * 2 + (size_t)wave * ldb;
        pc = C + (size_t)prob * bt.s2 + (size_t)(bi * TM + wm + fr) + (size_t)(bj * TN + wn + fk) * ldc;
        pci = Cin + (size_t)prob * bt.s3 + (size_t)(bi * TM + wm + fr) + (size_t)(bj * TN + wn + fk) * ldcin;
        db = (LOWER && bi == bj) ? 0 : (1 << 20);
    };
    auto stage = [&](int buf, const double *pa, const double *pb, int kt) {
        const size_t koff = (size_t)kt * TK;
#pragma unroll
        for (int qq = 0; qq < TK / 8; ++qq) {
            __builtin_amdgcn_global_load_lds(pa + (koff + 8 * qq) * lda, As + (buf * TK + wave + 8 * qq) * LDS_STRIDE, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(pb + (koff + 8 * qq) * ldb, Bs + (buf * TK + wave + 8 * qq) * LDS_STRIDE, 16, 0, 0);
        }
    };
    int buf = 0;
    double4_t acc[2][4], held[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0}; held[a][b] = acc[a][b]; }
    auto mfma_block = [&](int buf) {
        const double *Ac = As + buf * TK * LDS_STRIDE + wm + fr;
        const double *Bc = Bs + buf * TK * LDS_STRIDE + wn + fr;
#pragma unroll
        for (int ks = 0; ks < TK / 4; ++ks) {
            double af[4], bf[2];
#pragma unroll
            for (int u = 0; u < 4; ++u) af[u] = Ac[(ks * 4 + fk) * LDS_STRIDE + u * 16];
#pragma unroll
            for (int u = 0; u < 2; ++u) bf[u] = Bc[(ks * 4 + fk) * LDS_STRIDE + u * 16];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
    };
    // quarter g of the held tile: nt = g >> 1, r in {2 (g & 1), 2 (g & 1) + 1}, mt = 0..3  ->  element (m0 + 16 mt, n0 + 16 nt + 4 r).
    // Only tiles OFF the diagonal are held (no element of theirs is masked: loads and stores are unconditional, no branch around
    // a vector-memory instruction); a diagonal tile of a lower product gets a conventional, masked epilogue at once.
    double cv[8];
    auto quarter_load = [&](int g) {
        if (!HAS_BETA) return;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int nt = g >> 1, r = 2 * (g & 1) + rr, dn = nt * 16 + 4 * r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) cv[rr * 4 + mt] = hd_ci[mt * 16 + (size_t)dn * ldcin];
        }
    };
    auto quarter_store = [&](int g) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int nt = g >> 1, r = 2 * (g & 1) + rr, dn = nt * 16 + 4 * r;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                double v = alpha * held[nt][mt][r];
                if (HAS_BETA) v = fma(beta, cv[rr * 4 + mt], v);
                hd_c[mt * 16 + (size_t)dn * ldc] = v;
            }
        }
    };
    auto plain_step = [&](int kt, bool has_next) {
        if (kt + 1 < KT) stage(buf ^ 1, cur_a, cur_b, kt + 1);
        else if (has_next) stage(buf ^ 1, nxt_a, nxt_b, 0);
        mfma_block(buf);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this k-step's DMA has landed, this wave's fragment reads are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        buf ^= 1;
    };

    locate(t, cur_a, cur_b, cur_c, cur_ci, cur_db);
    stage(0, cur_a, cur_b, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bool have_held = false;
    for (;;) {
        const int tn = t + W;
        const bool has_next = tn < r1;
        if (has_next) locate(tn, nxt_a, nxt_b, nxt_c, nxt_ci, nxt_db);
        if (have_held) {
            // k-steps 0..4 carry the held tile's epilogue (K >= 128: KT >= 8).  Issue order per k-step: stores of the previous quarter, the
            // LDS-DMA of the next operand tile, the loads of this quarter; vmcnt(8) retires all but the 8 loads.
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (j >= 1) quarter_store(j - 1);
                asm volatile("" ::: "memory");
                stage(buf ^ 1, cur_a, cur_b, j + 1);
                asm volatile("" ::: "memory");
                if (j < 4) quarter_load(j);
                asm volatile("" ::: "memory");
                mfma_block(buf);
                if (j < 4 && HAS_BETA) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                buf ^= 1;
            }
        } else {
            for (int kt = 0; kt < 5; ++kt) plain_step(kt, has_next);
        }
        for (int kt = 5; kt < KT; ++kt) plain_step(kt, has_next);
        if (cur_db == 0) {
            // diagonal tile: masked epilogue now (m < n is neither read nor written)
            have_held = false;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                double dv[4][4];
                if (HAS_BETA) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            dv[r][mt] = (dm + mt * 16 < nt * 16 + 4 * r) ? 0.0 : cur_ci[mt * 16 + (size_t)(nt * 16 + 4 * r) * ldcin];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        double v = alpha * acc[nt][mt][r];
                        if (HAS_BETA) v = fma(beta, dv[r][mt], v);
                        if (!(dm + mt * 16 < nt * 16 + 4 * r)) cur_c[mt * 16 + (size_t)(nt * 16 + 4 * r) * ldc] = v;
                    }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { held[a][b] = acc[a][b]; acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0}; }
            hd_c = cur_c, hd_ci = cur_ci;
            have_held = true;
        }
        if (!has_next) break;
        cur_a = nxt_a, cur_b = nxt_b, cur_c = nxt_c, cur_ci = nxt_ci, cur_db = nxt_db;
        t = tn;
    }
    // the last tile of this workgroup, if it is still held: a conventional epilogue
    if (have_held) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { quarter_load(g); quarter_store(g); }
    }
}

}  // namespace

// the persistent-tile kernel for short-K products (single problems and lockstep batches; no ktri, no urgent tiles)
void gpk_gemm_nt_pt(hipStream_t s, int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb, double beta, double *C,
                    int ldc, int lower, gp_batch bt, const double *Cin, int ldcin, int num_cu) {
    if (M <= 0 || N <= 0 || bt.count <= 0) return;
    const int tpp = lower ? ((N / TN) * (M / TM) - (N / TN) * ((N / TN) - 1) / 2) : (M / TM) * (N / TN);
    const int total = tpp * bt.count;
    const int grid = std::min(total, num_cu > 0 ? num_cu : 256);
    const bool hb = beta != 0.0;
#define GP_LAUNCH_PT(LO, HB) hipLaunchKernelGGL((gemm_pt_kernel<LO, HB>), dim3(grid), dim3(512), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bt, Cin, ldcin, tpp, total)
    if (lower) { if (hb) GP_LAUNCH_PT(1, 1); else GP_LAUNCH_PT(1, 0); }
    else { if (hb) GP_LAUNCH_PT(0, 1); else GP_LAUNCH_PT(0, 0); }
#undef GP_LAUNCH_PT
}



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
