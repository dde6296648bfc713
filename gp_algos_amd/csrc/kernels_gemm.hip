// fp64 MFMA GEMM for gfx950 (MI355X):  C = beta*C + alpha * A * B^T   ("NT", everything column-major).
//
// This one kernel carries every O(n^3) term of the reference's hot path:
//   - Cholesky trailing update  A22 -= P P^T            (breeze.linalg.cholesky, GpPredictor.scala:120)
//   - posterior solve           Vt_i -= Vt_{<i} L_i,<i^T  (forwardSolve(L, K*^T), GpPredictor.scala:34)
//   - full predictive covariance Gram(X*) - Vt Vt^T      (GpPredictor.scala:36)
//   - K^-1 = L^-T L^-1 for the LML gradient              (GpPredictor.scala:66-67)
//   - EP  Sigma = K - V^T V                               (EpParameterEstimator.scala:60)
//
// Design (CDNA4): 128x128 output tile per workgroup of 8 waves (2 x 4, 64x32 per wave = 4x2
// v_mfma_f64_16x16x4_f64 accumulators; a 4-wave 64x64-per-wave variant is kept for A/B runs), BK = 16,
// LDS double-buffered and filled by LDS-DMA (global_load_lds_dwordx4, no staging VGPRs, no ds_write
// pass).  Both operands are "row-contiguous" in memory (column-major, NT form), so one wave
// instruction of 64 x 16 B moves a full 1 KiB tile column: perfectly coalesced.  LDS image is
// [k][row] with a row stride of 144 doubles (1152 B = 128 mod 256 B), which makes the
// ds_read_b64 fragment reads (lanes 0-15: 16 rows at k, lanes 16-31: the same rows at k+1)
// bank-conflict free.  The MFMA is issued with the B-fragment as srcA and the A-fragment as srcB
// so each lane's accumulator holds 4 consecutive-n values at one m: stores are 128-B row segments
// of column-major C.
#include <type_traits>
#include "gpcore_internal.h"
#include <algorithm>

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

// Row reductions fused into the epilogue (posterior step: C is one 128-column block of V^T, the variance needs the row sums
// of squares and the mean needs the row dot products with t = L^-1 y):  sumsq[m] += sum_n C(m,n)^2,  dots[m] += sum_n C(m,n) tvec[n].
struct gemm_rowred {
    double *sumsq = nullptr;
    const double *tvec = nullptr;   // N values (this block column's slice of t); may be null together with dots
    double *dots = nullptr;
};

template <int LOWER, int HAS_BETA, int NW, int RR = 0>
__global__ __launch_bounds__(NW * 64, NW / 2) void gemm_nt_f64_kernel(int M, int N, int K, double alpha, const double *__restrict__ A, int lda,
                                                           const double *__restrict__ B, int ldb, double beta,
                                                           double *__restrict__ C, int ldc, int ktri, gp_batch bt, gemm_rowred rr,
                                                           const double *__restrict__ Cin, int ldcin, int *__restrict__ uflag) {
    __shared__ __attribute__((aligned(16))) double smem[2 * 2 * TK * LDS_STRIDE];
    double *As = smem;                          // [2][TK][LDS_STRIDE]
    double *Bs = smem + 2 * TK * LDS_STRIDE;    // [2][TK][LDS_STRIDE]

    int bi, bj;
    bool urgent = false;
    int prob = blockIdx.y;   // problem of a lockstep batch (same shapes, operands bt.s* doubles apart); a single problem has gridDim.y == 1
    if (LOWER && ktri) {
        // The k loop of tile row bi runs over K - bi*128, so equal tile COUNTS per XCD would leave the XCD holding the first rows
        // with twice the average work (measured: 30 TFLOP/s).  Deal whole tile rows to the XCDs instead (XCD x owns rows
        // x, x+8, ...): the k-work per XCD is then equal within 3 %, a row's tiles share their A row panel in that XCD's L2,
        // and each XCD starts with its longest rows.  A batch multiplies every row's entry count (1-D grid, gridDim.y == 1).
        const int x = blockIdx.x & 7, nb = M / TM;
        int e = blockIdx.x >> 3;
        bi = -1;
        for (int r = x; r < nb; r += 8) {
            const int cnt = (r + 1) * bt.count;
            if (e < cnt) { bi = r; prob = e / (r + 1); bj = e - prob * (r + 1); break; }
            e -= cnt;
        }
        if (bi < 0) return;   // padding workgroups of the XCDs that own fewer tiles
    } else if (LOWER) {
        // XCD-aware: block ids are dealt round-robin over the 8 XCDs; hand each XCD a contiguous run of the order above
        int nwg = gridDim.x, id = blockIdx.x;
        if (uflag) {
            // "urgent" tiles (single problems; the EP chain): the two tiles of tile row 1 -- what the next chain kernel reads of this
            // update -- are computed by two extra workgroups at the head of the grid, which announce them (release, agent scope) as
            // soon as they are stored; the regular workgroups that own them step aside.  The chain kernel waits for the counter
            // instead of for the whole launch.
            nwg -= 2;
            if (id < 2) { urgent = true; bi = 1; bj = id; }
            else id -= 2;
        }
        if (!urgent) {
            const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
            const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
            tile_coords_lower(swz, M / TM, N / TN, bi, bj);
            if (uflag && bi == 1) return;
        }
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
    A += (size_t)prob * bt.s0;
    B += (size_t)prob * bt.s1;
    C += (size_t)prob * bt.s2;
    if (!Cin) { Cin = C; ldcin = ldc; }   // the beta term is read from C itself unless the caller names another source
    else Cin += (size_t)prob * bt.s3;
    const int row0 = bi * TM, col0 = bj * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = (NW == 8) ? 2 : 4;          // 16-column accumulator tiles per wave (8 waves: 64 x 32 per wave)
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * (NT * 16);

    // ktri: A(i,k) == 0 for k < i (e.g. L^-T): the product over k can start at this tile's row block
    const int kbeg = ktri ? row0 : 0;
    const double *Ag = A + row0 + (size_t)kbeg * lda;
    const double *Bg = B + col0 + (size_t)kbeg * ldb;

    double4_t acc[NT][4];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};

    const int KT = (K - kbeg) / TK;
    // Staging: LDS-DMA (global_load_lds_dwordx4).  One wave instruction moves 64 lanes x 16 B = one full k-row of the
    // tile (128 doubles = 1 KiB, contiguous in LDS; the 16-double pad follows it), so the lane-linear destination rule
    // is met with the padded image.  No staging VGPRs, no ds_write pass; wave w fetches k-rows w, w+NW, ...
    const double *Asrc = Ag + lane * 2 + (size_t)wave * lda;
    const double *Bsrc = Bg + lane * 2 + (size_t)wave * ldb;
    auto stage = [&](int buf, int kt) {
        const size_t koff = (size_t)kt * TK;
#pragma unroll
        for (int q = 0; q < TK / NW; ++q) {
            __builtin_amdgcn_global_load_lds(Asrc + (koff + NW * q) * lda, As + (buf * TK + wave + NW * q) * LDS_STRIDE, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(Bsrc + (koff + NW * q) * ldb, Bs + (buf * TK + wave + NW * q) * LDS_STRIDE, 16, 0, 0);
        }
    };
    if (KT > 0) stage(0, 0);
    __syncthreads();

    const int fr = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) stage(cur ^ 1, kt + 1);
        const double *Ac = As + cur * TK * LDS_STRIDE + wm + fr;
        const double *Bc = Bs + cur * TK * LDS_STRIDE + wn + fr;
#pragma unroll
        for (int ks = 0; ks < TK / 4; ++ks) {
            double af[4], bf[NT];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = Ac[(ks * 4 + fk) * LDS_STRIDE + t * 16];
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = Bc[(ks * 4 + fk) * LDS_STRIDE + t * 16];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        __syncthreads();   // drains the DMA of tile kt+1 (vmcnt) and fences the reads of tile kt
    }

    // epilogue: acc[nt][mt][r] = D[n = wn + nt*16 + fk + 4r][m = wm + mt*16 + fr].
    // C is read in batches of 16 independent loads per 16-column group, then written: a load/wait/store chain per
    // element (what a naive `v += beta * *cp` compiles to) serialises 64 memory round trips per tile.
    // Lower-trapezoid launches: only the tiles ON the diagonal mask anything, and the two cases are separate code.  With the mask's test
    // around every load and store of every tile (`(diag_tile && m < n) ? 0.0 : load`, one exec-masked block per element) the compiler
    // cannot count what is outstanding and puts `s_waitcnt vmcnt(0)` in front of each store: every store then waits for the
    // acknowledgement of the one before it, 32 round trips per tile on every tile of every trailing update (found in round 4 in the
    // single-launch factorisation's copy of this loop; rounds 1-3 ran with it).
    const bool diag_tile = LOWER && (bi == bj);
    double rsq[4] = {0.0, 0.0, 0.0, 0.0}, rdt[4] = {0.0, 0.0, 0.0, 0.0};   // RR: per (mt) row partial sums of this lane
    auto epilogue = [&](auto masked_t) {
        constexpr bool MASKED = decltype(masked_t)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double cv[4][4];
            if (HAS_BETA) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = col0 + wn + nt * 16 + fk + 4 * r;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int m = row0 + wm + mt * 16 + fr;
                        cv[r][mt] = (MASKED && m < n) ? 0.0 : Cin[m + (size_t)n * ldcin];
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
                    if (!(MASKED && m < n)) C[m + (size_t)n * ldc] = v;
                    if (RR) {
                        rsq[mt] = fma(v, v, rsq[mt]);
                        if (rr.dots) rdt[mt] = fma(v, rr.tvec[n], rdt[mt]);
                    }
                }
            }
        }
    };
    if (diag_tile) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if (urgent) {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's stores are acknowledged
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            __hip_atomic_fetch_add(uflag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (RR) {
        // lane (fr, fk) holds rows wm + mt*16 + fr over its columns; fold the 4 fk groups, then the NW/2 waves that share wm,
        // in a fixed order, and let one thread per row add into the global accumulators (this tile owns its rows in this launch)
        double *red = smem;   // [2][NW/2][TM]: the staging buffers are free after the k loop's last barrier
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            rsq[mt] += __shfl_xor(rsq[mt], 16, 64);
            rsq[mt] += __shfl_xor(rsq[mt], 32, 64);
            rdt[mt] += __shfl_xor(rdt[mt], 16, 64);
            rdt[mt] += __shfl_xor(rdt[mt], 32, 64);
            if (fk == 0) {
                red[(wave >> 1) * TM + wm + mt * 16 + fr] = rsq[mt];
                red[(NW / 2 + (wave >> 1)) * TM + wm + mt * 16 + fr] = rdt[mt];
            }
        }
        __syncthreads();
        if (tid < TM) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int g = 0; g < NW / 2; ++g) { a += red[g * TM + tid]; b += red[(NW / 2 + g) * TM + tid]; }
            rr.sumsq[row0 + tid] += a;
            if (rr.dots) rr.dots[row0 + tid] += b;
        }
    }
}

// The posterior step on 256 x 128 tiles by ONE workgroup per CU (16 waves x 64x32, or 8 waves x 64x64 = default) instead of two
// 8-wave workgroups on 128 x 128 tiles: one barrier domain and the B tile (the Lw block row, shared by every row of V^T)
// staged once per CU -- 48 KB instead of 64 KB of LDS-DMA per k-tile, 106 KB of LDS.
// Measured +2.5 % on the C2 posterior (66.4 vs 68.0 ms per 65 536 points): with two independent workgroups per CU the pair
// ran at 95 % of the pipe rate while both were resident (timestamps in DESIGN.md section 7).
constexpr int FM = 256, FSTRIDE = FM + 16;
// LOWER: tiles (bi, bj) of 256 x 128 whose last row reaches the diagonal (256 bi + 255 >= 128 bj); elements above the diagonal
// are computed and not stored.  Tile rows are enumerated bi ascending, bj ascending inside.
template <int LOWER, int HAS_BETA, int RR, int WV = 16>
__global__ __launch_bounds__(WV * 64, 1) void gemm_fused_kernel(int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb,
                                                            double beta, double *C, int ldc, gp_batch bt, gemm_rowred rr) {
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    double *As = fsm;                           // [2][TK][FSTRIDE]
    double *Bs = fsm + 2 * TK * FSTRIDE;        // [2][TK][LDS_STRIDE]
    const int nbm = M / FM, nbn = N / TN;
    int bi, bj;
    if (LOWER) {
        int t = blockIdx.x;
        bi = 0;
        for (;;) {   // row bi holds min(nbn, 2 bi + 2) tiles
            const int cnt = (2 * bi + 2 < nbn) ? 2 * bi + 2 : nbn;
            if (t < cnt) break;
            t -= cnt;
            ++bi;
        }
        bj = t;
    } else {
        // XCD-aware: consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous run of tiles sharing a B panel
        const int nwg = gridDim.x, id = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = id & 7;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
        bi = swz % nbm;
        bj = swz / nbm;
    }
    A += (size_t)blockIdx.y * bt.s0;
    B += (size_t)blockIdx.y * bt.s1;
    C += (size_t)blockIdx.y * bt.s2;
    const int row0 = bi * FM, col0 = bj * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = (WV == 16) ? 2 : 4;      // 16-column accumulator tiles per wave: 64 x 32 (16 waves) or 64 x 64 (8 waves)
    const int wm = (wave & 3) * 64, wn = (wave >> 2) * (NT * 16);
    const int fr = lane & 15, fk = lane >> 4;
    const double *Asrc = A + row0 + lane * 2 + (size_t)wave * lda;   // wave w stages k-rows w, w + WV, ...: two halves of A, one row of B
    const double *Bsrc = B + col0 + lane * 2 + (size_t)wave * ldb;
    auto stage = [&](int buf, int kt) {
        const size_t koff = (size_t)kt * TK;
#pragma unroll
        for (int q = 0; q < TK / WV; ++q) {
            const int kr = wave + WV * q;
            __builtin_amdgcn_global_load_lds(Asrc + (koff + WV * q) * lda, As + (buf * TK + kr) * FSTRIDE, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(Asrc + 128 + (koff + WV * q) * lda, As + (buf * TK + kr) * FSTRIDE + 128, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(Bsrc + (koff + WV * q) * ldb, Bs + (buf * TK + kr) * LDS_STRIDE, 16, 0, 0);
        }
    };
    double4_t acc[NT][4];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int KT = K / TK;
    if (KT > 0) stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) stage(cur ^ 1, kt + 1);
        const double *Ac = As + cur * TK * FSTRIDE + wm + fr;
        const double *Bc = Bs + cur * TK * LDS_STRIDE + wn + fr;
#pragma unroll
        for (int ks = 0; ks < TK / 4; ++ks) {
            double af[4], bf[NT];
#pragma unroll
            for (int t = 0; t < 4; ++t) af[t] = Ac[(ks * 4 + fk) * FSTRIDE + t * 16];
#pragma unroll
            for (int t = 0; t < NT; ++t) bf[t] = Bc[(ks * 4 + fk) * LDS_STRIDE + t * 16];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        __syncthreads();
    }
    const bool diag = LOWER && (row0 < col0 + TN);   // the tile touches or crosses the diagonal
    double rsq[4] = {0.0, 0.0, 0.0, 0.0}, rdt[4] = {0.0, 0.0, 0.0, 0.0};
    auto epilogue = [&](auto masked_t) {      // masked and unmasked tiles as separate code: see gemm_nt_f64_kernel
        constexpr bool MASKED = decltype(masked_t)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double cv[4][4];
            if (HAS_BETA) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = col0 + wn + nt * 16 + fk + 4 * r;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const int m = row0 + wm + mt * 16 + fr;
                        cv[r][mt] = (MASKED && m < n) ? 0.0 : C[m + (size_t)n * ldc];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = col0 + wn + nt * 16 + fk + 4 * r;
                const double tn = (RR && rr.dots) ? rr.tvec[n] : 0.0;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int m = row0 + wm + mt * 16 + fr;
                    double v = alpha * acc[nt][mt][r];
                    if (HAS_BETA) v = fma(beta, cv[r][mt], v);
                    if (!(MASKED && m < n)) C[m + (size_t)n * ldc] = v;
                    if (RR) {
                        rsq[mt] = fma(v, v, rsq[mt]);
                        rdt[mt] = fma(v, tn, rdt[mt]);
                    }
                }
            }
        }
    };
    if (diag) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if (RR) {
        constexpr int NG = WV / 4;   // column groups of waves
        double *red = fsm;   // [2][NG][FM]
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            rsq[mt] += __shfl_xor(rsq[mt], 16, 64);
            rsq[mt] += __shfl_xor(rsq[mt], 32, 64);
            rdt[mt] += __shfl_xor(rdt[mt], 16, 64);
            rdt[mt] += __shfl_xor(rdt[mt], 32, 64);
            if (fk == 0) {
                red[(wave >> 2) * FM + wm + mt * 16 + fr] = rsq[mt];
                red[(NG + (wave >> 2)) * FM + wm + mt * 16 + fr] = rdt[mt];
            }
        }
        __syncthreads();
        if (tid < FM) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int g = 0; g < NG; ++g) { a += red[g * FM + tid]; b += red[(NG + g) * FM + tid]; }
            rr.sumsq[row0 + tid] += a;
            if (rr.dots) rr.dots[row0 + tid] += b;
        }
    }
}

// C -= A B^T (K a multiple of 32) on 64 x 64 tiles (4 waves x 32 x 32): the in-panel update of the blocked Cholesky
// (A21 column block -= A21_k A21_k^T confined to one outer panel: r rows x <= 384 columns).  With the general kernel's 128 x 128
// tiles such a launch is a few hundred workgroups of 14 us of MFMA issue each -- one round, bound by the latency of a single
// tile, 25 us whatever the size.  Quarter-size tiles put 4x the workgroups on the chip and shorten the dependent chain per
// workgroup 4x; the operands (the panel the row solve has just written) come from L2, staged through LDS in chunks of 32 k
// with the next chunk's loads in flight in registers.  LOWER: tiles on/below the diagonal of C only, i >= j stored on its tiles.
constexpr int SK_T = 64, SK_S = 80, SK_C = 32;
template <int LOWER>
__global__ __launch_bounds__(256) void gemm_k128_kernel(int M, int N, int K, const double *__restrict__ A, int lda, const double *__restrict__ B,
                                                        int ldb, double *__restrict__ C, int ldc) {
    __shared__ __attribute__((aligned(16))) double as[SK_C * SK_S], bs[SK_C * SK_S];
    int bi, bj;
    const int tm = M / SK_T;
    if (LOWER) {   // column-major enumeration of the lower trapezoid: tile column bj holds rows bj .. tm-1
        int t = blockIdx.x;
        bj = 0;
        while (t >= tm - bj) { t -= tm - bj; ++bj; }
        bi = bj + t;
    } else {
        bi = blockIdx.x % tm;
        bj = blockIdx.x / tm;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const int i0 = bi * SK_T, j0 = bj * SK_T, wi = (wave & 1) * 32, wj = (wave >> 1) * 32;
    const double *Ap = A + i0 + lane + (size_t)wave * lda, *Bp = B + j0 + lane + (size_t)wave * ldb;   // thread: row `lane`, k = wave + 4 q
    double ra[SK_C / 4], rb[SK_C / 4];
#pragma unroll
    for (int q = 0; q < SK_C / 4; ++q) { ra[q] = Ap[(size_t)(4 * q) * lda]; rb[q] = Bp[(size_t)(4 * q) * ldb]; }
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int kc = 0; kc < K; kc += SK_C) {
        if (kc) __syncthreads();
#pragma unroll
        for (int q = 0; q < SK_C / 4; ++q) { as[(wave + 4 * q) * SK_S + lane] = ra[q]; bs[(wave + 4 * q) * SK_S + lane] = rb[q]; }
        if (kc + SK_C < K) {
#pragma unroll
            for (int q = 0; q < SK_C / 4; ++q) {
                ra[q] = Ap[(size_t)(kc + SK_C + 4 * q) * lda];
                rb[q] = Bp[(size_t)(kc + SK_C + 4 * q) * ldb];
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < SK_C / 4; ++ks) {
            double af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = as[(4 * ks + fk) * SK_S + wi + 16 * t + fr];     // rows of C
                bf[t] = bs[(4 * ks + fk) * SK_S + wj + 16 * t + fr];     // columns of C
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[nt], af[mt], acc[nt][mt], 0, 0, 0);   // reg r: (n = fk + 4 r, m = fr)
        }
    }
    const bool diag_tile = LOWER && (bi == bj);
    auto epilogue = [&](auto masked_t) {      // masked and unmasked tiles as separate code: see gemm_nt_f64_kernel
        constexpr bool MASKED = decltype(masked_t)::value;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                double cv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = i0 + wi + 16 * mt + fr, n = j0 + wj + 16 * nt + fk + 4 * r;
                    cv[r] = (MASKED && m < n) ? 0.0 : C[m + (size_t)n * ldc];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = i0 + wi + 16 * mt + fr, n = j0 + wj + 16 * nt + fk + 4 * r;
                    if (!(MASKED && m < n)) C[m + (size_t)n * ldc] = cv[r] - acc[nt][mt][r];
                }
            }
    };
    if (diag_tile) epilogue(std::true_type{});
    else epilogue(std::false_type{});
}

constexpr int FUSED_LDS = (2 * TK * FSTRIDE + 2 * TK * LDS_STRIDE) * (int)sizeof(double);

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
                 double beta, double *C, int ldc, int lower, int ktri, gp_batch bt, const double *Cin, int ldcin, int *uflag) {
    if (M <= 0 || N <= 0 || bt.count <= 0) return;
    if (uflag && !(lower && !ktri && bt.count == 1 && M >= 2 * TM && N >= 2 * TN)) uflag = nullptr;   // urgent tiles: single lower products with a tile row 1
    const gemm_rowred rr;
    const bool hb = beta != 0.0;
    // (The 256 x 128 / 16-wave tile of the posterior step was also tried here for the general launches: C3 658-663 vs 668-675
    // settings/s, C4 75.8 vs 82.3 sweeps/s -- with several column tiles per launch and short k its unhidden prologue and
    // epilogue cost more than the shared B tile saves.  Not used.)
    // 8 waves per workgroup (64 x 32 per wave, 4 waves/SIMD at 2 workgroups/CU).  A 4-wave form (64 x 64 per wave, 178 VGPRs) measured
    // equal on long-K launches (65 TFLOP/s both) and ~5 % worse on the short-K Cholesky updates; it was an A/B switch until round 4.
    int ntiles = lower ? ((N / TN) * (M / TM) - (N / TN) * ((N / TN) - 1) / 2) : (M / TM) * (N / TN);   // lower: trapezoid, M >= N
    int gy = bt.count;
    if (uflag) ntiles += 2;
    if (ktri) {
        if (!lower || M != N) return;   // ktri is the square lower product T T^T only
        int most = 0;                   // grid = 8 x (entries of the XCD that owns the most tiles)
        for (int x = 0; x < 8; ++x) {
            int cnt = 0;
            for (int r = x; r < M / TM; r += 8) cnt += (r + 1) * bt.count;
            most = cnt > most ? cnt : most;
        }
        ntiles = 8 * most;
        gy = 1;
    }
#define GP_LAUNCH(LO, HB) hipLaunchKernelGGL((gemm_nt_f64_kernel<LO, HB, 8>), dim3(ntiles, gy), dim3(512), 0, s, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, ktri, bt, rr, Cin, ldcin, uflag)
    if (lower) { if (hb) GP_LAUNCH(1, 1); else GP_LAUNCH(1, 0); }
    else { if (hb) GP_LAUNCH(0, 1); else GP_LAUNCH(0, 0); }
#undef GP_LAUNCH
}

// C = A B^T (no beta) with the row reductions above; C may alias the last N columns of A (each tile reads only its own
// rows of A and stores after its k loop) -- the in-place posterior step.
// per device, from gp_ctx_create: the fused kernel's 106 KB of LDS is above the default dynamic limit
int gpk_init_gemm_kernels() {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_fused_kernel<0, 0, 1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_LDS) == hipSuccess ? 0 : 1;
}

void gpk_gemm_nt_rowred(hipStream_t s, int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc,
                        double *sumsq, const double *tvec, double *dots) {
    if (M <= 0 || N != TN) return;   // one column tile: a tile is the only writer of its rows' accumulators
    // 256-row tiles by ONE 8-wave workgroup per CU where they fit (8 waves x 64x64; 16 waves x 64x32 measured 0.6 % slower: a third more
    // LDS fragment reads in a power-limited loop); an odd 128-row remainder goes through the 128-row kernel below
    if (M >= FM) {
        // One workgroup per CU: a launch costs whole ROUNDS of num_cu tiles, whatever the last round holds.  Full rounds go to the
        // 256-row kernel; of what is left, up to num_cu 128-row tiles are cheaper on the 8-wave kernel below (one workgroup per CU on
        // half a tile's work: 0.55-0.6 of a round) than a 256-row round that is at most half full.  (Config C5: the last batch of a
        // 10^6-point request, 82 496 rows = 322 tiles, cost two rounds -- as much as a full batch of 131 072; profiles/r03_n_c5_*.)
        static const int ncu = [] { int dev = 0; hipDeviceProp_t pr; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
        int Mf = M / FM * FM;
        const int rem_tiles = (Mf / FM) % ncu;
        if (Mf / FM > ncu && rem_tiles > 0 && rem_tiles * 2 <= ncu) Mf -= rem_tiles * FM;      // the partial round goes to the 128-row kernel
        else if (Mf / FM < ncu && Mf / FM * 2 <= ncu / 2) Mf = 0;                               // a lone quarter round or less: 128-row tiles fill more CUs
        if (Mf == 0) goto small_tiles;
        gemm_rowred r2;
        r2.sumsq = sumsq, r2.tvec = tvec, r2.dots = dots;
        hipLaunchKernelGGL((gemm_fused_kernel<0, 0, 1, 8>), dim3(Mf / FM), dim3(512), FUSED_LDS, s, Mf, TN, K, 1.0, A, lda, B, ldb, 0.0, C, ldc, gp_batch(), r2);
        if (Mf == M) return;
        A += Mf, C += Mf, sumsq += Mf, M -= Mf;
        if (dots) dots += Mf;
    }
small_tiles:
    gemm_rowred rr;
    rr.sumsq = sumsq, rr.tvec = tvec, rr.dots = dots;
    const int ntiles = (M / TM) * (N / TN);
    hipLaunchKernelGGL((gemm_nt_f64_kernel<0, 0, 8, 1>), dim3(ntiles), dim3(512), 0, s, M, N, K, 1.0, A, lda, B, ldb, 0.0, C, ldc, 0, gp_batch(), rr, nullptr, 0, nullptr);
}

// C (M x N, lower trapezoid if `lower`) -= A (M x 128) B (N x 128)^T; M, N multiples of 64
void gpk_gemm_k128_sub(hipStream_t s, int M, int N, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int lower, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return;
    const int tm = M / SK_T, tn = N / SK_T;
    if (lower) hipLaunchKernelGGL(gemm_k128_kernel<1>, dim3(tn * tm - tn * (tn - 1) / 2), dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc);
    else hipLaunchKernelGGL(gemm_k128_kernel<0>, dim3(tm * tn), dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc);
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
