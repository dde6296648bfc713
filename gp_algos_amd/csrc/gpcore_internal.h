// Internal declarations shared by the HIP translation units of libgpcore.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <functional>
#include <vector>

#include "../../include/gpcore.h"

// All device matrices are padded to a multiple of GP_NB rows/cols.  A padded SPD matrix carries an
// identity in the pad block, so chol([[K,0],[0,I]]) = [[L,0],[0,I]] and no kernel needs edge guards.
constexpr int GP_NB = 128;   // Cholesky panel width == GEMM tile edge
constexpr int GP_BK = 16;    // GEMM k-step
constexpr int GP_OUTER = 512; // outer panel width of the two-level Cholesky (trailing update runs with K = 512)
static inline int gp_pad(int n) { return (n + GP_NB - 1) / GP_NB * GP_NB; }
// tuning knob given as a column count: a positive multiple of GP_NB, else 0 (= use the default).  Read on every call, so a
// test can switch the blocked algorithms between their forms inside one process.
static inline int gp_env_blocks(const char *name) {
    const char *e = getenv(name);
    const int v = e ? atoi(e) : 0;
    return (v >= GP_NB && v % GP_NB == 0) ? v : 0;
}

struct gp_prof_slot {
    int64_t launches = 0;
    double work = 0.0;
    std::vector<hipEvent_t> ev;  // pairs (start, stop), resolved lazily at read time
};

struct gp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side = nullptr;      // look-ahead / overlap stream
    hipStream_t side2 = nullptr;     // second and third masked streams: the EP refactorisation that runs under the site loop (gp_ep_sweep):
    hipStream_t side3 = nullptr;     //   side2 the factorisation chain, side3 the rank-k updates of the next covariance
    bool own_stream = false;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    int prof_which = GP_PROF_OFF;
    gp_prof_slot prof[GP_PROF_NCLASSES];
    int num_cu = 256;
    int lookahead = -1;           // far trailing update of the Cholesky on the side stream: 1 on, 0 off, -1 by size (chol_blocked); GPCORE_LOOKAHEAD
    char err[512] = {0};
    // scratch reused across calls
    double *d_scalars = nullptr;  // small device scratch (256 doubles)
    int *d_info = nullptr;        // device-side failing-pivot flag
};

struct gp_model {
    gp_ctx *ctx = nullptr;
    int n = 0, d = 0, np = 0;        // np = padded n
    int ldl = 0;                     // leading dimension of dL = np + GP_NB: one extra row strip carries y^T through the factorisation
    bool has_x = false;              // false for gp_fit_from_gram
    int kind = 0;                    // kernel the model rebuilds Gram matrices with: 0 = ARD-RBF (theta = d + 2), 1 = Co2Kernel (d = 1, theta = 11)
    double *dX = nullptr;            // n x d, ld = n
    double *dcen = nullptr;          // d: centroid of the rows of dX (centre of the matrix-core Gram forms), set with dX
    double *dy = nullptr;            // np
    double *dL = nullptr;            // (np + GP_NB) x np, ld = ldl; row np holds y^T -> (L^-1 y)^T
    double *dalpha = nullptr;        // np
    double *ddinv = nullptr;         // np x 16: inverses of the 16x16 diagonal tiles of L
    double *dtmp = nullptr;          // 2 x np: t = L^-1 y (kept), scratch for the backward solve
    bool alpha_valid = false;        // alpha = L^-T t is computed lazily (predict and the LML only need t)
    double *dlml = nullptr;          // 1 double on device
    double *dLw = nullptr;           // np x np, lower: block row i = L_ii^-1 [ -L_i,<i | I ], built on demand for large posterior batches
    bool lw_valid = false;
    std::vector<double> theta;       // d + 2
    double sigma_noise = NAN;
    int last_info = 0;
};

#define GP_SET_ERR(ctx, ...) do { if (ctx) snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); } while (0)
#define GP_HIP(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    GP_SET_ERR(ctx, "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); return GP_EHIP; } } while (0)
// after a sequence of kernel launches: a launch that failed (bad configuration, wrong device current) must not go unnoticed
#define GP_LAUNCH_CHECK(ctx) GP_HIP(ctx, hipGetLastError())
#define GP_REQUIRE(ctx, cond, msg) do { if (!(cond)) { GP_SET_ERR(ctx, "invalid argument: %s", msg); return GP_EINVAL; } } while (0)
#define GP_TRY(call) do { gp_status s_ = (call); if (s_ != GP_OK) return s_; } while (0)

// ---- profiling helpers (gpcore_api.hip) ----
void gp_prof_begin(gp_ctx *ctx, int cls, hipStream_t s = nullptr);
void gp_prof_end(gp_ctx *ctx, int cls, double work, hipStream_t s = nullptr);

// A lockstep batch: `count` problems of identical shape handled by ONE launch (blockIdx.y picks the problem); the three
// strides (in doubles) are applied to the launcher's pointer operands in the order they appear in its signature.
struct gp_batch {
    int count = 1;
    size_t s0 = 0, s1 = 0, s2 = 0;
    // further strides, each named by the launcher that uses it: gpk_gemm_nt s3 = Cin; gpk_trsm_panel128 s3 = X2, s4 = tvec and cs2 (one
    // small per-problem buffer holds both), s5 = dots
    size_t s3 = 0, s4 = 0, s5 = 0;
    int tri = 0;   // trsm_panel128 only: problem g touches rows [0, (g+1)*128) -- the block columns of an upper-triangular matrix
};

// ---- kernel launchers (each asynchronous on `s`) ----
// C[MxN] = beta*C + alpha * A[MxK] * B[NxK]^T, column-major; M,N multiples of 128, K multiple of 16.
// lower != 0: M >= N, only tiles on/below the diagonal (bi >= bj) are computed; on diagonal tiles only i >= j is stored.
// ktri != 0: A(i,k) is zero for k < i (upper-triangular operand): each tile starts its k loop at its row block.
void gpk_gemm_nt(hipStream_t s, int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb,
                 double beta, double *C, int ldc, int lower, int ktri = 0, gp_batch bt = gp_batch(),   // strides: A, B, C
                 const double *Cin = nullptr, int ldcin = 0,    // beta term read from Cin instead of C (C = beta*Cin + alpha*A*B^T; batch stride bt.s3)
                 int *uflag = nullptr);   // single lower products: *uflag += 1 (release) when tile (1,0) / (1,1) is stored, by two workgroups at the head of the grid
// C (M x N; lower != 0: the lower trapezoid, i >= j on its diagonal tiles) -= A (M x 128) B (N x 128)^T (K a multiple of 32) on 64 x 64 tiles: the
// latency-bound updates of a single factorisation -- few 128 x 128 tiles, short K (M, N multiples of 64)
void gpk_gemm_k128_sub(hipStream_t s, int M, int N, const double *A, int lda, const double *B, int ldb, double *C, int ldc, int lower, int K = 128);
// C[M x 128] = A[M x K] * B[128 x K]^T with fused row reductions (sumsq[m] += sum_n C(m,n)^2, dots[m] += sum_n C(m,n) tvec[n]);
// C may be the last 128 columns of A (in-place posterior step).
void gpk_gemm_nt_rowred(hipStream_t s, int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc,
                        double *sumsq, const double *tvec, double *dots);
// ARD-RBF Gram.  theta on host.  symmetric: Xb == Xa, noise on the diagonal, tiles bi >= bj only (mirrored if full).
// far_flag: one int of device memory owned by the caller's context and used in stream order (gp_gram_flag(ctx)); nullptr = the per-pair
// kernel without the scan
// center: d doubles on the device (gpk_centroid of the training points; what z = (x - c) / l of the matrix-core forms is taken against), or
// nullptr = the first point
void gpk_gram_sym(hipStream_t s, const double *X, int n, int d, int ldx, const double *theta, double *K, int ldk, int full, double extra_diag, int *far_flag,
                  const double *center = nullptr);
void gpk_centroid(hipStream_t s, const double *X, int n, int d, int ldx, double *c);
void gpk_dgram_sym(hipStream_t s, const double *X, int n, int d, int ldx, const double *theta, int pos, double *D, int ldd);
void gpk_gram_cross(hipStream_t s, const double *Xs, int m, int ldxs, const double *X, int n, int ldx, int d, const double *theta, double *Ks, int ldks, int *far_flag,
                    const double *center = nullptr);
inline int *gp_gram_flag(gp_ctx *ctx) { return ctx->d_info + 4; }     // d_info holds 8 ints: [0..3] factorisation status, [4] the Gram scan's flag
// per-call centre of a Gram build whose inputs no model owns (GP_DMAX doubles behind the 8 ints of d_info, used in stream order)
inline double *gp_gram_center(gp_ctx *ctx) { return reinterpret_cast<double *>(ctx->d_info + 8); }
// Co2Kernel (gp/regression/Co2Prediction.scala:29-137), 1-D inputs, theta = hp1..hp11 on the host; pos = 0: kernel, 1..11: derivative
void gpk_co2_gram(hipStream_t s, const double *xr, int nr, const double *xc, int nc, const double *theta, int pos, double *K, int ldk, int sym,
                  int full, double extra);
// out[0] = 1/2 tr((alpha alpha^T - Kinv) D) for ANY symmetric D given as a matrix (lower triangles read; partial: n doubles of scratch):
// the per-parameter term of GpPredictor.logLikelihoodWithDerivatives (:76) when the derivative matrix is materialised
void gpk_co2_trace(hipStream_t s, int n, const double *alpha, const double *Kinv, int ldk, const double *D, int ldd, double *partial, double *out);
double gpk_co2_kss(const double *theta);
// set rows/cols [n, np) of the np x np matrix to identity (pad block) and zero the cross blocks.
void gpk_pad_identity(hipStream_t s, double *A, int n, int np, int lda);
void gpk_zero_upper(hipStream_t s, double *A, int n, int lda);
void gpk_fill(hipStream_t s, double *p, size_t count, double v);
// X (M x 128) <- X * Ukk^{-T} with Ukk UPPER triangular (back substitution per row)
void gpk_trsm_panel_upper(hipStream_t s, double *X, int M, int ldx, const double *Ukk, int ldu);
void gpk_transpose(hipStream_t s, double *dst, int ldd, const double *src, int lds, int rows, int cols);
// stage = 1: dst (upper form) <- -(strict block-lower part of src)^T with identity 128-blocks on the diagonal;
// stage = 0: dst (lower form) <- (upper form src)^T, block-lower part and diagonal blocks only.
void gpk_lw_transpose(hipStream_t s, double *dst, int ldd, const double *src, int lds, int np, int stage);
void gpk_set_identity(hipStream_t s, double *A, int n, int lda);
void gpk_set_identity_upper(hipStream_t s, double *A, int n, int lda);   // the 128-blocks on and above the diagonal only (n a multiple of 128)
void gpk_add_diag(hipStream_t s, double *A, int n, int lda, double v);   // A(i,i) += v, i < n
// LML-gradient traces (GpPredictor.scala:70-78 fused): out[0..d+1] = g_p for W = alpha alpha^T - Kinv (lower triangle of Kinv read)
void gpk_lml_grad_traces(hipStream_t s, const double *X, int n, int d, int ldx, const double *theta, const double *alpha,
                         const double *Kinv, int ldk, double *partials, double *out);
int gpk_lml_grad_partials_size(int n, int d);
// y (len M) -= A (M x 128, lda) * x (128)
void gpk_gemv_panel_sub(hipStream_t s, const double *A, int M, int lda, const double *x, double *y);
// out[i] = sum_j Ks(i,j) * alpha[j], deterministic, j ascending per row chunk
void gpk_gemv_rows(hipStream_t s, const double *Ks, int m, int n, int ldks, const double *alpha, double *out, double *partial, int nchunk, int upper_blocks = 0);
// lml = -0.5 t.t - sum log L_ii - n/2 log 2pi with t = L^-1 y  (y.alpha = |L^-1 y|^2; n real rows)
void gpk_lml(hipStream_t s, const double *L, int n, int ldl, const double *t, double *out);
// var[i] = kss - sumsq[i]
void gpk_var_finish(hipStream_t s, double *var, const double *sumsq, int m, double kss);
void gpk_copy_2d(hipStream_t s, double *dst, int ldd, const double *src, int lds, int rows, int cols);
double gpk_probe_mfma(hipStream_t s, int num_cu, int waves_per_simd, double *clock_mhz, double *cycles_per_mfma);
int gpk_init_diag_kernels();
int gpk_init_gemm_kernels();
// MFMA-blocked critical-path kernels (kernels_diag.hip).  dinv holds the inverses of the 16x16 diagonal
// tiles of L: tile b (rows 16b..16b+15) at dinv + 256*b, element (c,k) at c + 16k; a 128-block owns 8 tiles.
void gpk_potrf_diag128(hipStream_t s, double *A, int lda, double *dinv_k, int *d_info, int base, gp_batch bt = gp_batch());
// the whole two-level factorisation of one matrix as one persistent launch over a task list (16 ints per task: kernels_diag.hip chol_task)
void gpk_chol_mega(hipStream_t s, int num_cu, double *A, int lda, double *dinv, int *d_info, const int *tasks, int ntasks, int *done, int *counter,
                   int epoch, int *err, double *sums, unsigned long long *stamps = nullptr);   // sums: one NB x NB scratch tile per outer panel   // stamps: lab, 4 x 8 bytes per task   // strides: A, dinv; d_info + 1 per problem
// optional fused row reductions: sumsq[p] += sum_c X(p,c)^2 ; dots[p] += sum_c X(p,c) tvec[c]
// EP: the 128 rows of the delayed columns that belong to the next site block (X <- X Lmat^-T in place, X2 = X diag(cs2), dots += X tvec)
// and the 128 x 128 lower tile D -= X2 X^T the next block kernel reads, one workgroup (kernels_diag.hip)
void gpk_ep_link(hipStream_t s, double *X, int ldx, const double *Lmat, const double *dinv, const double *tvec, double *dots, double *X2,
                 const double *cs2, double *D, int ldd);
void gpk_trsm_panel128(hipStream_t s, double *X, int M, int ldx, const double *Lkk, int ldl, const double *dinv_k, double *sumsq,
                       const double *tvec = nullptr, double *dots = nullptr, gp_batch bt = gp_batch(),   // strides: X, Lkk, dinv; s3 X2, s4 tvec / cs2, s5 dots (sumsq: single problems only)
                       double *X2 = nullptr, const double *cs2 = nullptr);   // optional second output X2(p,c) = X(p,c) * cs2[c] (same ld as X)
void gpk_fwd_step(hipStream_t s, const double *L, int ldl, const double *dinv_k, double *t, double *sol, int k0, int r);
void gpk_bwd_step(hipStream_t s, const double *L, int ldl, const double *dinv_k, double *t, double *sol, int k0);
void gpk_tile_inverses(hipStream_t s, const double *L, int np, int ldl, double *dinv);
// dst[j*dst_stride] = src[j*src_stride], j < count
void gpk_copy_strided(hipStream_t s, double *dst, size_t dst_stride, const double *src, size_t src_stride, int count);

// ---- host-side helpers shared between translation units (defined in gpcore_api.hip) ----
enum { WS_VT = 0, WS_PARTIAL, WS_SUMSQ, WS_A, WS_B, WS_C, WS_D, WS_E, WS_F, WS_G, WS_H, WS_COUNT };
gp_status gpi_ws_get(gp_ctx *ctx, int slot, size_t bytes, double **out);
gp_status gpi_upload_2d(gp_ctx *ctx, double *dst, int ldd, const double *src, int lds, int rows, int cols);
gp_status gpi_download_2d(gp_ctx *ctx, double *dst, int ldd, const double *src, int lds, int rows, int cols);
gp_status gpi_read_info(gp_ctx *ctx, int *info);
gp_ctx *gpi_child_ctx(gp_ctx *ctx, int k);
gp_status gpi_model_alpha(gp_model *m, double *dst);
gp_status gpi_model_alloc(gp_ctx *ctx, int n, int d, bool has_x, gp_model **out);
// L-BFGS (`history` pairs, at most max_iter iterations, NC batched backtracking steps per iteration, best-seen rule) maximising
// F over the first nparams entries of theta: the driver behind gp_optimize_rbf and gp_ep_optimize_rbf
gp_status gpi_lbfgs_maximize(gp_ctx *ctx, int P, int nparams, const double *theta0, int max_iter, int history, int NC,
                             const std::function<gp_status(const double *, int, double *, double *, int *)> &evaluate,
                             double *theta_out, double *f_out, int *iters_out, int *evals_out);
gp_status gpi_ctx_ep_streams(gp_ctx *ctx);   // side2 / side3, created on first use
void gpi_chol_blocked(gp_ctx *ctx, double *A, int np, int lda, double *dinv, int extra);
// one 128-column step (diagonal block k0) of the same two-level factorisation on stream s, for callers that feed the columns
// one block at a time: diagonal factor, panel solve, in-panel update, and the K = OUTER trailing update when k0 closes an outer panel
void gpi_chol_panel_step(gp_ctx *ctx, hipStream_t s, double *A, int np, int lda, double *dinv, int extra, int k0, hipEvent_t solved = nullptr,
                         hipStream_t far = nullptr, hipEvent_t far_done = nullptr,
                         int count = 1, size_t strideA = 0, size_t strideDinv = 0, int *info = nullptr);   // count > 1: a lockstep batch (info + g per problem)
void gpi_solve_rows_lower(gp_ctx *ctx, double *Vt, int mp, const double *L, int np, int ldl, const double *dinv, double *sumsq,
                          const double *tvec = nullptr, double *dots = nullptr);
void gpi_back_solve_vec(gp_ctx *ctx, const double *L, int np, int ldl, const double *dinv, double *z, double *alpha);
void gpi_inverse_transpose_lower(gp_ctx *ctx, double *T, const double *L, int np, int ldl, const double *dinv);
void gpi_forward_solve_vec(gp_ctx *ctx, const double *L, int np, int ldl, const double *dinv, double *t, double *z);
