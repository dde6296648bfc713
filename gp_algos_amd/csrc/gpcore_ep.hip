// EP for probit GP binary classification on a ready-made Gram matrix (gfx950).
//   EpParameterEstimator.estimateSiteParams  gp/classification/EpParameterEstimator.scala:29-69
//   EpParameterEstimator.epMarginalLikelihood :71-96, marginalMoments :98-109
//   GpClassifier.classify                     gp/classification/GpClassifier.scala:24-47
//
// The reference visits the n sites strictly in order and after each one applies a rank-1 update to the
// full n x n Sigma plus a full Sigma*nu gemv (24 n^3 bytes of traffic per sweep).  Same mathematics here,
// restructured as DELAYED rank-128 updates:
//   inside a block of 128 sites:  s_t = Sigma0[:, i_t] - S (c o S[i_t, :]^T)   (O(n t), panel S is L2-resident)
//                                 mu <- mu + s_t (dnu - c_t (mu_i + dnu s_ii))  (O(n), replaces the gemv)
//   after the block:              Sigma0 <- Sigma0 - S diag(c) S^T              (fp64 MFMA GEMM, 2 n^2 128 flops)
// so a sweep costs 2 n^3 MFMA flops and 16 n^3 / 128 bytes instead of 24 n^3 bytes.  The end-of-sweep
// refactorisation (L = chol(I + S^1/2 K S^1/2), V = L \ S^1/2 K, Sigma = K - V^T V, mu = Sigma nu) reuses
// the blocked Cholesky, the row-panel solve and the MFMA syrk of the regression path.
#include "gpcore_internal.h"
#include "dpp_tile.h"
#include "trsm_tile.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <new>
#include <thread>
#include <vector>

// Outer panel of the site loop's delayed updates (two-level, like the Cholesky): the columns up to one block past the current outer
// panel take every block's rank-128 update at once (the chain needs them), everything to the right ONE rank-EP_OUTER update per panel.
constexpr int EP_OUTER = 4 * GP_NB;

struct gp_ep {
    gp_ctx *ctx = nullptr;
    int n = 0, np = 0;
    double *K = nullptr;      // np x np, full symmetric, pad = identity
    double *Sig = nullptr;    // np x np, full symmetric
    double *Sig2 = nullptr;   // np x np: the NEXT Sigma, built under the site loop by the streamed refactorisation (allocated on first use)
    double *L = nullptr;      // (2 np) x np, ld = ldl = 2 np: rows [0, np) the lower factor of B = I + S^1/2 K S^1/2, rows [np, 2 np) ride through
                              //   the factorisation and come out as Vt = (K S^1/2) L^-T  (V = L \ (S^1/2 K), :59)
    int ldl = 0;
    double *dinv = nullptr;   // np x 16
    double *S = nullptr;      // np x 128 panel of delayed columns
    double *Sc = nullptr;     // np x EP_OUTER: the delayed columns scaled by c, one 128-column block per site block of the current outer panel
    double *blk = nullptr;    // 2 x (128 x 128 unit-lower block factor + its 8 tile inverses), by block parity
    double *vec = nullptr;    // 10 x np: tau, nu, tau_old, nu_old, mu, cav_tau, cav_nu, st, tmp1, tmp2
    double *cvec = nullptr;   // 2 x (128 c + 128 coef), by block parity
    std::vector<hipEvent_t> ev;   // 4 per block: block factor ready | next block's rows solved | side-stream update done | Vt block column final
    hipEvent_t ev_chol = nullptr, ev_parta = nullptr, ev_partb = nullptr;   // end-of-sweep refactorisation: see ep_refactor
    hipEvent_t ev_w = nullptr, ev_pipe = nullptr;   // streamed refactorisation: sweep start on the main stream | its last launch
    int *flags = nullptr;         // np/128 device flags (+ 1 error word): "solved rows of block b are in memory" (ep_block2_kernel -> side stream)
    int epoch = 0;                // token of the current sweep's flags; never reset, so a stale flag cannot match
    int *uflags = nullptr;        // np/128 counters: urgent tiles of block b's trailing update stored (two per sweep; gpk_gemm_nt -> ep_block2_kernel)
    int uepoch = 0;               // sweeps that counted so far: block b's counter stands at 2 * uepoch when its tiles of this sweep are in
    bool owns = true;             // false: a view into an ep_slab (problem g of a lockstep batch); the slab's first problem owns the memory
    bool side_pending = false;    // the side stream still owes the second part of Sigma / mu
    bool sig_mirrored = false;    // the strict upper triangle of Sig mirrors the lower one (only gp_ep_get needs it)
    int *y = nullptr;
    int sweeps = 0;
    double *tau() { return vec; }
    double *nu() { return vec + np; }
    double *tau_old() { return vec + 2 * (size_t)np; }
    double *nu_old() { return vec + 3 * (size_t)np; }
    double *mu() { return vec + 4 * (size_t)np; }
    double *cav_tau() { return vec + 5 * (size_t)np; }
    double *cav_nu() { return vec + 6 * (size_t)np; }
    double *st() { return vec + 7 * (size_t)np; }
    double *tmp1() { return vec + 8 * (size_t)np; }
    double *tmp2() { return vec + 9 * (size_t)np; }
};

namespace {

// 1/x to within an ulp or two: v_rcp_f64 + two Newton steps (the body of the IEEE division sequence without its scaling
// and fix-up instructions; operands here are O(1) variances and precisions, never denormal).  Only used on the serial
// per-site chain of the block kernels, where every dependent instruction is paid 128 times per block.
__device__ __forceinline__ double rcp_nr(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ double rl64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dnorm_d(double x) { return exp(-(x * x) / 2.0 - log(sqrt(2.0 * M_PI))); }   // StatsUtils.scala:15
__device__ __forceinline__ double pnorm_d(double x) { return 0.5 * (1.0 + erf(x / sqrt(2.0))); }              // StatsUtils.scala:17

// A(I-tile, J-tile) -= (S(I-tile, chunk) diag(c)) S(J-tile, chunk)^T for the 16 columns of the chunk starting at column cs0: one
// 16 x 16 tile of the block's delayed update on the matrix cores, operands and result in LDS (leading dimension LS)
__device__ __forceinline__ void ep_chunk_tile(double *A, const double *cs, int LS, int I, int J, int cs0, int fr, int fg) {
    const int ri = 16 * I, rj = 16 * J;
    double4_t acc;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[rr] = A[(ri + fr) + (rj + fg + 4 * rr) * LS];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int qq = cs0 + 4 * ks + fg;
        const double aop = -(cs[qq] * A[(rj + fr) + qq * LS]);
        const double bop = A[(ri + fr) + qq * LS];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) A[(ri + fr) + (rj + fg + 4 * rr) * LS] = acc[rr];
}

#ifdef EP_STAMPS   // lab instrumentation (tools/ep_block2_stamps.py); never defined in the library build
__device__ unsigned long long ep_stamps[4 * GP_NB];
// phase stamps of ep_block2_kernel (tools/ep_block2_stamps.py): slot 2k = s_memrealtime (100 MHz), 2k + 1 = s_memtime (shader clock)
#define EP2_STAMP(k) do { if (i0 == 5 * GP_NB && tid == 0 && blockIdx.x == 0) { ep_stamps[2 * (k)] = __builtin_amdgcn_s_memrealtime(); ep_stamps[2 * (k) + 1] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define EP2_STAMP(k) do { } while (0)
#endif

// All site updates of one block of <= 128 consecutive sites (EpParameterEstimator.scala:44-55), on chip, by ONE
// workgroup.  Inside a block the recurrence only ever touches the block's own rows:
//   s_t[r] = Sigma0[r, i_t] - sum_{q<t} c_q S[r,q] S[i_t,q]      (r in the block)
//   mu[r] += s_t[r] (dnu - c_t (mu_i + dnu s_ii))                 (O(n) form of mu = Sigma nu, Appendix A.3)
// so the 128 x 128 diagonal block of Sigma0, the block of S and mu live in LDS for the whole block.  Outputs: the new
// site/cavity parameters, c, -coef, and Lmat = I + strict_lower(S_blk diag(c)) -- the unit-lower factor with which the
// full-height columns follow afterwards from ONE row-panel solve,  S = Sigma0[:, blk] Lmat^-T  (trsm_panel128).
// ---- the block of sites with the site loop on ONE wave and no barrier per site ----
// (Rounds 1-2 ran the site loop on a scalar lane with two waves of row threads behind it and one barrier per site: ~1950 cycles per
// site of which the chain of dependent fp64 operations is ~1100, the rest barrier and LDS hand-overs; that kernel, ep_block_kernel, was
// removed in round 4.)  Here one wave owns all 128
// rows (lane l: rows l and l + 64) and the values a site needs from "its" row are lane broadcasts (v_readlane), so a site costs
// its chain plus a handful of instructions; everything that does not depend on the site's result -- the part of the next column
// that comes from the chunk's earlier sites -- is independent code in the same instruction stream and fills the issue slots the
// chain leaves empty.  The chunk's 16 finished columns and their c live in registers (statically indexed: the 16 sites of a chunk
// are unrolled), columns also go to LDS for the matrix-core tiles and the unit-lower factor.  Barriers only at chunk boundaries:
// the next chunk's own tile column is updated at once (one tile per wave), the tile columns to the right of it by waves 1-7 while
// wave 0 runs the next chunk.  Outputs (:45-51) are computed for all sites together after the last one.
struct ep_site_regs {
    double S0[16], S1[16];     // this chunk's finished columns, rows lane and lane + 64
    double cq[16];             // c of this chunk's sites
    double m0, m1;             // running mean of rows lane, lane + 64
    double p0, p1;             // column of the site about to be processed, without the term of the site before it
    double sp0, sp1;           // previous site's finished column (rows lane, lane + 64)
    double c_prev, coef_prev;
    double a0, a1;             // column t+1 of A for rows lane, lane + 64, fetched from LDS one site ahead
    double to_n, no_n, yi_n;   // site parameters of the site about to be processed, fetched one site ahead
};

// (o0, o1): LDS row offsets of the lane's two rows.  Sites 0..63 of a block live in rows lane (register set 0), sites 64..127 in rows
// lane + 64: at the block's half-way point the caller SWAPS the two register sets (and passes o0 = 64, o1 = 0), so that the row of the
// site being processed is always in set 0 and ONE instantiation serves both halves -- two made 60 KB of code per kernel, and the
// first chunk of the second half paid an instruction-cache miss of ~4 us (profiles/r03_i_block2_stamps.txt)
template <int P, bool FULL>
__device__ __forceinline__ void ep_site_steps(ep_site_regs &st, double *A, double *cs, double *cf, double *ob, const double *tb,
                                              const double *nb, const double *yb, int cs0, int bsz, int lane, int o0, int o1) {
    constexpr int LS = GP_NB + 1;
    if constexpr (P < 16) {
        const int t = cs0 + P;
        if (FULL || t < bsz) {     // FULL: all 16 sites of the chunk exist -- no branches, one basic block per chunk
            const int own = t & 63;
            // What the chain needs from row t, as lane broadcasts of values that were final BEFORE site t-1's result: its column entry
            // without site t-1's term (Pv), S[t, t-1] (Bv) and its mean without site t-1's update (Mv) -- two operations then
            // separate (c, coef) of site t-1 from the start of this chain: s_ii = Pv - c Bv^2, mu_i = Mv + Bv coef.
            const double Pv = rl64(st.p0, own);
            const double Bv = rl64(st.sp0, own);          // (first site of the block: sp = 0)
            const double Mv = rl64(st.m0, own);
            const double sii = (P > 0) ? fma(-(st.c_prev * Bv), Bv, Pv) : Pv;    // chunk start: the boundary tiles applied site t-1 already
            const double mui = fma(Bv, st.coef_prev, Mv);
            // column t becomes final for every row (the term of site t-1 was the only one missing), the mean takes site t-1's update
            double s0 = st.p0, s1 = st.p1;
            if constexpr (P > 0) {
                const double w = st.c_prev * Bv;                        // c_{t-1} S[t, t-1]
                s0 = fma(-w, st.S0[P - 1], s0);
                s1 = fma(-w, st.S1[P - 1], s1);
            }
            st.m0 = fma(st.sp0, st.coef_prev, st.m0);      // (site 0 of the block: sp = coef_prev = 0)
            st.m1 = fma(st.sp1, st.coef_prev, st.m1);
            st.S0[P] = s0, st.S1[P] = s1, st.sp0 = s0, st.sp1 = s1;
            A[lane + o0 + t * LS] = s0;
            A[lane + o1 + t * LS] = s1;
            const double to = st.to_n, no = st.no_n, yi = st.yi_n;
            // column t+1 without the term of site t: independent of the chain below
            if constexpr (P < 15) {
                if (FULL || t + 1 < bsz) {
                    const int own1 = (t + 1) & 63;
                    double q0 = st.a0, q1 = st.a1;
                    // (LDS reads one site ahead of their use: column t+2, parameters of site t+1; clamped inside the arrays)
                    const int t2 = (P < 14 && t + 2 < GP_NB) ? t + 2 : t + 1;
                    st.a0 = A[lane + o0 + t2 * LS], st.a1 = A[lane + o1 + t2 * LS];
                    st.to_n = tb[t + 1], st.no_n = nb[t + 1], st.yi_n = yb[t + 1];
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        const double wq = st.cq[q] * rl64(st.S0[q], own1);
                        q0 = fma(-wq, st.S0[q], q0);
                        q1 = fma(-wq, st.S1[q], q1);
                    }
                    st.p0 = q0, st.p1 = q1;
                }
            }
            // The serial chain carries the tilted moments only.  With sigma^2 = sii, the new marginal variance sg and mean mi:
            //   c = 1/(1/dtau + sii) = (sii - sg)/sii^2,   coef = dnu - c (mui + dnu sii) = (mi - mui)/sii
            // (substitute dtau = 1/sg - 1/sii, dnu = mi/sg - mui/sii): no reciprocal of sg, of dtau or of 1 + dtau sii on
            // the chain, and the cavity variance 1/(1/sii - to) = sii/(1 - to sii) is ONE reciprocal deep instead of two.
            // (The reference's IEEE edge cases come out the same: dtau = 0 gives sg = sii, c = 0; sii^2 = inf gives c = NaN where the
            // reference has 1/(1/dtau + sii) = 1/0 = inf and a covariance of NaN from there -- tests/test_gpu_ep_edge_cases.py.)
            const double rs = rcp_nr(sii);
            const double cvr = sii * rcp_nr(fma(-to, sii, 1.0));
            const double nc = fma(mui, rs, -no);
            const double cm = nc * cvr;
            const double rt = rsqrt(1.0 + cvr);
            const double z = (yi * cm) * rt;
            // (a hand-written erf / exp pair -- Cody's rational approximations by Estrin's scheme, one exponential for both functions,
            // scalar range branches -- was measured here: 1690 cycles per site against 1370 with the device library's, and four
            // times the code; profiles/r03_i_block2_stamps.txt)
            const double Phi = 0.5 * (1.0 + erf(z * 0.70710678118654752440));
            const double ratio = dnorm_d(z) * rcp_nr(Phi);
            const double mi_hat = cm + (yi * cvr) * (ratio * rt);
            const double sg_hat = cvr - (cvr * cvr) * (ratio * (z + ratio)) * (rt * rt);
            const double c = (sii - sg_hat) * (rs * rs);
            const double coef = (mi_hat - mui) * rs;
            st.cq[P] = c, st.c_prev = c, st.coef_prev = coef;
            // (every lane holds the same values and writes them to the same addresses: no branch for one lane's sake)
            cs[t] = c;
            cf[t] = coef;
            double *o = ob + 5 * t;
            o[0] = rs, o[1] = to, o[2] = sg_hat, o[3] = mi_hat, o[4] = nc;
        }
        ep_site_steps<P + 1, FULL>(st, A, cs, cf, ob, tb, nb, yb, cs0, bsz, lane, o0, o1);
    }
}

constexpr int EP_BLOCK1_LDS = (GP_NB * (GP_NB + 1) + 8 * GP_NB + 5 * GP_NB + 16) * (int)sizeof(double);
constexpr int EP_BLOCK1_WAVES = 4;   // wave 0: the site loop (it needs > 256 registers: four waves leave it 512); waves 1-3: tiles
__global__ __launch_bounds__(64 * EP_BLOCK1_WAVES) void ep_block1_kernel(int n, int np, int i0, int bsz, const double *__restrict__ Sig0,
                                                                        const double *__restrict__ mu, const int *__restrict__ y,
                                                                        double *__restrict__ tau, double *__restrict__ nu,
                                                                        double *__restrict__ cav_tau, double *__restrict__ cav_nu,
                                                                        double *__restrict__ cvec, double *__restrict__ ncoef,
                                                                        double *__restrict__ Lmat, double *__restrict__ Ldinv) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int LS = GP_NB + 1;
    double *A = sm;                   // column c: Sigma0[blk, i0+c] until site c is processed, afterwards s_c[blk]
    double *cs = sm + GP_NB * LS;     // c
    double *cf = cs + GP_NB;          // coef
    double *mb = cf + GP_NB;          // mu restricted to the block (staging only)
    double *tb = mb + GP_NB;          // site parameters and labels, staged once; tb / nb receive the new ones at the end
    double *nb = tb + GP_NB;
    double *yb = nb + GP_NB;
    double *ctb = yb + GP_NB;
    double *cnb = ctb + GP_NB;
    double *ob = cnb + GP_NB;         // [site][5]: 1/sii, tau_old, sg, mi, cavity nu
    const int tid = threadIdx.x, r = tid, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const bool rowthread = tid < GP_NB;
    {   // the block of Sigma0: every thread half of a row's columns, 16 loads in flight
        const int rr = tid & (GP_NB - 1), cq0 = 64 * (tid >> 7);
#pragma unroll
        for (int c0 = 0; c0 < 64; c0 += 16) {
            double v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = Sig0[(i0 + rr) + (size_t)(i0 + cq0 + c0 + c) * np];
#pragma unroll
            for (int c = 0; c < 16; ++c) A[rr + (cq0 + c0 + c) * LS] = v[c];
        }
    }
    if (rowthread) {
        mb[r] = (i0 + r < n) ? mu[i0 + r] : 0.0;
        cs[r] = 0.0;
        cf[r] = 0.0;
        const bool live = r < bsz;
        tb[r] = live ? tau[i0 + r] : 0.0;
        nb[r] = live ? nu[i0 + r] : 0.0;
        yb[r] = live ? (double)y[i0 + r] : 0.0;
    }
    __syncthreads();
    const int jtiles = (bsz + 15) >> 4;
    ep_site_regs st;
    if (wave == 0) {
        st.m0 = mb[lane], st.m1 = mb[lane + 64];
        st.sp0 = st.sp1 = 0.0, st.c_prev = st.coef_prev = 0.0;
    }
    for (int ch = 0; ch < jtiles; ++ch) {
        const int cs0 = 16 * ch;
        if (wave == 0) {
            if (cs0 == 64) {                     // second half of the block: the site rows are the lanes' second rows from here on
                double t_;
                t_ = st.m0, st.m0 = st.m1, st.m1 = t_;
                t_ = st.sp0, st.sp0 = st.sp1, st.sp1 = t_;
            }
            const int o0 = cs0 < 64 ? 0 : 64, o1 = 64 - o0;
            st.p0 = A[lane + o0 + cs0 * LS];     // the chunk's first column is final as it stands (the boundary tiles applied every earlier site)
            st.p1 = A[lane + o1 + cs0 * LS];
            const int c1 = cs0 + 1 < GP_NB ? cs0 + 1 : cs0;
            st.a0 = A[lane + o0 + c1 * LS], st.a1 = A[lane + o1 + c1 * LS];
            st.to_n = tb[cs0], st.no_n = nb[cs0], st.yi_n = yb[cs0];
            if (cs0 + 16 <= bsz) ep_site_steps<0, true>(st, A, cs, cf, ob, tb, nb, yb, cs0, bsz, lane, o0, o1);
            else ep_site_steps<0, false>(st, A, cs, cf, ob, tb, nb, yb, cs0, bsz, lane, o0, o1);
        } else if (ch > 0) {
            // the chunk closed at the last boundary, applied to the tile columns right of the current chunk's: J = ch+1 .. jtiles-1, I = J .. 7
            int k = wave - 1;
            for (int J = ch + 1; J < jtiles; ++J)
                for (int I = J; I < 8; ++I, k = (k == 0 ? EP_BLOCK1_WAVES - 2 : k - 1))
                    if (k == 0) ep_chunk_tile(A, cs, LS, I, J, cs0 - 16, fr, fg);
        }
        __syncthreads();
        if (ch + 1 < jtiles) {
            // the finished chunk applied to the next chunk's own tile column
            const int jc = ch + 1;
            for (int I = jc + wave; I < 8; I += EP_BLOCK1_WAVES) ep_chunk_tile(A, cs, LS, I, jc, cs0, fr, fg);
            __syncthreads();
        }
    }
    // outputs of all sites (:45-51 as written), one thread per site
    if (rowthread && r < bsz) {
        const double *o = ob + 5 * r;
        const double tc = o[0] - o[1];                              // cavity tau  :45
        const double isg = rcp_nr(o[2]);
        const double dtau = isg - tc - o[1];                        // :49
        tau[i0 + r] = o[1] + dtau;                                  // :50
        nu[i0 + r] = o[3] * isg - o[4];                             // :51
        cav_tau[i0 + r] = tc;
        cav_nu[i0 + r] = o[4];
        cvec[r] = cs[r];
        ncoef[r] = cf[r];
    }
    if (rowthread) {
        if (r >= bsz) { cvec[r] = 0.0; ncoef[r] = 0.0; }
        for (int c = 0; c < GP_NB; ++c)
            Lmat[r + (size_t)c * GP_NB] = (r == c) ? 1.0 : ((r > c && c < bsz) ? A[r + c * LS] * cs[c] : 0.0);
    }
    if (wave == 2 || wave == 3) {   // unit-lower tile inverses of Lmat
        const int c0 = 16 * (4 * (wave - 2) + fg);
        double row[16], sv[16], x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            row[k] = (c0 + k < bsz) ? A[(c0 + fr) + (c0 + k) * LS] * cs[c0 + k] : 0.0;
            sv[k] = (k == fr) ? 1.0 : 0.0;
        }
        tile_unit_inverse<0>(row, sv, x);
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) Ldinv[(c0 / 16) * 256 + rr + 16 * fr] = x[rr];
    }
}

// ---- the fused chain kernel: link to the block before + all site updates of this block, ONE launch per block ----
// Between two block kernels the sweep used to run a second single-workgroup kernel (ep_link_kernel: the 128 rows of the previous
// block's delayed columns that belong to this block, X = Sigma0[blk b, blk b-1] Lmat^-T, and the diagonal tile
// D = Sigma0[blk b, blk b] -= X diag(c) X^T this block's site loop starts from).  Everything the link produces for the CHAIN stays
// on one CU, so it is this kernel's prologue: the X strip is solved in the LDS the site loop has not started to use yet, the 36
// lower tiles of D are updated in registers and dropped straight into the site loop's LDS image (they never go back to global
// memory: Sigma[blk b, blk b] is dead after this block), the mean update lands in the staged mean.  One launch, one launch gap,
// the D round trip through L2 (66 KB out, 128 KB in) and the block kernel's own load of A are gone from the chain.
// What OTHER streams need from the link -- the solved rows X (in place in Sigma's dead panel: the other operand of the
// column-panel update below this block's diagonal tile) -- is stored from the prologue while the matrix cores run the tile
// update, and announced by a device flag (release, agent scope) as soon as every wave's stores have been acknowledged: the side
// stream's consumer is preceded by ep_wait_flag_kernel, so it starts ~25 us into this kernel instead of after its ~100 us.
// blockIdx.x = problem of a lockstep batch (ep_strides; a single run has one).
struct ep_strides {
    size_t sig = 0, vec = 0, y = 0, blk = 0, cvec = 0, sc = 0;   // in elements, per problem
    int flag = 0;
};
constexpr int EP2_XS = 144;   // LDS column stride of the prologue's 128-row strip: 1152 B = 128 (mod 256) -> conflict-free fragments
constexpr int EP_BLOCK2_LDS = (GP_NB * EP2_XS + 8 * GP_NB + 5 * GP_NB + 16) * (int)sizeof(double);   // 160 896 B of the CU's 163 840
constexpr int EP_BLK_ELEMS = GP_NB * GP_NB + 8 * 256;   // Lmat + its tile inverses

// Everything of chunk j (sites 16 j .. 16 j + 15) that other kernels read, written by the three helper waves while wave 0 runs
// the next chunk: the chunk's columns of the unit-lower factor Lmat = I + strict_lower(S diag(c)) (final as soon as its sites are:
// column q of S is complete for every row of the block once site q has been processed), the inverse of its 16 x 16 diagonal tile,
// and the reference's outputs for its sites (:45-51 as written).  After the last chunk only that chunk's share is left -- the
// all-at-the-end form was 9.4 us of every block's ~137 (profiles/r03_i_block2_stamps.txt).
__device__ __forceinline__ void ep_chunk_outputs(int j, int bsz, int i0, const double *A, const double *cs, const double *ob, double *tau,
                                                 double *nu, double *cav_tau, double *cav_nu, double *Lmat, double *Ldinv, int hid, int nh,
                                                 int fr, bool inv_wave) {
    constexpr int LS = GP_NB + 1;
    const int c0 = 16 * j;
    for (int e = hid; e < 16 * GP_NB; e += nh) {
        const int r = e & (GP_NB - 1), c = c0 + (e >> 7);
        Lmat[r + (size_t)c * GP_NB] = (r == c) ? 1.0 : ((r > c && c < bsz) ? A[r + c * LS] * cs[c] : 0.0);
    }
    if (hid < 16 && c0 + hid < bsz) {
        const int r = c0 + hid;
        const double *o = ob + 5 * r;
        const double tc = o[0] - o[1];                              // cavity tau  :45
        const double isg = rcp_nr(o[2]);
        const double dtau = isg - tc - o[1];                        // :49
        tau[i0 + r] = o[1] + dtau;                                  // :50
        nu[i0 + r] = o[3] * isg - o[4];                             // :51
        cav_tau[i0 + r] = tc;
        cav_nu[i0 + r] = o[4];
    }
    if (inv_wave) {   // one wave: the tile in each of its four 16-lane rows (DPP row broadcasts, dpp_tile.h); the first row stores
        double row[16], sv[16], x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            row[k] = (c0 + k < bsz) ? A[(c0 + fr) + (c0 + k) * LS] * cs[c0 + k] : 0.0;
            sv[k] = (k == fr) ? 1.0 : 0.0;
        }
        tile_unit_inverse<0>(row, sv, x);
        if ((hid & 63) < 16) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) Ldinv[j * 256 + rr + 16 * fr] = x[rr];
        }
    }
}

template <bool PRO>
__global__ __launch_bounds__(64 * EP_BLOCK1_WAVES) void ep_block2_kernel(int n, int np, int i0, int bsz, double *__restrict__ Sig,
                                                                        double *__restrict__ vec, const int *__restrict__ y,
                                                                        double *__restrict__ blk, double *__restrict__ cvbase,
                                                                        double *__restrict__ Scbase, int *__restrict__ flags, int par,
                                                                        int token, ep_strides es, int scoff, const int *__restrict__ uwait,
                                                                        int utarget, int *__restrict__ uerr) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    constexpr int LS = GP_NB + 1;
    Sig += (size_t)blockIdx.x * es.sig, vec += (size_t)blockIdx.x * es.vec, y += (size_t)blockIdx.x * es.y;
    blk += (size_t)blockIdx.x * es.blk, cvbase += (size_t)blockIdx.x * es.cvec, Scbase += (size_t)blockIdx.x * es.sc;
    flags += (size_t)blockIdx.x * es.flag;
    double *tau = vec, *nu = vec + np, *cav_tau = vec + 5 * (size_t)np, *cav_nu = vec + 6 * (size_t)np;
    const double *mu = vec + 4 * (size_t)np;
    double *Lmat = blk + (size_t)par * EP_BLK_ELEMS, *Ldinv = Lmat + GP_NB * GP_NB;
    double *cvec = cvbase + (size_t)par * 2 * GP_NB, *ncoef = cvec + GP_NB;
    double *A = sm;                          // column c: Sigma0[blk, i0+c] until site c is processed, afterwards s_c[blk]
    double *xs = sm;                         // prologue only: the X strip, 128 x EP2_XS (overlaps A: A is written after the strip is dead)
    double *cs = sm + GP_NB * EP2_XS;        // c
    double *cf = cs + GP_NB;                 // coef
    double *mb = cf + GP_NB;                 // mu restricted to the block (staging only)
    double *tb = mb + GP_NB;                 // site parameters and labels, staged once
    double *nb = tb + GP_NB;
    double *yb = nb + GP_NB;
    double *pcs = yb + GP_NB;                // prologue: c and coef of the block before
    double *pcf = pcs + GP_NB;
    double *ob = pcf + GP_NB;                // [site][5]: 1/sii, tau_old, sg, mi, cavity nu
    const int tid = threadIdx.x, r = tid, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
    const bool rowthread = tid < GP_NB;
    EP2_STAMP(0);
    if (rowthread) {
        cs[r] = 0.0;
        cf[r] = 0.0;
        const bool live = r < bsz;
        tb[r] = live ? tau[i0 + r] : 0.0;
        nb[r] = live ? nu[i0 + r] : 0.0;
        yb[r] = live ? (double)y[i0 + r] : 0.0;
    }
    if constexpr (PRO) {
        if (uwait) {
            // The two tiles of Sigma this prologue reads were updated by the side stream's trailing update of block b-2, which
            // announces them with a counter as soon as they are stored (gpk_gemm_nt's urgent tiles) -- this kernel was launched
            // without waiting for that whole update.  Bounded: the update was enqueued before this kernel and depends on nothing
            // after it; ~0.3 s without the count means something else is wrong, recorded in *uerr (the sweep returns GP_EHIP).
            if (tid == 0) {
                int it = 0;
                while (__hip_atomic_load(uwait, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < utarget) {
                    if (__hip_atomic_load(uerr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                    if (++it > 1000000) { atomicExch(uerr, 1000 + i0 / GP_NB); break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
            __syncthreads();
        }
        const double *pL = blk + (size_t)(1 - par) * EP_BLK_ELEMS, *pdinv = pL + GP_NB * GP_NB;     // the block before: unit-lower factor,
        const double *pc = cvbase + (size_t)(1 - par) * 2 * GP_NB;                                  //   its tile inverses, c and coef
        double *X = Sig + i0 + (size_t)(i0 - GP_NB) * np;      // rows of this block, columns of the block before (solved in place)
        double *X2 = Scbase + scoff + i0;                      // X diag(c), leading dimension np, in the column block of the block before
        const double *D = Sig + i0 + (size_t)i0 * np;
        {   // strip by LDS-DMA: one wave instruction = one 128-row column (1 KiB), 32 per wave, all in flight
            const double *src = X + lane * 2 + (size_t)wave * np;
#pragma unroll
            for (int q = 0; q < GP_NB / 4; ++q) __builtin_amdgcn_global_load_lds(src + (size_t)(4 * q) * np, xs + (wave + 4 * q) * EP2_XS, 16, 0, 0);
        }
        if (rowthread) pcs[r] = pc[r], pcf[r] = pc[GP_NB + r];
        // this wave's nine lower tiles of D (q = wave, wave + 4, ...): fetched under the solve
        double4_t dacc[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            int I, J;
            tri_coords(wave + 4 * u, I, J);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) dacc[u][rr] = D[(16 * I + fr) + (size_t)(16 * J + fg + 4 * rr) * np];
        }
        // ALL fragments of the unit-lower factor and of its tile inverses up front (112 + 32 doubles per lane): the factor was
        // written by the kernel before on another CU, so each fetch is an L2 miss served over the fabric; fetched one chunk ahead,
        // as the stand-alone link kernel does, those latencies sat between the eight short chunk steps
        double f1[28], f2[28], f3[28], f4[28], f5[28], f6[28], f7[28], dq[8][4];
        trsm_load_frags<1>(pL, GP_NB, fr, fg, f1);
        trsm_load_frags<2>(pL, GP_NB, fr, fg, f2);
        trsm_load_frags<3>(pL, GP_NB, fr, fg, f3);
        trsm_load_frags<4>(pL, GP_NB, fr, fg, f4);
        trsm_load_frags<5>(pL, GP_NB, fr, fg, f5);
        trsm_load_frags<6>(pL, GP_NB, fr, fg, f6);
        trsm_load_frags<7>(pL, GP_NB, fr, fg, f7);
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) dq[cb][rr] = pdinv[cb * 256 + fr + 16 * (fg + 4 * rr)];
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMA has landed (and the loads above)
        __syncthreads();
        EP2_STAMP(1);
        // X <- X Lmat^-T: rows are independent, every wave solves its own 32 (two 16-row sub-strips share the L fragments)
        const int sp0 = wave * 32 + fr, sp1 = sp0 + 16;
#define EP2_CHUNK(CB, F) trsm_chunk_pre2<CB, EP2_XS>(xs, sp0, sp1, fr, fg, F, dq[CB])
        EP2_CHUNK(0, f1);
        EP2_CHUNK(1, f1);
        EP2_CHUNK(2, f2);
        EP2_CHUNK(3, f3);
        EP2_CHUNK(4, f4);
        EP2_CHUNK(5, f5);
        EP2_CHUNK(6, f6);
        EP2_CHUNK(7, f7);
#undef EP2_CHUNK
        EP2_STAMP(2);
        __syncthreads();
        EP2_STAMP(3);
        {   // the solved rows and their scaled copy go out now (operands of the side stream's trailing update, which covers this
            // block's column panel too): the stores drain while the matrix cores run the tile update
            const int li = lane * 2, lc = wave;
#pragma unroll 8
            for (int q = 0; q < GP_NB / 4; ++q) {
                const int c = lc + 4 * q;
                double2_t v = *reinterpret_cast<const double2_t *>(xs + c * EP2_XS + li);
                *reinterpret_cast<double2_t *>(X + li + (size_t)c * np) = v;
                const double sc = pcs[c];
                v.x *= sc, v.y *= sc;
                *reinterpret_cast<double2_t *>(X2 + li + (size_t)c * np) = v;
            }
        }
        if (rowthread) {   // mean of this block's rows: + X coef of the block before (the O(n) form of mu = Sigma nu, Appendix A.3)
            double acc = 0.0;      // (one accumulator, columns ascending: the link kernel's order, bit for bit)
#pragma unroll 16
            for (int c = 0; c < GP_NB; ++c) acc = fma(xs[c * EP2_XS + r], pcf[c], acc);
            mb[r] = ((i0 + r < n) ? mu[i0 + r] : 0.0) + acc;
        }
        EP2_STAMP(4);
        {   // D(lower) -= X diag(c) X^T: nine tiles per wave, K = 128, operands from the strip; c of this lane's k values once
            double ck[32];
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) ck[ks] = -pcs[4 * ks + fg];
            // Tiles in pairs (a triple at the end): independent accumulator chains, and the LDS operands of the next two k steps
            // requested before the MFMAs of the current two are issued (explicit double buffering between sched_barriers; the
            // compiler's own order -- read, wait out the LDS latency, multiply -- ran at 112 cycles per 64-cycle MFMA).  Every
            // tile's sum stays in k order: the same bits as the stand-alone link kernel.
#define EP2_LD(buf, T, ks) do { const int k_ = 4 * (ks) + fg; \
            _Pragma("unroll") for (int t_ = 0; t_ < T; ++t_) { buf[2 * t_] = xs[k_ * EP2_XS + 16 * Jt[t_] + fr]; buf[2 * t_ + 1] = xs[k_ * EP2_XS + 16 * It[t_] + fr]; } } while (0)
#pragma unroll
            for (int u = 0; u < 6; u += 2) {
                int It[2], Jt[2];
                tri_coords(wave + 4 * u, It[0], Jt[0]);
                tri_coords(wave + 4 * (u + 1), It[1], Jt[1]);
                double4_t acc0 = dacc[u], acc1 = dacc[u + 1];
                double c0[4], c1[4], n0[4], n1[4];
                EP2_LD(c0, 2, 0);
                EP2_LD(c1, 2, 1);
#pragma unroll
                for (int ks = 0; ks < 32; ks += 2) {
                    if (ks + 2 < 32) { EP2_LD(n0, 2, ks + 2); EP2_LD(n1, 2, ks + 3); }
                    __builtin_amdgcn_sched_barrier(0);
                    acc0 = MFMA(ck[ks] * c0[0], c0[1], acc0);
                    acc1 = MFMA(ck[ks] * c0[2], c0[3], acc1);
                    acc0 = MFMA(ck[ks + 1] * c1[0], c1[1], acc0);
                    acc1 = MFMA(ck[ks + 1] * c1[2], c1[3], acc1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) c0[i] = n0[i], c1[i] = n1[i];
                }
                dacc[u] = acc0, dacc[u + 1] = acc1;
                if (u == 0) {
                    // The solved rows have had a tile pair's worth of time (~3 us) to drain: announce them now rather than after
                    // the whole update -- the side stream's trailing update of the block before starts ~10 us earlier, and it is
                    // as much on the sweep's critical path as this kernel (profiles/r03_e_sweep_fused.txt)
                    __builtin_amdgcn_s_waitcnt(0x0F70);   // this wave's stores of X / X2 are acknowledged
                    __syncthreads();
                    if (tid == 64) {
                        __threadfence();
                        __hip_atomic_store(flags + i0 / GP_NB, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            {
                int It[3], Jt[3];
                tri_coords(wave + 24, It[0], Jt[0]);
                tri_coords(wave + 28, It[1], Jt[1]);
                tri_coords(wave + 32, It[2], Jt[2]);
                double4_t acc0 = dacc[6], acc1 = dacc[7], acc2 = dacc[8];
                double c0[6], c1[6], n0[6], n1[6];
                EP2_LD(c0, 3, 0);
                EP2_LD(c1, 3, 1);
#pragma unroll
                for (int ks = 0; ks < 32; ks += 2) {
                    if (ks + 2 < 32) { EP2_LD(n0, 3, ks + 2); EP2_LD(n1, 3, ks + 3); }
                    __builtin_amdgcn_sched_barrier(0);
                    acc0 = MFMA(ck[ks] * c0[0], c0[1], acc0);
                    acc1 = MFMA(ck[ks] * c0[2], c0[3], acc1);
                    acc2 = MFMA(ck[ks] * c0[4], c0[5], acc2);
                    acc0 = MFMA(ck[ks + 1] * c1[0], c1[1], acc0);
                    acc1 = MFMA(ck[ks + 1] * c1[2], c1[3], acc1);
                    acc2 = MFMA(ck[ks + 1] * c1[4], c1[5], acc2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 6; ++i) c0[i] = n0[i], c1[i] = n1[i];
                }
                dacc[6] = acc0, dacc[7] = acc1, dacc[8] = acc2;
            }
#undef EP2_LD
        }
        EP2_STAMP(5);
        EP2_STAMP(6);
        __syncthreads();                      // nobody reads the strip any more
        EP2_STAMP(7);
        // the site loop's image of the block: lower tiles from the accumulators.  (The strictly upper tiles keep whatever the strip
        // left there: the site loop carries them through the lanes that own rows above the diagonal and never reads them back.)
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            int I, J;
            tri_coords(wave + 4 * u, I, J);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) A[(16 * I + fr) + (16 * J + fg + 4 * rr) * LS] = dacc[u][rr];
        }
    } else {
        // first block of a sweep: the block of Sigma0 as it stands, every thread half of a row's columns, 16 loads in flight
        const int rr = tid & (GP_NB - 1), cq0 = 64 * (tid >> 7);
#pragma unroll
        for (int c0 = 0; c0 < 64; c0 += 16) {
            double v[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] = Sig[(i0 + rr) + (size_t)(i0 + cq0 + c0 + c) * np];
#pragma unroll
            for (int c = 0; c < 16; ++c) A[rr + (cq0 + c0 + c) * LS] = v[c];
        }
        if (rowthread) mb[r] = (i0 + r < n) ? mu[i0 + r] : 0.0;
    }
    __syncthreads();
    EP2_STAMP(8);
    const int jtiles = (bsz + 15) >> 4;
    const int hid = tid - 64;                // helper thread index (waves 1-3)
    ep_site_regs st;
    if (wave == 0) {
        st.m0 = mb[lane], st.m1 = mb[lane + 64];
        st.sp0 = st.sp1 = 0.0, st.c_prev = st.coef_prev = 0.0;
    }
    for (int ch = 0; ch < jtiles; ++ch) {
        const int cs0 = 16 * ch;
        if (wave == 0) {
            if (cs0 == 64) {                     // second half of the block: the site rows are the lanes' second rows from here on
                double t_;
                t_ = st.m0, st.m0 = st.m1, st.m1 = t_;
                t_ = st.sp0, st.sp0 = st.sp1, st.sp1 = t_;
            }
            const int o0 = cs0 < 64 ? 0 : 64, o1 = 64 - o0;
            st.p0 = A[lane + o0 + cs0 * LS];     // the chunk's first column is final as it stands (the boundary tiles applied every earlier site)
            st.p1 = A[lane + o1 + cs0 * LS];
            const int c1 = cs0 + 1 < GP_NB ? cs0 + 1 : cs0;
            st.a0 = A[lane + o0 + c1 * LS], st.a1 = A[lane + o1 + c1 * LS];
            st.to_n = tb[cs0], st.no_n = nb[cs0], st.yi_n = yb[cs0];
            if (cs0 + 16 <= bsz) ep_site_steps<0, true>(st, A, cs, cf, ob, tb, nb, yb, cs0, bsz, lane, o0, o1);
            else ep_site_steps<0, false>(st, A, cs, cf, ob, tb, nb, yb, cs0, bsz, lane, o0, o1);
            EP2_STAMP(9 + ch);
        } else if (ch > 0) {
            // the chunk closed at the last boundary, applied to the tile columns right of the current chunk's: J = ch+1 .. jtiles-1, I = J .. 7
            int k = wave - 1;
            for (int J = ch + 1; J < jtiles; ++J)
                for (int I = J; I < 8; ++I, k = (k == 0 ? EP_BLOCK1_WAVES - 2 : k - 1))
                    if (k == 0) ep_chunk_tile(A, cs, LS, I, J, cs0 - 16, fr, fg);
            // ... and everything other kernels read of that chunk
            ep_chunk_outputs(ch - 1, bsz, i0, A, cs, ob, tau, nu, cav_tau, cav_nu, Lmat, Ldinv, hid, 64 * (EP_BLOCK1_WAVES - 1), fr, wave == 1);
        }
        __syncthreads();
        if (ch + 1 < jtiles) {
            // the finished chunk applied to the next chunk's own tile column
            const int jc = ch + 1;
            for (int I = jc + wave; I < 8; I += EP_BLOCK1_WAVES) ep_chunk_tile(A, cs, LS, I, jc, cs0, fr, fg);
            __syncthreads();
        }
    }
    EP2_STAMP(17);
    // what is left: the last chunk's share, the identity for sites a ragged block does not have, c and coef of every site
    if (wave > 0) ep_chunk_outputs(jtiles - 1, bsz, i0, A, cs, ob, tau, nu, cav_tau, cav_nu, Lmat, Ldinv, hid, 64 * (EP_BLOCK1_WAVES - 1), fr, wave == 1);
    else {
        for (int j = jtiles; j < 8; ++j) ep_chunk_outputs(j, bsz, i0, A, cs, ob, tau, nu, cav_tau, cav_nu, Lmat, Ldinv, lane, 64, fr, true);
        cvec[lane] = lane < bsz ? cs[lane] : 0.0, cvec[lane + 64] = lane + 64 < bsz ? cs[lane + 64] : 0.0;
        ncoef[lane] = lane < bsz ? cf[lane] : 0.0, ncoef[lane + 64] = lane + 64 < bsz ? cf[lane + 64] : 0.0;
    }
    EP2_STAMP(18);
}

// One thread per problem waits for the flag ep_block2_kernel sets when the solved rows of block `b` are in memory.  The wait is
// BOUNDED: the producer was launched before this kernel and depends on nothing that comes after it (gp_ep_sweep), so the flag
// arrives within the producer's first ~25 us; if it has not arrived after ~0.2 s something else is wrong, the kernel gives up,
// records it in *err and the sweep returns GP_EHIP -- a loud failure instead of a hung queue.
__global__ void ep_wait_flag_kernel(const int *__restrict__ flags, int stride, int token, int *__restrict__ err) {
    const int *f = flags + (size_t)blockIdx.x * stride;
    int it = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != token) {
        // once a wait of this sweep has timed out the result is lost anyway: the later ones fall through at once, so that a
        // run under a tool that serialises kernels (rocprofv3 --pmc: producer and waiter never run side by side) ends in
        // seconds with GP_EHIP instead of crawling through one time-out per block
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        if (++it > 200000) { atomicExch(err, 1 + (int)blockIdx.x); break; }
        __builtin_amdgcn_s_sleep(32);
    }
}

// D (128 x 128, lower triangle, leading dimension ldd) -= Sc St^T with Sc, St the 128 x 128 row blocks (leading dimension ld)
// of the scaled and unscaled delayed columns: the ONE tile of a block's rank-128 update that the next block kernel reads.
// It sits on the serial chain of the sweep, where the general 128 x 128-tile GEMM would run it on a single CU (14 us of
// MFMA issue alone); here its 36 lower 16 x 16 tiles go to 36 one-wave workgroups with the operands straight from L2.
__global__ __launch_bounds__(64) void ep_diag_update_kernel(double *__restrict__ D, int ldd, const double *__restrict__ Sc,
                                                            const double *__restrict__ St, int ld) {
    int I = 0;
    const int q = blockIdx.x;
    while ((I + 1) * (I + 2) / 2 <= q) ++I;
    const int J = q - I * (I + 1) / 2;
    const int lane = threadIdx.x, fr = lane & 15, fg = lane >> 4;
    double a[32], b[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        a[ks] = -Sc[(16 * J + fr) + (size_t)(4 * ks + fg) * ld];
        b[ks] = St[(16 * I + fr) + (size_t)(4 * ks + fg) * ld];
    }
    double4_t acc;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) acc[rr] = D[(16 * I + fr) + (size_t)(16 * J + fg + 4 * rr) * ldd];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
        if (16 * I + fr >= 16 * J + fg + 4 * rr) D[(16 * I + fr) + (size_t)(16 * J + fg + 4 * rr) * ldd] = acc[rr];
}

__global__ void scale_cols_kernel(double *__restrict__ dst, int ldd, const double *__restrict__ src, int lds, const double *__restrict__ c,
                                  int rows, int cols) {
    size_t total = (size_t)rows * cols;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int i = (int)(e % rows), j = (int)(e / rows);
        dst[i + (size_t)j * ldd] = src[i + (size_t)j * lds] * c[j];
    }
}

// B = I + (st st^T) o K on the lower triangle (EpParameterEstimator.scala:56-58); pad rows have st = 0
__global__ void ep_bmat_kernel(double *__restrict__ B, int ldb, const double *__restrict__ K, const double *__restrict__ st, int np) {
    for (int j = blockIdx.y; j < np; j += gridDim.y)
        for (int i = j + blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x)
            B[i + (size_t)j * ldb] = (i == j ? 1.0 : 0.0) + (st[i] * st[j]) * K[i + (size_t)j * np];
}

// ---- streamed refactorisation (gp_ep_sweep): the working matrix W = ep->L, (2 np) x np ----
// Start of a sweep: rows [0, np) <- K on and below the diagonal, rows [np, 2 np) <- K.  Nothing here depends on the site
// parameters: with M = K + S^-1 only the DIAGONAL of what is factored depends on them, and B = S^1/2 M S^1/2 means
// chol(B) = S^1/2 chol(M) -- row i of the factor, of the trailing matrix and of nothing else carries the factor s_i.  So the
// right-looking factorisation can run on unscaled rows and columns for every site that has not been visited yet.
__global__ void ep_winit_kernel(double *__restrict__ W, int ldw, const double *__restrict__ K, int np, size_t sW = 0, size_t sK = 0) {
    W += (size_t)blockIdx.z * sW, K += (size_t)blockIdx.z * sK;      // blockIdx.z = problem of a lockstep batch
    for (int j = blockIdx.y; j < np; j += gridDim.y) {
        const double *kj = K + (size_t)j * np;
        double *wj = W + (size_t)j * ldw;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * np; i += gridDim.x * blockDim.x) {
            if (i >= np) wj[i] = kj[i - np];
            else if (i >= j) wj[i] = kj[i];
        }
    }
}
// Site block k0's precisions are final: column k of the block (rows k .. 2 np - 1 of the trailing matrix and of the rows that
// ride along) takes its factor s_k, rows inside the block take s_i as well, the diagonal gains the identity of B; st <- s
// (blockIdx.y < 128: one column each).  The block's rows of the factor computed so far (columns < k0) take s_i (blockIdx.y >= 128:
// two columns per workgroup).  One launch for both parts.
__global__ __launch_bounds__(256) void ep_wscale_kernel(double *__restrict__ W, int ldw, int np, int k0, const double *__restrict__ tau, int n,
                                                        double *__restrict__ st, size_t sW = 0, size_t sVec = 0) {
    W += (size_t)blockIdx.z * sW, tau += (size_t)blockIdx.z * sVec, st += (size_t)blockIdx.z * sVec;   // blockIdx.z = problem
    if (blockIdx.y < GP_NB) {
        const int k = k0 + blockIdx.y;
        const double sk = (k < n) ? sqrt(tau[k]) : 0.0;
        if (blockIdx.x == 0 && threadIdx.x == 0) st[k] = sk;
        double *wk = W + (size_t)k * ldw;
        for (int i = k + blockIdx.x * blockDim.x + threadIdx.x; i < 2 * np; i += gridDim.x * blockDim.x) {
            double f = sk;
            if (i < k0 + GP_NB) f *= (i < n) ? sqrt(tau[i]) : 0.0;
            const double v = wk[i] * f;
            wk[i] = (i == k) ? 1.0 + v : v;
        }
    } else {
        const int i = k0 + (threadIdx.x & (GP_NB - 1));
        const double si = (i < n) ? sqrt(tau[i]) : 0.0;
        const int wg = (blockIdx.y - GP_NB) * gridDim.x + blockIdx.x, nwg = (gridDim.y - GP_NB) * gridDim.x;
        for (int j = wg * 2 + (threadIdx.x >> 7); j < k0; j += nwg * 2) W[i + (size_t)j * ldw] *= si;
    }
}

__global__ void mirror_lower_kernel(double *__restrict__ A, int np) {
    __shared__ double t[64][65];
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bi <= bj) return;   // strictly-lower tiles only; diagonal tiles handled below
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int q = ty; q < 64; q += 4) t[q][tx] = A[(bi * 64 + tx) + (size_t)(bj * 64 + q) * np];
    __syncthreads();
    for (int q = ty; q < 64; q += 4) A[(bj * 64 + tx) + (size_t)(bi * 64 + q) * np] = t[tx][q];
}
__global__ void mirror_diag_kernel(double *__restrict__ A, int np) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
        int i = e & 63, j = e >> 6;
        if (i > j) A[(b * 64 + j) + (size_t)(b * 64 + i) * np] = A[(b * 64 + i) + (size_t)(b * 64 + j) * np];
    }
}

__global__ void vec_sqrt_kernel(double *__restrict__ out, const double *__restrict__ in, int n, int np) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) out[i] = (i < n) ? sqrt(in[i]) : 0.0;
}
__global__ void vec_div_kernel(double *__restrict__ out, const double *__restrict__ a, const double *__restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] / b[i];
}
__global__ void vec_sub_kernel(double *__restrict__ out, const double *__restrict__ a, const double *__restrict__ b, int n, int np) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) out[i] = (i < n) ? a[i] - b[i] : 0.0;
}
// M(i,j) = st_i * M(i,j) * st_j on the lower triangle
__global__ void scale_sym_lower_kernel(double *__restrict__ M, const double *__restrict__ st, int n, int ld) {
    for (int j = blockIdx.y; j < n; j += gridDim.y)
        for (int i = j + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) M[i + (size_t)j * ld] *= st[i] * st[j];
}
__global__ void vec_mul_kernel(double *__restrict__ out, const double *__restrict__ a, const double *__restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] * b[i];
}
// w = nu - st o z
__global__ void ep_w_kernel(double *__restrict__ w, const double *__restrict__ nu, const double *__restrict__ st, const double *__restrict__ z, int n, int np) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) w[i] = (i < n) ? nu[i] - st[i] * z[i] : 0.0;
}
__global__ void ep_prob_kernel(double *__restrict__ prob, const double *__restrict__ fmean, const double *__restrict__ sumsq,
                               const double *__restrict__ kss, int m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) prob[i] = pnorm_d(fmean[i] / sqrt(1.0 + (kss[i] - sumsq[i])));   // GpClassifier.scala:43-45
}

// EP log marginal likelihood (EpParameterEstimator.scala:71-96); strict: the fourth/first term is dropped as compiled
__global__ __launch_bounds__(1024) void ep_lml_kernel(int n, int ldl, const double *__restrict__ L, const double *__restrict__ tau,
                                                      const double *__restrict__ nu, const double *__restrict__ mu,
                                                      const double *__restrict__ cav_tau, const double *__restrict__ cav_nu,
                                                      const int *__restrict__ y, int strict, double *__restrict__ out) {
    __shared__ double red[1024];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double ct = cav_tau[i], cm = cav_nu[i] / ct, T = tau[i] + ct;
        double third = log(pnorm_d(y[i] * cm / sqrt(1.0 + 1.0 / ct)));
        double fourth = strict ? 0.0 : 0.5 * log(1.0 + tau[i] / ct) - log(L[i + (size_t)i * ldl]);
        // 0.5 * [ nu^T (Sigma - diag(1/T)) nu + sum (cm ct / T)(tau cm - 2 nu) ],  nu^T Sigma nu = nu . mu
        double first = nu[i] * mu[i] - nu[i] * nu[i] / T;
        double second = ((cm * ct) * (1.0 / T)) * (tau[i] * cm - nu[i] * 2.0);
        acc += third + fourth + 0.5 * (first + second);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

inline dim3 g1(int n) { return dim3((n + 255) / 256); }

// out[i] = sum_j A(i,j) x[j] for rows i in [lo, lo + m) of a SYMMETRIC n x n matrix of which only the lower triangle is stored
// (mu = Sigma nu, EpParameterEstimator.scala:61, without mirroring Sigma first), in three small launches with a fixed
// summation order:  row part  sum_{j <= i} A(i,j) x[j]  (thread = row, coalesced over i, the j range cut into SYMV_CHUNKS
// chunks -> partial[chunk][i]);  column part  sum_{j > i} A(j,i) x[j]  (one wave per column, lanes striding the rows ->
// partial[SYMV_CHUNKS][i]);  then the chunks are added in order.
constexpr int SYMV_CHUNKS = 16;
__global__ __launch_bounds__(256) void ep_symv_rows_kernel(const double *__restrict__ A, int ld, int n, const double *__restrict__ x,
                                                           double *__restrict__ partial, int lo, int m, size_t sA = 0, size_t sx = 0, size_t sp = 0) {
    A += (size_t)blockIdx.z * sA, x += (size_t)blockIdx.z * sx, partial += (size_t)blockIdx.z * sp;   // blockIdx.z = problem
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= m) return;
    const int i = lo + r, per = (n + SYMV_CHUNKS - 1) / SYMV_CHUNKS;
    const int j0 = blockIdx.y * per, j1 = min(min(n, j0 + per), i + 1);
    double acc = 0.0;
#pragma unroll 8
    for (int j = j0; j < j1; ++j) acc = fma(A[i + (size_t)j * ld], x[j], acc);
    partial[(size_t)blockIdx.y * n + i] = acc;
}
__global__ __launch_bounds__(256) void ep_symv_cols_kernel(const double *__restrict__ A, int ld, int n, const double *__restrict__ x,
                                                           double *__restrict__ partial, int lo, int m, size_t sA = 0, size_t sx = 0, size_t sp = 0) {
    A += (size_t)blockIdx.z * sA, x += (size_t)blockIdx.z * sx, partial += (size_t)blockIdx.z * sp;
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= m) return;
    const int c = lo + r;
    double p0 = 0.0, p1 = 0.0;
    int j = c + 1 + lane;
    for (; j + 64 < n; j += 128) {
        p0 = fma(A[j + (size_t)c * ld], x[j], p0);
        p1 = fma(A[j + 64 + (size_t)c * ld], x[j + 64], p1);
    }
    if (j < n) p0 = fma(A[j + (size_t)c * ld], x[j], p0);
    double pr = p0 + p1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pr += __shfl_xor(pr, o);
    if (lane == 0) partial[(size_t)SYMV_CHUNKS * n + c] = pr;
}
__global__ void ep_symv_sum_kernel(const double *__restrict__ partial, int n, double *__restrict__ out, int lo, int m, size_t sp = 0, size_t so = 0) {
    partial += (size_t)blockIdx.z * sp, out += (size_t)blockIdx.z * so;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    const int i = lo + r;
    double acc = 0.0;
    for (int c = 0; c <= SYMV_CHUNKS; ++c) acc += partial[(size_t)c * n + i];
    out[i] = acc;
}
// count > 1: a lockstep batch (problem g at A + g sA, x / out + g sx, partial + g sp)
void ep_symv_lower(hipStream_t s, const double *A, int ld, int n, const double *x, double *partial, double *out, int lo, int m, int count = 1,
                   size_t sA = 0, size_t sx = 0, size_t sp = 0) {
    hipLaunchKernelGGL(ep_symv_rows_kernel, dim3((m + 255) / 256, SYMV_CHUNKS, count), dim3(256), 0, s, A, ld, n, x, partial, lo, m, sA, sx, sp);
    hipLaunchKernelGGL(ep_symv_cols_kernel, dim3((m + 3) / 4, 1, count), dim3(256), 0, s, A, ld, n, x, partial, lo, m, sA, sx, sp);
    hipLaunchKernelGGL(ep_symv_sum_kernel, dim3((m + 255) / 256, 1, count), dim3(256), 0, s, partial, n, out, lo, m, sp, sx);
}

// end of sweep: L, Sigma, mu from the current site parameters (EpParameterEstimator.scala:56-61)
//   B = I + S^1/2 K S^1/2 in rows [0, np) of ep->L, K S^1/2 in rows [np, 2 np): ONE blocked factorisation leaves L on top and
//   Vt = (K S^1/2) L^-T below it (the extra rows ride through the panel solves and the trailing updates exactly like y^T does
//   in the regression fit), so V = L \ (S^1/2 K) costs no separate chain of 2 np/128 launches;  Sigma = K - Vt Vt^T, lower
//   triangle only (nothing in a sweep reads above the diagonal; gp_ep_get mirrors on demand), mu = Sigma nu from that triangle.
//   The product is cut at column 512: the first 4 tile columns (all the next sweep's first blocks read) run on the main
//   stream, the other (np/128 - 4)(np/128 - 3)/2 tiles -- one resident round at n = 4096 instead of 528 tiles = one round
//   plus a 16-tile tail -- on the side stream, under the next sweep's first block kernels.  The side stream's work of the next
//   sweep is queued behind it, and the main stream only reads beyond column 512 after it has waited for that work.
gp_status ep_refactor(gp_ep *ep) {
    gp_ctx *ctx = ep->ctx;
    hipStream_t s = ctx->stream, s2 = ctx->side;
    const int n = ep->n, np = ep->np, ldl = ep->ldl;
    hipLaunchKernelGGL(vec_sqrt_kernel, g1(np), dim3(256), 0, s, ep->st(), ep->tau(), n, np);
    hipLaunchKernelGGL(ep_bmat_kernel, dim3(8, np < 65535 ? np : 65535), dim3(256), 0, s, ep->L, ldl, ep->K, ep->st(), np);
    double *Vt = ep->L + np;
    hipLaunchKernelGGL(scale_cols_kernel, dim3(2048), dim3(256), 0, s, Vt, ldl, ep->K, np, ep->st(), np, np);
    gpi_chol_blocked(ctx, ep->L, np, ldl, ep->dinv, np);   // strict upper triangle of the top block stays zero from allocation
    const bool split = np > 1024 && [] { const char *e = getenv("GPCORE_EP_OVERLAP"); return !e || atoi(e) != 0; }();
    const int c1 = split ? 512 : np;
    gp_prof_begin(ctx, GP_PROF_SYRK);
    gpk_gemm_nt(s, np, c1, np, -1.0, Vt, ldl, Vt, ldl, 1.0, ep->Sig, np, 1, 0, gp_batch(), ep->K, np);
    gp_prof_end(ctx, GP_PROF_SYRK, 2.0 * GP_NB * GP_NB * np * ((double)(np / GP_NB) * (c1 / GP_NB) - (double)(c1 / GP_NB) * (c1 / GP_NB - 1) / 2.0));
    double *partial;   // rows of the two parts are disjoint, so both streams may use it at once
    GP_TRY(gpi_ws_get(ctx, WS_PARTIAL, sizeof(double) * (size_t)(SYMV_CHUNKS + 1) * np, &partial));
    // the second part's mu = Sigma nu runs under the NEXT sweep's first block kernels, which already write new site parameters:
    // it multiplies a snapshot of nu taken here (found by tools/ep_identity_check.py: several sweeps in one call drifted from
    // one sweep per call by 1e-4 .. 1e-2 relative at n = 2500 / 3000 before this copy existed)
    const double *nu_now = ep->nu();
    if (split) {
        GP_HIP(ctx, hipMemcpyAsync(ep->tmp2(), ep->nu(), sizeof(double) * (size_t)np, hipMemcpyDeviceToDevice, s));
        nu_now = ep->tmp2();
    }
    ep_symv_lower(s, ep->Sig, np, np, nu_now, partial, ep->mu(), 0, c1);
    if (split) {
        const int r = np - c1;
        // the second part starts when the first is done: started together, the two launches share the chip and the columns the
        // next sweep needs first arrive last
        GP_HIP(ctx, hipEventRecord(ep->ev_parta, s));
        GP_HIP(ctx, hipStreamWaitEvent(s2, ep->ev_parta, 0));
        gp_prof_begin(ctx, GP_PROF_SYRK, s2);
        gpk_gemm_nt(s2, r, r, np, -1.0, Vt + c1, ldl, Vt + c1, ldl, 1.0, ep->Sig + (size_t)c1 + (size_t)c1 * np, np, 1, 0, gp_batch(),
                    ep->K + (size_t)c1 + (size_t)c1 * np, np);
        gp_prof_end(ctx, GP_PROF_SYRK, 2.0 * GP_NB * GP_NB * np * ((double)(r / GP_NB) * (r / GP_NB + 1) / 2.0), s2);
        ep_symv_lower(s2, ep->Sig, np, np, nu_now, partial, ep->mu(), c1, r);   // rows >= 512 of mu also read the first 512 columns
        GP_HIP(ctx, hipEventRecord(ep->ev_partb, s2));
        ep->side_pending = true;
    }
    ep->sig_mirrored = false;
    return GP_OK;
}

// everything the side stream still owes (second part of Sigma and mu) is ordered before whatever the main stream does next
gp_status ep_join_side(gp_ep *ep) {
    if (ep->side_pending) GP_HIP(ep->ctx, hipStreamWaitEvent(ep->ctx->stream, ep->ev_partb, 0));
    ep->side_pending = false;
    return GP_OK;
}

// device buffers + labels; K is left for the caller to fill (host upload or device Gram), then ep_start()
// G > 1: every buffer holds G problems back to back (ep_slab); the returned object is problem 0 and owns the memory
gp_status ep_alloc(gp_ctx *ctx, int n, const int32_t *y, gp_ep **out, int G = 1) {
    *out = nullptr;
    for (int i = 0; i < n; ++i) GP_REQUIRE(ctx, y[i] == 1 || y[i] == -1, "targets must contain values from set {-1,1}");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ep_block1_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, EP_BLOCK1_LDS));
    GP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ep_block2_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, EP_BLOCK2_LDS));
    GP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ep_block2_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, EP_BLOCK2_LDS));
    gp_ep *ep = new (std::nothrow) gp_ep();
    if (!ep) return GP_ENOMEM;
    ep->ctx = ctx; ep->n = n; ep->np = gp_pad(n); ep->ldl = 2 * ep->np;
    const size_t np = ep->np, nn = (size_t)G * np * np * sizeof(double);
    hipError_t e = hipMalloc(&ep->K, nn);
    if (e == hipSuccess) e = hipMalloc(&ep->Sig, nn);
    if (e == hipSuccess) e = hipMalloc(&ep->L, 2 * nn);
    if (e == hipSuccess) e = hipMalloc(&ep->dinv, G * np * 16 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&ep->S, np * GP_NB * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&ep->Sc, G * np * EP_OUTER * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&ep->blk, (size_t)G * 2 * (GP_NB * GP_NB + 8 * 256) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&ep->vec, G * 10 * np * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&ep->cvec, (size_t)G * 4 * GP_NB * sizeof(double));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ep->ev_chol, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ep->ev_parta, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ep->ev_partb, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ep->ev_w, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ep->ev_pipe, hipEventDisableTiming);
    ep->ev.assign(4 * (np / GP_NB), nullptr);
    for (hipEvent_t &ev : ep->ev)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&ep->y, np * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&ep->flags, (size_t)G * (np / GP_NB + 1) * sizeof(int));
    hipStream_t s = ctx->stream;
    if (e == hipSuccess) e = hipMemsetAsync(ep->flags, 0, (size_t)G * (np / GP_NB + 1) * sizeof(int), s);
    if (e == hipSuccess) e = hipMalloc(&ep->uflags, (np / GP_NB + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(ep->uflags, 0, (np / GP_NB + 1) * sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(ep->K, 0, nn, s);
    if (e == hipSuccess) e = hipMemsetAsync(ep->L, 0, 2 * nn, s);
    if (e == hipSuccess) e = hipMemsetAsync(ep->y, 0, np * sizeof(int), s);
    if (e == hipSuccess) e = hipMemcpyAsync(ep->y, y, n * sizeof(int), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) { GP_SET_ERR(ctx, "EP allocation (n=%d) failed: %s", n, hipGetErrorString(e)); gp_ep_destroy(ep); return GP_ENOMEM; }
    *out = ep;
    return GP_OK;
}

// (re)start EP on the Gram matrix now in ep->K (rows/cols < n): zero site parameters, Sigma = K  (:32-38)
gp_status ep_start(gp_ep *ep) {
    gp_ctx *ctx = ep->ctx;
    hipStream_t s = ctx->stream;
    GP_HIP(ctx, hipMemsetAsync(ep->vec, 0, 10 * (size_t)ep->np * sizeof(double), s));
    GP_HIP(ctx, hipMemsetAsync(ep->cvec, 0, 4 * GP_NB * sizeof(double), s));
    gpk_pad_identity(s, ep->K, ep->n, ep->np, ep->np);
    gpk_copy_2d(s, ep->Sig, ep->np, ep->K, ep->np, ep->np, ep->np);   // sigmaMatrix = kernelMatrix.copy (:35)
    ep->sig_mirrored = true;
    ep->side_pending = false;
    ep->sweeps = 0;
    GP_HIP(ctx, hipStreamSynchronize(s));
    return GP_OK;
}

// Block pb = b - 1's delayed update of the trailing covariance, enqueued on `st` in iteration b (i0 = 128 b: first row / column that
// still matters): Sigma[i0:, i0:] -= Sc S^T (lower), rank 128, everything at once.  Sc holds the scaled columns (block pb in column
// block pb % 4 of its buffer), S is in place in Sigma's dead column panels.  bt: lockstep batch (strides A = Sc, B = C = Sigma).
// (A two-level form -- rank-128 updates on the columns the chain reads next, one rank-512 update per outer panel for the rest -- was
// measured equal in a batch, 228.9 vs 229.1 sweeps/s aggregate, and slower in a single run, 184.7 vs 188.0: the site loop's updates are
// an eighth of a sweep's flops.  Removed in round 4.)
void ep_trailing_update(gp_ctx *ctx, hipStream_t st, int np, int i0, double *Sc, double *Sig, gp_batch bt, int *uflag = nullptr) {
    const int pb = i0 / GP_NB - 1, jb = pb % (EP_OUTER / GP_NB);
    const int M = np - i0;
    if (M <= GP_NB) return;          // block b is the last one: nothing below or right of it
    const double tiles = (double)(M / GP_NB) * (M / GP_NB) - (double)(M / GP_NB) * (M / GP_NB - 1) / 2.0;
    gp_prof_begin(ctx, GP_PROF_GEMM, st);
    gpk_gemm_nt(st, M, M, GP_NB, -1.0, Sc + (size_t)jb * GP_NB * np + i0, np, Sig + (size_t)i0 + (size_t)(i0 - GP_NB) * np, np, 1.0,
                Sig + (size_t)i0 + (size_t)i0 * np, np, 1, 0, bt, nullptr, 0, uflag);
    gp_prof_end(ctx, GP_PROF_GEMM, bt.count * tiles * 2.0 * GP_NB * GP_NB * GP_NB, st);
}

// ---- lockstep batch: G EP problems of one size advance through a sweep together (MeshHyperParamsLogLikelihoodEvaluator.scala:26-40
// and the trial steps of HyperParamsOptimization.scala:31-55 evaluate independent settings one after the other) ----
// A single sweep is bound by its serial site chain (one workgroup per block of 128 sites) and by the launch latencies of the ~12
// small kernels per block that feed it; the matrix work beside it leaves most of the chip idle (C4: 33 of 78 TFLOP/s).  G problems
// share every launch instead: the chain kernel runs as G workgroups on G CUs (blockIdx.x = problem), the solves, updates and the
// refactorisation under the site loop as gp_batch launches with G times the tiles -- the same sweep, the same arithmetic per problem
// (bit for bit: every kernel computes a problem's tiles exactly as in a single run), G times the work per launch latency.
struct ep_slab {
    gp_ctx *ctx = nullptr;
    int n = 0, np = 0, G = 0;
    std::vector<gp_ep *> ep;        // ep[0] owns the memory, ep[g] views problem g
    int *info = nullptr;            // G failing-pivot words
    double *partial = nullptr;      // G x (SYMV_CHUNKS + 1) x np
    size_t sMat() const { return (size_t)np * np; }
    size_t sL() const { return (size_t)2 * np * np; }
    size_t sVec() const { return (size_t)10 * np; }
    size_t sBlk() const { return (size_t)2 * EP_BLK_ELEMS; }
    size_t sCv() const { return (size_t)4 * GP_NB; }
    size_t sSc() const { return (size_t)np * EP_OUTER; }
    size_t sDinv() const { return (size_t)np * 16; }
    int sFlag() const { return np / GP_NB + 1; }
    size_t sPart() const { return (size_t)(SYMV_CHUNKS + 1) * np; }
};

void ep_slab_free(ep_slab &sl) {
    if (sl.ctx) (void)hipSetDevice(sl.ctx->device);
    for (size_t g = sl.ep.size(); g-- > 1;) gp_ep_destroy(sl.ep[g]);
    if (!sl.ep.empty()) gp_ep_destroy(sl.ep[0]);     // synchronises the context's streams first
    sl.ep.clear();
    if (sl.info) (void)hipFree(sl.info);
    if (sl.partial) (void)hipFree(sl.partial);
    sl.info = nullptr, sl.partial = nullptr;
}

gp_status ep_slab_alloc(gp_ctx *ctx, int n, const int32_t *y, int G, ep_slab &sl) {
    sl.ctx = ctx, sl.n = n, sl.np = gp_pad(n), sl.G = G;
    gp_ep *e0 = nullptr;
    GP_TRY(ep_alloc(ctx, n, y, &e0, G));
    sl.ep.push_back(e0);
    hipError_t e = hipMalloc(&e0->Sig2, sizeof(double) * G * sl.sMat());
    if (e == hipSuccess) e = hipMalloc(&sl.info, sizeof(int) * G);
    if (e == hipSuccess) e = hipMalloc(&sl.partial, sizeof(double) * G * sl.sPart());
    if (e != hipSuccess) {
        (void)hipGetLastError();
        GP_SET_ERR(ctx, "EP lockstep batch of %d problems (n = %d): out of device memory", G, n);
        ep_slab_free(sl);
        return GP_ENOMEM;
    }
    for (int g = 1; g < G; ++g) {
        gp_ep *v = new (std::nothrow) gp_ep();
        if (!v) { ep_slab_free(sl); return GP_ENOMEM; }
        v->owns = false, v->ctx = ctx, v->n = n, v->np = sl.np, v->ldl = e0->ldl;
        v->K = e0->K + g * sl.sMat(), v->Sig = e0->Sig + g * sl.sMat(), v->Sig2 = e0->Sig2 + g * sl.sMat(), v->L = e0->L + g * sl.sL();
        v->dinv = e0->dinv + g * sl.sDinv(), v->S = e0->S, v->Sc = e0->Sc + g * sl.sSc(), v->blk = e0->blk + g * sl.sBlk();
        v->vec = e0->vec + g * sl.sVec(), v->cvec = e0->cvec + g * sl.sCv(), v->y = e0->y, v->flags = e0->flags + (size_t)g * sl.sFlag();
        sl.ep.push_back(v);
    }
    return GP_OK;
}

// ONE sweep of problems 0 .. count-1 of the slab: gp_ep_sweep's fused chain + streamed refactorisation, every launch a batch.
gp_status ep_sweep_lockstep(ep_slab &sl, int count) {
    gp_ctx *ctx = sl.ctx;
    GP_TRY(gpi_ctx_ep_streams(ctx));
    gp_ep *e0 = sl.ep[0];
    // The fourth stream is CU-masked for the single run's sake (64 CUs kept free of its long GEMMs so that the chain's small kernels
    // find a CU); a batch is bound by GEMM throughput instead and loses by the mask (12 problems at n = 4096, aggregate sweeps/s with
    // 0 / 32 / 64 / 96 CUs reserved: 289 / 281 / 271 / 245), so its long GEMMs go to the unmasked third stream behind the factorisation
    // chain from np = 4096 on (n = 4096: 287.6 against 272.1 sweeps/s aggregate; n = 8192: 39.8 either way; below that the fourth stream
    // pays: n = 2048 1669 against 1559).  GPCORE_EP_LOCK_S4 = 0 / 1 forces.
    const bool own_s4 = [np = sl.np] { const char *e = getenv("GPCORE_EP_LOCK_S4"); return e ? atoi(e) != 0 : np < 4096; }();
    hipStream_t s = ctx->stream, s2 = ctx->side, s3 = ctx->side2, s4 = own_s4 ? ctx->side3 : ctx->side2;
    const int n = sl.n, np = sl.np, nblk = np / GP_NB, ldl = e0->ldl;
    // columns of Vt per next-covariance update: a batch is bound by GEMM throughput, not by the chain, so long updates pay (n = 4096,
    // 12 problems, K = 256 / 512 / 1024: 229 / 239 / 244 sweeps/s aggregate)
    const int sig_blocks = [np] { const int v = gp_env_blocks("GPCORE_EP_SIG_K"); return v >= GP_NB ? v / GP_NB : (np >= 4096 ? 8 : 2); }();
    const bool far_split = [] { const char *e = getenv("GPCORE_EP_FAR"); return !e || atoi(e) != 0; }();
    ep_strides es;
    es.sig = sl.sMat(), es.vec = sl.sVec(), es.y = 0, es.blk = sl.sBlk(), es.cvec = sl.sCv(), es.sc = sl.sSc(), es.flag = sl.sFlag();
    gp_batch bSig, bTrsm, bVt;
    bSig.count = bTrsm.count = bVt.count = count;
    bSig.s0 = sl.sSc(), bSig.s1 = sl.sMat(), bSig.s2 = sl.sMat();                      // C (Sigma) -= A (Sc) B (Sigma's dead panel)^T
    bTrsm.s0 = sl.sMat(), bTrsm.s1 = sl.sBlk(), bTrsm.s2 = sl.sBlk(), bTrsm.s3 = sl.sSc(), bTrsm.s4 = sl.sCv(), bTrsm.s5 = sl.sVec();
    bVt.s0 = bVt.s1 = sl.sL(), bVt.s2 = sl.sMat(), bVt.s3 = sl.sMat();                 // Sigma' (+= K the first time) -= Vt Vt^T
    const int token = ++e0->epoch;
    int *flag_err = e0->flags + np / GP_NB;        // problem 0's error word collects every problem's
    GP_HIP(ctx, hipMemsetAsync(sl.info, 0, sizeof(int) * count, s));
    GP_HIP(ctx, hipMemsetAsync(flag_err, 0, sizeof(int), s));
    GP_HIP(ctx, hipEventRecord(e0->ev_w, s));
    GP_HIP(ctx, hipStreamWaitEvent(s3, e0->ev_w, 0));
    GP_HIP(ctx, hipStreamWaitEvent(s4, e0->ev_w, 0));
    hipLaunchKernelGGL(ep_winit_kernel, dim3(8, np < 4096 ? np : 4096, count), dim3(256), 0, s3, e0->L, ldl, e0->K, np, sl.sL(), sl.sMat());
    int b = 0, pend0 = 0;
    hipEvent_t last_side = nullptr;
    for (int i0 = 0; i0 < n; i0 += GP_NB, ++b) {
        const int bsz = (n - i0 < GP_NB) ? n - i0 : GP_NB;
        const int par = b & 1;
        double *Lmat = e0->blk + (size_t)par * EP_BLK_ELEMS, *bdinv = Lmat + GP_NB * GP_NB;
        double *cvec = e0->cvec + (size_t)par * 2 * GP_NB, *ncoef = cvec + GP_NB;
        if (b >= 2) GP_HIP(ctx, hipStreamWaitEvent(s, e0->ev[4 * (b - 2) + 2], 0));
        if (b > 0)
            hipLaunchKernelGGL(ep_block2_kernel<true>, dim3(count), dim3(64 * EP_BLOCK1_WAVES), EP_BLOCK2_LDS, s, n, np, i0, bsz, e0->Sig, e0->vec, e0->y,
                               e0->blk, e0->cvec, e0->Sc, e0->flags, par, token, es, ((b - 1) % (EP_OUTER / GP_NB)) * GP_NB * np, nullptr, 0, nullptr);
        else
            hipLaunchKernelGGL(ep_block2_kernel<false>, dim3(count), dim3(64 * EP_BLOCK1_WAVES), EP_BLOCK2_LDS, s, n, np, i0, bsz, e0->Sig, e0->vec, e0->y,
                               e0->blk, e0->cvec, e0->Sc, e0->flags, par, token, es, 0, nullptr, 0, nullptr);
        hipEvent_t ev_fac = e0->ev[4 * b], ev_vt = e0->ev[4 * b + 3];
        GP_HIP(ctx, hipEventRecord(ev_fac, s));
        // refactorisation under the site loop, one block behind it (third and fourth stream)
        GP_HIP(ctx, hipStreamWaitEvent(s3, ev_fac, 0));
        {
            const int gx = (2 * np - i0 + 1023) / 1024;
            hipLaunchKernelGGL(ep_wscale_kernel, dim3(gx, GP_NB + std::min((i0 / 2 + gx - 1) / gx, 256), count), dim3(256), 0, s3, e0->L, ldl, np, i0,
                               e0->tau(), n, e0->st(), sl.sL(), sl.sVec());
        }
        gpi_chol_panel_step(ctx, s3, e0->L, np, ldl, e0->dinv, np, i0, ev_vt, (far_split && own_s4) ? s4 : nullptr, e0->ev_parta, count, sl.sL(), sl.sDinv(), sl.info);
        if ((b + 1) % sig_blocks == 0 || b >= nblk - 2) {
            GP_HIP(ctx, hipStreamWaitEvent(s4, ev_vt, 0));
            const int kw = i0 + GP_NB - pend0;
            const double *Vb = e0->L + np + (size_t)pend0 * ldl;
            gp_prof_begin(ctx, GP_PROF_SYRK, s4);
            gpk_gemm_nt(s4, np, np, kw, -1.0, Vb, ldl, Vb, ldl, 1.0, e0->Sig2, np, 1, 0, bVt, pend0 == 0 ? e0->K : nullptr, np);
            gp_prof_end(ctx, GP_PROF_SYRK, (double)count * np * ((double)np + GP_NB) * kw, s4);
            pend0 = i0 + GP_NB;
        }
        if (b > 0) {
            // second stream: block b-1's trailing update of every problem, after every problem's flag
            if (np - i0 > GP_NB) {
                hipLaunchKernelGGL(ep_wait_flag_kernel, dim3(count), dim3(1), 0, s2, e0->flags + b, sl.sFlag(), token, flag_err);
                ep_trailing_update(ctx, s2, np, i0, e0->Sc, e0->Sig, bSig);
            }
            GP_HIP(ctx, hipEventRecord(e0->ev[4 * (b - 1) + 2], s2));
            last_side = e0->ev[4 * (b - 1) + 2];
        }
        const int r0 = i0 + GP_NB, rt = np - r0;
        if (rt <= 0 || r0 >= n) continue;
        const int rest = rt - GP_NB;
        if (rest > 0) {
            double *St = e0->Sig + (size_t)r0 + (size_t)i0 * np, *Sct = e0->Sc + (size_t)(b % (EP_OUTER / GP_NB)) * GP_NB * np + r0;
            GP_HIP(ctx, hipStreamWaitEvent(s2, ev_fac, 0));
            gpk_trsm_panel128(s2, St + GP_NB, rest, np, Lmat, GP_NB, bdinv, nullptr, ncoef, e0->mu() + r0 + GP_NB, bTrsm, Sct + GP_NB, cvec);
        }
    }
    if (last_side) GP_HIP(ctx, hipStreamWaitEvent(s, last_side, 0));
    GP_HIP(ctx, hipEventRecord(e0->ev_pipe, s4));
    GP_HIP(ctx, hipStreamWaitEvent(s, e0->ev_pipe, 0));
    GP_HIP(ctx, hipEventRecord(e0->ev_chol, s3));
    GP_HIP(ctx, hipStreamWaitEvent(s, e0->ev_chol, 0));
    for (gp_ep *v : sl.ep) { std::swap(v->Sig, v->Sig2); v->sig_mirrored = false; }
    ep_symv_lower(s, e0->Sig, np, np, e0->nu(), sl.partial, e0->mu(), 0, np, count, sl.sMat(), sl.sVec(), sl.sPart());
    GP_LAUNCH_CHECK(ctx);
    for (int g = 0; g < count; ++g) sl.ep[g]->sweeps += 1;
    return GP_OK;
}

}  // namespace

extern "C" {

gp_status gp_ep_create(gp_ctx *ctx, const double *K, int n, int ldk, const int32_t *y, gp_ep **out) {
    if (!ctx || !out) return GP_EINVAL;
    *out = nullptr;
    GP_REQUIRE(ctx, K && y && n >= 1 && ldk >= n, "bad arguments");   // require(kernelMatrix.rows == targets.length)
    gp_ep *ep = nullptr;
    GP_TRY(ep_alloc(ctx, n, y, &ep));
    gp_status st = gpi_upload_2d(ctx, ep->K, ep->np, K, ldk, n, n);
    if (st != GP_OK) { gp_ep_destroy(ep); return st; }
    st = ep_start(ep);
    if (st != GP_OK) { gp_ep_destroy(ep); return st; }
    *out = ep;
    return GP_OK;
}

gp_status gp_ep_sweep(gp_ep *ep, int nsweeps, double *tau, double *nu, int *info) {
    if (!ep) return GP_EINVAL;
    gp_ctx *ctx = ep->ctx;
    GP_REQUIRE(ctx, nsweeps >= 0, "nsweeps < 0");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int n = ep->n, np = ep->np;
    if (info) *info = 0;
    GP_HIP(ctx, hipMemsetAsync(ctx->d_info, 0, sizeof(int), s));
    // GPCORE_EP_OVERLAP=0: every launch of a site block on one stream (the form the overlapped one is tested against)
    const bool overlap = [] { const char *e = getenv("GPCORE_EP_OVERLAP"); return !e || atoi(e) != 0; }();
    // GPCORE_EP_PIPELINE: 1 / 0 force the streamed refactorisation on / off; default on above np = 1024 (sweeps/s with / without
    // it: n = 1024 552 / 592, n = 2048 382 / 265, n = 4096 162 / 102, n = 8192 29.1 / 27.2)
    const bool pipe = overlap && np >= 2 * GP_NB && [np] { const char *e = getenv("GPCORE_EP_PIPELINE"); return e ? atoi(e) != 0 : np > 1024; }();
    if (pipe) GP_TRY(gpi_ctx_ep_streams(ctx));
    hipStream_t s2 = ctx->side, s3 = ctx->side2, s4 = ctx->side3;
    // The link between two block kernels as ONE launch (gpk_ep_link) or as panel solve + tile update on 2 + 36 workgroups: alone the two
    // launches are faster (the single workgroup is bound by one CU's matrix pipe: n = 4096 end-of-sweep form 106.9 vs 103.3 sweeps/s),
    // under the streamed refactorisation's GEMMs the fused one is (167.4 vs 163.9) -- so it follows `pipe` (GPCORE_EP_LINK overrides)
    const bool fused_link = [pipe] { const char *e = getenv("GPCORE_EP_LINK"); return e ? atoi(e) != 0 : pipe; }();
    // GPCORE_EP_FUSED: the link of block b-1 is the prologue of block b's kernel (ep_block2_kernel) -- one launch per block on the
    // chain; 0: block kernel + link kernel (the form it is tested against, bit for bit).  Measured (sweeps/s fused / two launches):
    // n = 4096 181.0 / 182.1, n = 2048 425.5 / 411.8, n = 1024 627.0 / 668.3 -- the prologue costs what the link kernel did (34 us:
    // profiles/r03_c_sweep_fused.txt), the chain is as much bound by the side stream's solve + updates as by its own kernels, so
    // the default follows the streamed refactorisation (np > 1024); the lockstep batch (ep_sweep_lockstep) always uses it
    const bool fused = overlap && [pipe] { const char *e = getenv("GPCORE_EP_FUSED"); return e ? atoi(e) != 0 : pipe; }();
    int *flag_err = ep->flags + np / GP_NB;
    if (fused) GP_HIP(ctx, hipMemsetAsync(flag_err, 0, sizeof(int), s));
    // urgent tiles (see the launch of ep_block2_kernel below)
    const bool urgent = fused && [] { const char *e = getenv("GPCORE_EP_URGENT"); return !e || atoi(e) != 0; }();
    const bool far_split = [] { const char *e = getenv("GPCORE_EP_FAR"); return !e || atoi(e) != 0; }();
    // columns of Vt per next-covariance update (GPCORE_EP_SIG_K; n = 4096 sweeps/s at 128 / 256 / 384 / 512 / 1024 / 2048: 173 / 181 / 175 / 177 /
    // 172 / 155 -- short enough to spread the fourth stream's load evenly, long enough for the GEMM)
    // (n = 8192: 29.3 sweeps/s at K = 512 against 28.0 at 256, n = 2048: 423 against 406 -- 512 from np = 6144 on)
    const int sig_blocks = [np] { const int v = gp_env_blocks("GPCORE_EP_SIG_K"); return v >= GP_NB ? v / GP_NB : (np >= 6144 ? 4 : 2); }();
    const int nblk = np / GP_NB;
    double *partial = nullptr;
    if (pipe) {
        if (!ep->Sig2 && hipMalloc(&ep->Sig2, sizeof(double) * (size_t)np * np) != hipSuccess) {
            (void)hipGetLastError();
            GP_SET_ERR(ctx, "EP: %d x %d second covariance buffer: out of device memory", np, np);
            return GP_ENOMEM;
        }
        GP_TRY(gpi_ws_get(ctx, WS_PARTIAL, sizeof(double) * (size_t)(SYMV_CHUNKS + 1) * np, &partial));
    }
    for (int sw = 0; sw < nsweeps; ++sw) {
        const int token = ++ep->epoch;
        const int utarget = urgent ? ++ep->uepoch : 0;
        // Only the TRAILING part of the recurrence is carried: the sites after a block read mu_i and Sigma_ii "as of now",
        // which depend on the earlier blocks through rows/columns >= their own block only, and the end-of-sweep
        // refactorisation (:56-61) rebuilds Sigma and mu from the site parameters anyway.  So the delayed columns
        // S = Sigma0[r0:, blk] Lmat^-T, the mean update mu[r0:] += S coef and the rank-128 update
        // Sigma[r0:, r0:] -= S diag(c) S^T cover rows/columns r0 = i0 + 128 onwards (lower triangle) -- n^3/3 flops per
        // sweep on the MFMA syrk instead of 2 n^3 on full-square updates.  The block's own column panel of Sigma is dead after
        // the block, so the panel solve runs in place on it; the same launch writes the scaled copy S diag(c) and adds S coef
        // to the mean (row dot fused into the panel solve).
        //
        // Two streams.  The serial chain -- block kernel b (one workgroup, ~110 us) -> the 128 rows of S that belong to block
        // b+1 -> the ONE 128 x 128 tile of the update block kernel b+1 reads -> block kernel b+1 -- stays on the main stream;
        // the rest of block b's panel solve and rank-128 update (all rows from block b+2 on) runs on the side stream UNDER block
        // kernel b+1.  The side stream is CU-masked (gp_ctx_create), so the block kernel, which needs most of a CU's LDS,
        // always finds a free CU.  Small per-block buffers (Lmat, tile inverses, c, coef) alternate by block parity: the
        // side stream still reads block b's while block kernel b+1 writes its own.
        //
        // Third stream: the refactorisation of :56-61 runs UNDER the site loop, one block behind it.  A site is visited once per
        // sweep, so block b's precisions are final when block kernel b ends; with the factor written as chol(B) = S^1/2 chol(K + S^-1)
        // (ep_winit_kernel) column block b of the right-looking factorisation needs exactly those and nothing later.  Per block:
        // scale block b's rows and columns of the working matrix, one step of the two-level Cholesky with the np rows of K S^1/2
        // riding along (-> Vt[:, b] = ((K S^1/2) L^-T)[:, b]), and the rank-128 update  Sigma_next -= Vt[:, b] Vt[:, b]^T  of the
        // NEXT covariance (second buffer; the site loop still works in the current one).  What is left after the last block
        // kernel is the last block's own step and mu = Sigma nu.
        if (pipe) {
            GP_HIP(ctx, hipEventRecord(ep->ev_w, s));
            GP_HIP(ctx, hipStreamWaitEvent(s3, ep->ev_w, 0));
            GP_HIP(ctx, hipStreamWaitEvent(s4, ep->ev_w, 0));
            hipLaunchKernelGGL(ep_winit_kernel, dim3(8, np < 4096 ? np : 4096), dim3(256), 0, s3, ep->L, ep->ldl, ep->K, np);
        }
        int b = 0, pend0 = 0;   // pend0: first column of Vt not yet in the next covariance
        hipEvent_t last_side = nullptr;
        for (int i0 = 0; i0 < n; i0 += GP_NB, ++b) {
            const int bsz = (n - i0 < GP_NB) ? n - i0 : GP_NB;
            const int par = b & 1;
            double *Lmat = ep->blk + (size_t)par * (GP_NB * GP_NB + 8 * 256), *bdinv = Lmat + GP_NB * GP_NB;
            double *cvec = ep->cvec + (size_t)par * 2 * GP_NB, *ncoef = cvec + GP_NB;   // c and coef of every site of the block
            if (fused) {
                // the prologue reads Sigma[blk b, blk b-1 .. b]: entries the side stream's update of block b-2 wrote.  With urgent tiles
                // (GPCORE_EP_URGENT, default) the kernel is launched WITHOUT waiting for that launch to finish: the update announces those
                // two tiles with a counter the moment they are stored, and the prologue waits for the counter (bounded).  The chain then
                // never waits for the rest of a 500-tile update it does not read -- the block periods of 200-260 us in the first half of
                // a sweep (profiles/r03_j_c4_sweep_summary.txt) were exactly that.  (Computing the two tiles in a kernel of their own ahead
                // of the update -- 100 one-wave workgroups that need no LDS slot -- was measured too: 190.8 against 197.0 sweeps/s without
                // any urgent tiles on the same box; one more launch on the side stream costs more than the earlier tiles gain.)
                const bool uw = urgent && b >= 2;
                if (b >= 2 && !uw) GP_HIP(ctx, hipStreamWaitEvent(s, ep->ev[4 * (b - 2) + 2], 0));
                if (b > 0)
                    hipLaunchKernelGGL(ep_block2_kernel<true>, dim3(1), dim3(64 * EP_BLOCK1_WAVES), EP_BLOCK2_LDS, s, n, np, i0, bsz, ep->Sig, ep->vec,
                                       ep->y, ep->blk, ep->cvec, ep->Sc, ep->flags, par, token, ep_strides(), ((b - 1) % (EP_OUTER / GP_NB)) * GP_NB * np,
                                       uw ? ep->uflags + (b - 2) : nullptr, 2 * utarget, flag_err);
                else
                    hipLaunchKernelGGL(ep_block2_kernel<false>, dim3(1), dim3(64 * EP_BLOCK1_WAVES), EP_BLOCK2_LDS, s, n, np, i0, bsz, ep->Sig, ep->vec,
                                       ep->y, ep->blk, ep->cvec, ep->Sc, ep->flags, par, token, ep_strides(), 0, nullptr, 0, nullptr);
            } else
                hipLaunchKernelGGL(ep_block1_kernel, dim3(1), dim3(64 * EP_BLOCK1_WAVES), EP_BLOCK1_LDS, s, n, np, i0, bsz, ep->Sig, ep->mu(), ep->y,
                                   ep->tau(), ep->nu(), ep->cav_tau(), ep->cav_nu(), cvec, ncoef, Lmat, bdinv);
            hipEvent_t ev_fac = ep->ev[4 * b], ev_rows = ep->ev[4 * b + 1], ev_side = ep->ev[4 * b + 2], ev_vt = ep->ev[4 * b + 3];
            if (overlap) GP_HIP(ctx, hipEventRecord(ev_fac, s));
            if (pipe) {
                GP_HIP(ctx, hipStreamWaitEvent(s3, ev_fac, 0));
                {
                    const int gx = (2 * np - i0 + 1023) / 1024;
                    hipLaunchKernelGGL(ep_wscale_kernel, dim3(gx, GP_NB + std::min((i0 / 2 + gx - 1) / gx, 256)), dim3(256), 0, s3, ep->L, ep->ldl, np, i0,
                                       ep->tau(), n, ep->st());
                }
                gpi_chol_panel_step(ctx, s3, ep->L, np, ep->ldl, ep->dinv, np, i0, ev_vt, far_split ? s4 : nullptr, ep->ev_parta);
                // the finished columns of Vt go into the next covariance on a stream of their own, two blocks (K = 256) at a time;
                // the last two blocks one by one so that only a K = 128 update is left after the last block kernel
                if ((b + 1) % sig_blocks == 0 || b >= nblk - 2) {
                    GP_HIP(ctx, hipStreamWaitEvent(s4, ev_vt, 0));
                    const int kw = i0 + GP_NB - pend0;
                    const double *Vb = ep->L + np + (size_t)pend0 * ep->ldl;
                    gp_prof_begin(ctx, GP_PROF_SYRK, s4);
                    gpk_gemm_nt(s4, np, np, kw, -1.0, Vb, ep->ldl, Vb, ep->ldl, 1.0, ep->Sig2, np, 1, 0, gp_batch(), pend0 == 0 ? ep->K : nullptr, np);
                    gp_prof_end(ctx, GP_PROF_SYRK, (double)np * ((double)np + GP_NB) * kw, s4);
                    pend0 = i0 + GP_NB;
                }
            }
            if (fused && b > 0) {
                // side stream: block b-1's whole trailing update, Sigma[i0:, i0:] -= Sc S^T (lower), in ONE launch.  Its operands' first
                // 128 rows are what this kernel's prologue has just solved: announced by a device flag, awaited by a one-thread kernel
                // (enqueued AFTER the producer, which depends on nothing later on any stream; the wait is bounded and reports instead
                // of hanging).  The rows below were solved by the side stream itself in the iteration before.  (The tile of this
                // block itself, which the chain keeps in LDS, is updated here too -- nobody reads it again -- so that the product is
                // one lower-trapezoid launch.)  Every launch on this stream costs ~40 us while the other streams' GEMMs hold the CUs,
                // whatever its size (profiles/r03_d_sweep_urgent_first.txt), so the chain's slack is spent on three launches per block
                // (solve, wait, update), not on five.
                if (np - i0 > GP_NB) {
                    hipLaunchKernelGGL(ep_wait_flag_kernel, dim3(1), dim3(1), 0, s2, ep->flags + b, 0, token, flag_err);
                    ep_trailing_update(ctx, s2, np, i0, ep->Sc, ep->Sig, gp_batch(), urgent ? ep->uflags + (b - 1) : nullptr);
                }
                GP_HIP(ctx, hipEventRecord(ep->ev[4 * (b - 1) + 2], s2));
                last_side = ep->ev[4 * (b - 1) + 2];
            }
            const int r0 = i0 + GP_NB, rt = np - r0;
            if (rt <= 0 || r0 >= n) continue;
            double *St = ep->Sig + (size_t)r0 + (size_t)i0 * np;
            double *Sct = ep->Sc + (fused ? (size_t)(b % (EP_OUTER / GP_NB)) * GP_NB * np : (size_t)0) + r0;   // fused: the block's column block of the panel
            double *Ctr = ep->Sig + (size_t)r0 + (size_t)r0 * np;
            if (!overlap) {
                gpk_trsm_panel128(s, St, rt, np, Lmat, GP_NB, bdinv, nullptr, ncoef, ep->mu() + r0, gp_batch(), Sct, cvec);
                gp_prof_begin(ctx, GP_PROF_GEMM);
                gpk_gemm_nt(s, rt, rt, GP_NB, -1.0, Sct, np, St, np, 1.0, Ctr, np, 1);
                gp_prof_end(ctx, GP_PROF_GEMM, (double)rt * ((double)rt + GP_NB) * GP_NB);
                continue;
            }
            const int rest = rt - GP_NB;            // rows from block b+2 on
            if (rest > 0) {
                GP_HIP(ctx, hipStreamWaitEvent(s2, ev_fac, 0));
                // side stream, part 1: rows [r0+128, np) of the panel solve and their own lower triangle of the update
                gpk_trsm_panel128(s2, St + GP_NB, rest, np, Lmat, GP_NB, bdinv, nullptr, ncoef, ep->mu() + r0 + GP_NB, gp_batch(), Sct + GP_NB, cvec);
                if (!fused) {
                    gp_prof_begin(ctx, GP_PROF_GEMM, s2);
                    gpk_gemm_nt(s2, rest, rest, GP_NB, -1.0, Sct + GP_NB, np, St + GP_NB, np, 1.0, Ctr + GP_NB + (size_t)GP_NB * np, np, 1);
                    gp_prof_end(ctx, GP_PROF_GEMM, (double)rest * ((double)rest + GP_NB) * GP_NB, s2);
                }
            }
            if (fused) continue;     // the link is the next block kernel's prologue, part 2 follows its flag (above)
            // main stream: the 128 rows of block b+1 -- they read Sigma entries the side stream's update of block b-1 wrote
            if (b > 0) GP_HIP(ctx, hipStreamWaitEvent(s, ep->ev[4 * (b - 1) + 2], 0));
            if (fused_link) gpk_ep_link(s, St, np, Lmat, bdinv, ncoef, ep->mu() + r0, Sct, cvec, Ctr, np);
            else {
                gpk_trsm_panel128(s, St, GP_NB, np, Lmat, GP_NB, bdinv, nullptr, ncoef, ep->mu() + r0, gp_batch(), Sct, cvec);
                hipLaunchKernelGGL(ep_diag_update_kernel, dim3(36), dim3(64), 0, s, Ctr, np, Sct, St, np);
            }
            if (rest > 0) {
                GP_HIP(ctx, hipEventRecord(ev_rows, s));
                GP_HIP(ctx, hipStreamWaitEvent(s2, ev_rows, 0));
                // side stream, part 2: column panel of block b+1 below its diagonal tile (needs the rows the main stream just solved)
                gpk_gemm_nt(s2, rest, GP_NB, GP_NB, -1.0, Sct + GP_NB, np, St, np, 1.0, Ctr + GP_NB, np, 0);
            }
            GP_HIP(ctx, hipEventRecord(ev_side, s2));
            last_side = ev_side;
        }
        if (last_side) GP_HIP(ctx, hipStreamWaitEvent(s, last_side, 0));   // the refactorisation rewrites Sigma and mu
        if (pipe) {
            GP_HIP(ctx, hipEventRecord(ep->ev_pipe, s4));   // s4's last update waited for s3's last panel solve
            GP_HIP(ctx, hipStreamWaitEvent(s, ep->ev_pipe, 0));
            GP_HIP(ctx, hipEventRecord(ep->ev_chol, s3));   // (the last step has no update after its solve; kept for symmetry)
            GP_HIP(ctx, hipStreamWaitEvent(s, ep->ev_chol, 0));
            std::swap(ep->Sig, ep->Sig2);
            ep_symv_lower(s, ep->Sig, np, np, ep->nu(), partial, ep->mu(), 0, np);
            ep->sig_mirrored = false;
        } else {
            GP_TRY(ep_refactor(ep));
        }
        GP_LAUNCH_CHECK(ctx);
        ep->sweeps += 1;
    }
    GP_TRY(ep_join_side(ep));
    // (host time spent enqueueing, measured in round 3: 1.6 ms per 5 ms sweep at n = 4096 -- the host runs three sweeps ahead of the GPU; not launch-bound)
    int h = 0;
    GP_TRY(gpi_read_info(ctx, &h));
    if (fused) {
        int ferr = 0;
        GP_HIP(ctx, hipMemcpy(&ferr, flag_err, sizeof(int), hipMemcpyDeviceToHost));
        if (ferr) { GP_SET_ERR(ctx, "EP sweep: a device flag of the fused chain did not arrive within its bound (problem %d)", ferr - 1); return GP_EHIP; }
    }
    if (info) *info = h;
    if (h) { GP_SET_ERR(ctx, "I + S^1/2 K S^1/2 not positive definite at pivot %d (negative site precision?)", h); return GP_ENOTPD; }
    if (tau) GP_TRY(gpi_download_2d(ctx, tau, n, ep->tau(), np, n, 1));
    if (nu) GP_TRY(gpi_download_2d(ctx, nu, n, ep->nu(), np, n, 1));
    return GP_OK;
}

gp_status gp_ep_set_site_params(gp_ep *ep, const double *tau, const double *nu, int *info) {
    if (!ep) return GP_EINVAL;
    gp_ctx *ctx = ep->ctx;
    GP_REQUIRE(ctx, tau && nu, "null pointer");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    if (info) *info = 0;
    GP_HIP(ctx, hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
    GP_TRY(gpi_upload_2d(ctx, ep->tau(), ep->np, tau, ep->n, ep->n, 1));
    GP_TRY(gpi_upload_2d(ctx, ep->nu(), ep->np, nu, ep->n, ep->n, 1));
    GP_TRY(ep_refactor(ep));
    GP_TRY(ep_join_side(ep));
    int h = 0;
    GP_TRY(gpi_read_info(ctx, &h));
    if (info) *info = h;
    if (h) { GP_SET_ERR(ctx, "I + S^1/2 K S^1/2 not positive definite at pivot %d", h); return GP_ENOTPD; }
    if (ep->sweeps == 0) ep->sweeps = 1;
    return GP_OK;
}

gp_status gp_ep_lml(gp_ep *ep, int strict, double *lml) {
    if (!ep || !lml) return GP_EINVAL;
    gp_ctx *ctx = ep->ctx;
    GP_REQUIRE(ctx, ep->sweeps > 0, "no sweep has run yet");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(ep_lml_kernel, dim3(1), dim3(1024), 0, ctx->stream, ep->n, ep->ldl, ep->L, ep->tau(), ep->nu(), ep->mu(),
                       ep->cav_tau(), ep->cav_nu(), ep->y, strict, ctx->d_scalars);
    return gpi_download_2d(ctx, lml, 1, ctx->d_scalars, 1, 1, 1);
}

static gp_status ep_lml_grad_dev(gp_ep *ep, const double *dX, int d, const double *theta, int strict, double *grad);

gp_status gp_ep_lml_grad_rbf(gp_ep *ep, const double *X, int d, int ldx, const double *theta, int strict, double *grad) {
    if (!ep) return GP_EINVAL;
    gp_ctx *ctx = ep->ctx;
    const int n = ep->n;
    GP_REQUIRE(ctx, X && theta && grad && d >= 1 && d <= 64 && ldx >= n, "bad arguments (1 <= d <= 64)");
    GP_REQUIRE(ctx, ep->sweeps > 0, "no sweep has run yet");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dX;
    GP_TRY(gpi_ws_get(ctx, WS_A, sizeof(double) * (size_t)n * d, &dX));
    GP_TRY(gpi_upload_2d(ctx, dX, n, X, ldx, n, d));
    return ep_lml_grad_dev(ep, dX, d, theta, strict, grad);
}

// the same with the training inputs already in HBM (n x d, ld = n)
static gp_status ep_lml_grad_dev(gp_ep *ep, const double *dX, int d, const double *theta, int strict, double *grad) {
    gp_ctx *ctx = ep->ctx;
    const int n = ep->n, np = ep->np;
    hipStream_t s = ctx->stream;
    const int P = d + 2;
    double *partial, *parts, *dres;
    GP_TRY(gpi_ws_get(ctx, WS_PARTIAL, sizeof(double) * (size_t)(SYMV_CHUNKS + 1) * np, &partial));
    GP_TRY(gpi_ws_get(ctx, WS_SUMSQ, sizeof(double) * (size_t)gpk_lml_grad_partials_size(n, d), &parts));
    GP_TRY(gpi_ws_get(ctx, WS_C, sizeof(double) * (size_t)(P + 1), &dres));
    double *t1 = ep->tmp1(), *t2 = ep->tmp2();
    // rhs = S^1/2 K nu                                                    MarginalLikelihoodEvaluator.scala:53-54
    gpk_gemv_rows(s, ep->K, n, n, np, ep->nu(), t1, partial, 16);
    hipLaunchKernelGGL(vec_mul_kernel, g1(n), dim3(256), 0, s, t1, t1, ep->st(), n);
    const double *Kinv = nullptr;
    if (strict) {
        gpi_back_solve_vec(ctx, ep->L, np, ep->ldl, ep->dinv, t1, t2);                       // temp = L^T \ rhs           :53
        hipLaunchKernelGGL(vec_div_kernel, g1(n), dim3(256), 0, s, t2, t2, ep->st(), n);   // (S^1/2 L) x = temp  <=>  L x = temp / st
        gpi_forward_solve_vec(ctx, ep->L, np, ep->ldl, ep->dinv, t2, t1);                    //                              :55-56
        hipLaunchKernelGGL(vec_sub_kernel, g1(np), dim3(256), 0, s, t2, ep->nu(), t1, n, np);   // b = nu - x
    } else {
        gpi_forward_solve_vec(ctx, ep->L, np, ep->ldl, ep->dinv, t1, t2);
        gpi_back_solve_vec(ctx, ep->L, np, ep->ldl, ep->dinv, t2, t1);
        hipLaunchKernelGGL(ep_w_kernel, g1(np), dim3(256), 0, s, t2, ep->nu(), ep->st(), t1, n, np);    // b = nu - st o z
        // R = b b^T - S^1/2 (L L^T)^-1 S^1/2 :  (L L^T)^-1 = T T^T with T = L^-T
        double *T, *Binv;
        GP_TRY(gpi_ws_get(ctx, WS_VT, sizeof(double) * (size_t)np * np, &T));
        GP_TRY(gpi_ws_get(ctx, WS_D, sizeof(double) * (size_t)np * np, &Binv));
        gpi_inverse_transpose_lower(ctx, T, ep->L, np, ep->ldl, ep->dinv);
        gpk_gemm_nt(s, np, np, np, 1.0, T, np, T, np, 0.0, Binv, np, 1, 1);
        hipLaunchKernelGGL(scale_sym_lower_kernel, dim3(8, n < 65535 ? n : 65535), dim3(256), 0, s, Binv, ep->st(), n, np);
        Kinv = Binv;
    }
    gpk_lml_grad_traces(s, dX, n, d, n, theta, t2, Kinv, np, parts, dres);    // g_p = 0.5 tr(R C_p), all P parameters  :60-65
    return gpi_download_2d(ctx, grad, P, dres, P, P, 1);
}

gp_status gp_ep_get(gp_ep *ep, int what, double *out, int ld) {
    if (!ep || !out) return GP_EINVAL;
    gp_ctx *ctx = ep->ctx;
    const int n = ep->n, np = ep->np;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    switch (what) {
        case GP_EP_GET_L: GP_REQUIRE(ctx, ld >= n, "ld < n"); return gpi_download_2d(ctx, out, ld, ep->L, ep->ldl, n, n);
        case GP_EP_GET_SIGMA:
            GP_REQUIRE(ctx, ld >= n, "ld < n");
            if (!ep->sig_mirrored) {   // a sweep keeps the lower triangle only
                hipLaunchKernelGGL(mirror_lower_kernel, dim3(np / 64, np / 64), dim3(256), 0, ctx->stream, ep->Sig, np);
                hipLaunchKernelGGL(mirror_diag_kernel, dim3(np / 64), dim3(256), 0, ctx->stream, ep->Sig, np);
                ep->sig_mirrored = true;
            }
            return gpi_download_2d(ctx, out, ld, ep->Sig, np, n, n);
        case GP_EP_GET_MU: return gpi_download_2d(ctx, out, n, ep->mu(), np, n, 1);
        case GP_EP_GET_CAV_TAU: return gpi_download_2d(ctx, out, n, ep->cav_tau(), np, n, 1);
        case GP_EP_GET_CAV_NU: return gpi_download_2d(ctx, out, n, ep->cav_nu(), np, n, 1);
        default: GP_SET_ERR(ctx, "unknown selector %d", what); return GP_EINVAL;
    }
}

gp_status gp_ep_predict(gp_ep *ep, const double *Ks, int m, int ldks, const double *kss_diag, double *prob) {
    if (!ep) return GP_EINVAL;
    gp_ctx *ctx = ep->ctx;
    GP_REQUIRE(ctx, Ks && kss_diag && prob && m >= 0 && ldks >= m, "bad arguments");
    GP_REQUIRE(ctx, ep->sweeps > 0, "classifier not trained: run gp_ep_sweep first");
    if (m == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int n = ep->n, np = ep->np, mp = gp_pad(m);
    double *Vt, *partial, *sumsq, *dk, *dout;
    GP_TRY(gpi_ws_get(ctx, WS_VT, sizeof(double) * (size_t)mp * np, &Vt));
    GP_TRY(gpi_ws_get(ctx, WS_PARTIAL, sizeof(double) * (size_t)16 * (mp > np ? mp : np), &partial));
    GP_TRY(gpi_ws_get(ctx, WS_SUMSQ, sizeof(double) * (size_t)mp, &sumsq));
    GP_TRY(gpi_ws_get(ctx, WS_A, sizeof(double) * (size_t)mp, &dk));
    GP_TRY(gpi_ws_get(ctx, WS_C, sizeof(double) * (size_t)2 * mp, &dout));
    // z = st o L^T \ (L \ (st o (K nu)))                                  GpClassifier.scala:33-36
    gpk_gemv_rows(s, ep->K, n, n, np, ep->nu(), ep->tmp1(), partial, 16);
    hipLaunchKernelGGL(vec_mul_kernel, g1(n), dim3(256), 0, s, ep->tmp1(), ep->tmp1(), ep->st(), n);
    gpi_forward_solve_vec(ctx, ep->L, np, ep->ldl, ep->dinv, ep->tmp1(), ep->tmp2());
    gpi_back_solve_vec(ctx, ep->L, np, ep->ldl, ep->dinv, ep->tmp2(), ep->tmp1());
    hipLaunchKernelGGL(ep_w_kernel, g1(np), dim3(256), 0, s, ep->tmp2(), ep->nu(), ep->st(), ep->tmp1(), n, np);   // w = nu - z
    // test-train block: fMean = K* w; V = L \ (st o K*^T) as rows; var = kss - |v|^2    :37-45
    GP_HIP(ctx, hipMemsetAsync(Vt, 0, sizeof(double) * (size_t)mp * np, s));
    GP_TRY(gpi_upload_2d(ctx, Vt, mp, Ks, ldks, m, n));
    GP_HIP(ctx, hipMemcpyAsync(dk, kss_diag, sizeof(double) * m, hipMemcpyHostToDevice, s));
    gpk_gemv_rows(s, Vt, m, n, mp, ep->tmp2(), dout, partial, 16);
    hipLaunchKernelGGL(scale_cols_kernel, dim3(1024), dim3(256), 0, s, Vt, mp, Vt, mp, ep->st(), mp, np);
    GP_HIP(ctx, hipMemsetAsync(sumsq, 0, sizeof(double) * mp, s));
    gpi_solve_rows_lower(ctx, Vt, mp, ep->L, np, ep->ldl, ep->dinv, sumsq, nullptr, nullptr);
    hipLaunchKernelGGL(ep_prob_kernel, g1(m), dim3(256), 0, s, dout + mp, dout, sumsq, dk, m);
    return gpi_download_2d(ctx, prob, m, dout + mp, m, m, 1);
}

// EP log marginal likelihood at B hyper-parameter settings of the ARD-RBF kernel: what
// MeshHyperParamsLogLikelihoodEvaluator.recEvaluate (gp/classification/MeshHyperParamsLogLikelihoodEvaluator.scala:26-40) asks of
// MarginalLikelihoodEvaluator.logLikelihoodWithoutGrad (MarginalLikelihoodEvaluator.scala:24-31) one setting at a time, returned by
// setting index (the reference's map is mis-keyed).  Per setting: Gram on the device, EP sweeps until
// AvgBasedStopCriterion(stop_eps) holds (EpParameterEstimator.scala:187-202; sweep 0 always runs; stop_eps < 0: exactly
// max_sweeps sweeps), EP LML (strict: as compiled).  Settings are independent: a few run concurrently, each on its own context.
static gp_status ep_eval_batched(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *thetas, int B,
                                 double stop_eps, int max_sweeps, int strict, double *lml, double *grad, int *sweeps, int *info);

gp_status gp_ep_lml_rbf_batched(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *thetas, int B,
                                double stop_eps, int max_sweeps, int strict, double *lml, int *sweeps, int *info) {
    return ep_eval_batched(ctx, X, n, d, ldx, y, thetas, B, stop_eps, max_sweeps, strict, lml, nullptr, sweeps, info);
}

// MarginalLikelihoodEvaluator.logLikelihood (gp/classification/MarginalLikelihoodEvaluator.scala:33-44) at B settings: EP log
// marginal likelihood AND its gradient w.r.t. all d+2 hyper-parameters (:46-66), grad B x (d+2) row-major.
gp_status gp_ep_lml_grad_rbf_batched(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *thetas, int B,
                                     double stop_eps, int max_sweeps, int strict, double *lml, double *grad, int *sweeps, int *info) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, grad, "null gradient");
    return ep_eval_batched(ctx, X, n, d, ldx, y, thetas, B, stop_eps, max_sweeps, strict, lml, grad, sweeps, info);
}

// GradientHyperParamsOptimizer.optimizeHyperParams (gp/classification/HyperParamsOptimization.scala:31-55): maximise the EP log
// marginal likelihood over all d+2 hyper-parameters with the gradient-based optimiser it is wired with -- BreezeLbfgsOptimizer
// (optimization/Optimization.scala:30-63: L-BFGS m = history, maxIter, best-seen point).  Every objective evaluation is a full
// EP run (Gram -> sweeps until AvgBasedStopCriterion(stop_eps) or max_sweeps -> LML + gradient); the trial steps of one line
// search are independent EP problems and run concurrently on the context's helper contexts.
gp_status gp_ep_optimize_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *theta0, double stop_eps,
                             int max_sweeps, int strict, int max_iter, int history, double *theta_out, double *lml_out, int *iters_out,
                             int *evals_out) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, X && y && theta0 && theta_out, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && ldx >= n && max_sweeps >= 1 && max_iter >= 0 && history >= 1, "bad arguments");
    const int P = d + 2;
    constexpr int NC = 3;   // trial steps per iteration (evaluated by ep_eval_batched: concurrently where that pays, GPCORE_EP_WORKERS)
    auto evaluate = [&](const double *thetas, int count, double *f, double *g, int *bad) -> gp_status {
        return ep_eval_batched(ctx, X, n, d, ldx, y, thetas, count, stop_eps, max_sweeps, strict, f, g, nullptr, bad);
    };
    return gpi_lbfgs_maximize(ctx, P, P, theta0, max_iter, history, NC, evaluate, theta_out, lml_out, iters_out, evals_out);
}

// B settings through lockstep sweeps.  G problems (GPCORE_EP_GROUP, default 12, capped by B and by free device memory: 5 np^2 doubles
// each) live in one slab; a problem leaves when its stop criterion holds (AvgBasedStopCriterion as written, host side, per problem
// per sweep) or at max_sweeps, gets its LML (and gradient) from the single-problem entry points on its view of the slab, and its
// slot takes the next setting -- problems in one launch need not be at the same sweep.  Towards the end the slots above the last
// active one drop out of the launches; a finished problem below it keeps sweeping (its results are already out).
static gp_status ep_eval_lockstep(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *thetas, int B,
                                  double stop_eps, int max_sweeps, int strict, double *lml, double *grad, int *sweeps, int *info) {
    const int P = d + 2, np = gp_pad(n);
    GP_HIP(ctx, hipSetDevice(ctx->device));
    int G = 12;
    if (const char *e = getenv("GPCORE_EP_GROUP")) G = atoi(e);
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const double per = 5.2 * 8.0 * (double)np * np;
            G = std::min<long long>(G, (long long)(0.8 * (double)free_b / per));
        }
    }
    G = std::max(1, std::min(G, B));
    ep_slab sl;
    GP_TRY(ep_slab_alloc(ctx, n, y, G, sl));
    double *dX = nullptr;
    gp_status st = gpi_ws_get(ctx, WS_A, sizeof(double) * (size_t)n * d, &dX);
    if (st == GP_OK) st = gpi_upload_2d(ctx, dX, n, X, ldx, n, d);
    if (st == GP_OK) gpk_centroid(ctx->stream, dX, n, d, n, gp_gram_center(ctx));      // one centre for every setting of the call
    std::vector<int> slot_b(G, -1), slot_j(G, 0);
    std::vector<double> cur((size_t)G * 2 * n, 0.0), old((size_t)G * 2 * n, 0.0), pull((size_t)G * 2 * np);
    std::vector<int> hinfo(G, 0);
    int next = 0, active = 0;
    auto load = [&](int g) -> gp_status {      // the next setting into slot g: Gram on the device, sites zeroed, Sigma = K
        slot_b[g] = -1;
        if (next >= B) return GP_OK;
        const int b = next++;
        gp_ep *v = sl.ep[g];
        gpk_gram_sym(ctx->stream, dX, n, d, n, thetas + (size_t)b * P, v->K, np, 1, 0.0, gp_gram_flag(ctx), gp_gram_center(ctx));
        GP_TRY(ep_start(v));
        slot_b[g] = b, slot_j[g] = 0;
        std::fill(cur.begin() + (size_t)g * 2 * n, cur.begin() + (size_t)(g + 1) * 2 * n, 0.0);
        ++active;
        return GP_OK;
    };
    for (int g = 0; g < G && st == GP_OK; ++g) st = load(g);
    while (st == GP_OK && active > 0) {
        int count = 0;
        for (int g = 0; g < G; ++g) if (slot_b[g] >= 0) count = g + 1;
        st = ep_sweep_lockstep(sl, count);
        if (st != GP_OK) break;
        // tau | nu of every problem (the first 2 np doubles of its vector block) and the failing-pivot words, after ONE synchronisation
        hipError_t e = hipMemcpy2DAsync(pull.data(), sizeof(double) * 2 * np, sl.ep[0]->vec, sizeof(double) * sl.sVec(), sizeof(double) * 2 * np, count,
                                        hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(hinfo.data(), sl.info, sizeof(int) * count, hipMemcpyDeviceToHost, ctx->stream);
        int ferr = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&ferr, sl.ep[0]->flags + np / GP_NB, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { GP_SET_ERR(ctx, "EP lockstep batch: %s", hipGetErrorString(e)); st = GP_EHIP; break; }
        if (ferr) { GP_SET_ERR(ctx, "EP lockstep sweep: a device flag of the fused chain did not arrive within its bound (problem %d)", ferr - 1); st = GP_EHIP; break; }
        for (int g = 0; g < count && st == GP_OK; ++g) {
            const int b = slot_b[g];
            if (b < 0) continue;
            double *tn = cur.data() + (size_t)g * 2 * n, *to = old.data() + (size_t)g * 2 * n;
            std::copy(tn, tn + 2 * n, to);
            std::copy(pull.begin() + (size_t)g * 2 * np, pull.begin() + (size_t)g * 2 * np + n, tn);                 // tau
            std::copy(pull.begin() + (size_t)g * 2 * np + np, pull.begin() + (size_t)g * 2 * np + np + n, tn + n);   // nu
            const int j = ++slot_j[g];
            bool done = j >= max_sweeps, bad = hinfo[g] != 0;
            if (!done && !bad && stop_eps >= 0.0) {   // avgBetweenSiteParams :195-202: (sum / 2) * n, precedence as written
                double sum = 0.0;
                for (int i = 0; i < n; ++i) sum = sum + (tn[n + i] - to[n + i]) + (tn[i] - to[i]);
                done = std::fabs(sum / 2 * n) < stop_eps;
            }
            if (!done && !bad) continue;
            gp_ep *v = sl.ep[g];
            if (bad) {
                lml[b] = NAN;
                if (grad) for (int q = 0; q < P; ++q) grad[(size_t)b * P + q] = NAN;
                if (info) info[b] = hinfo[g];
            } else {
                st = gp_ep_lml(v, strict, lml + b);
                if (st == GP_OK && grad) st = ep_lml_grad_dev(v, dX, d, thetas + (size_t)b * P, strict, grad + (size_t)b * P);
                if (info) info[b] = 0;
            }
            if (sweeps) sweeps[b] = j;
            --active;
            if (st == GP_OK) st = load(g);
            if (st == GP_OK && slot_b[g] < 0 && bad) st = ep_start(v);     // a failed problem that stays in the launches: back to a benign state
        }
    }
    ep_slab_free(sl);
    return st;
}

static gp_status ep_eval_batched(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *thetas, int B,
                                 double stop_eps, int max_sweeps, int strict, double *lml, double *grad, int *sweeps, int *info) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, X && y && thetas && lml, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && ldx >= n && B >= 0 && max_sweeps >= 1, "bad dimensions");
    if (B == 0) return GP_OK;
    const int P = d + 2;
    {   // lockstep batch (ep_sweep_lockstep) whenever there are two settings and two site blocks; GPCORE_EP_LOCKSTEP = 0 selects the
        // one-setting-at-a-time path below.  12 settings x 10 sweeps, aggregate sweeps/s one at a time -> lockstep (profiles/r03_m_ep_mesh_perf.log,
        // r03_k_ep_mesh_perf2.log): n = 512 1778 -> 12110, n = 1024 831 -> 6081, n = 1536 570 -> 3028, n = 2048 416 -> 1600, n = 4096 184 -> 244
        // (44.7 TFLOP/s executed: bound by the batched GEMMs, kernel stats in profiles/r03_k_lockstep_kernel_stats.csv), n = 8192 28.9 -> 30.5
        const char *e = getenv("GPCORE_EP_LOCKSTEP");
        const bool lock = e ? atoi(e) != 0 : B >= 2;
        if (lock && gp_pad(n) >= 2 * GP_NB)
            return ep_eval_lockstep(ctx, X, n, d, ldx, y, thetas, B, stop_eps, max_sweeps, strict, lml, grad, sweeps, info);
    }
    // Concurrent EP problems, each on its own context: with the streamed refactorisation one run keeps four streams busy and a second
    // one only gets in its way (12 settings x 10 sweeps, aggregate sweeps/s with 1 / 2 / 3 workers: n = 2048 397 / 171 / 184, n = 4096
    // 177 / 69 / 80); the end-of-sweep form of the small problems still gains from a second run (n = 512 1270 / 1499 / 1061, n = 1024
    // 641 / 827 / 383).  GPCORE_EP_WORKERS overrides.
    int nw = gp_pad(n) > 1024 ? 1 : 2;
    if (const char *e = getenv("GPCORE_EP_WORKERS")) nw = atoi(e);
    nw = std::max(1, std::min(std::min(nw, 8), B));
    std::vector<gp_ctx *> ctxs(nw, nullptr);
    ctxs[0] = ctx;
    for (int k = 1; k < nw; ++k) {
        ctxs[k] = gpi_child_ctx(ctx, k - 1);
        if (!ctxs[k]) { nw = k; break; }
    }
    std::vector<gp_status> sts(nw, GP_OK);
    std::atomic<int> next{0};
    auto run = [&](int k) {
        gp_ctx *c = ctxs[k];
        gp_ep *ep = nullptr;
        gp_status st = ep_alloc(c, n, y, &ep);
        double *dX = nullptr;
        if (st == GP_OK) st = gpi_ws_get(c, WS_A, sizeof(double) * (size_t)n * d, &dX);
        if (st == GP_OK) st = gpi_upload_2d(c, dX, n, X, ldx, n, d);
        if (st == GP_OK) gpk_centroid(c->stream, dX, n, d, n, gp_gram_center(c));
        std::vector<double> tau(n), nu(n), tau_old(n), nu_old(n);
        while (st == GP_OK) {
            const int b = next.fetch_add(1);
            if (b >= B) break;
            gpk_gram_sym(c->stream, dX, n, d, n, thetas + (size_t)b * P, ep->K, ep->np, 1, 0.0, gp_gram_flag(c), gp_gram_center(c));
            st = ep_start(ep);
            int j = 0, h = 0;
            std::fill(tau.begin(), tau.end(), 0.0);
            std::fill(nu.begin(), nu.end(), 0.0);
            gp_status es = GP_OK;
            while (st == GP_OK && es == GP_OK) {
                tau_old = tau, nu_old = nu;
                es = gp_ep_sweep(ep, 1, tau.data(), nu.data(), &h);
                if (es != GP_OK) break;
                ++j;
                if (j >= max_sweeps) break;
                if (stop_eps >= 0.0) {   // avgBetweenSiteParams :195-202: (sum / 2) * n, precedence as written
                    double sum = 0.0;
                    for (int i = 0; i < n; ++i) sum = sum + (nu[i] - nu_old[i]) + (tau[i] - tau_old[i]);
                    if (std::fabs(sum / 2 * n) < stop_eps) break;
                }
            }
            if (es == GP_ENOTPD) {
                lml[b] = NAN;
                if (grad) for (int q = 0; q < P; ++q) grad[(size_t)b * P + q] = NAN;
                if (info) info[b] = h;
            } else if (es != GP_OK) {
                st = es;
            } else if (st == GP_OK) {
                st = gp_ep_lml(ep, strict, lml + b);
                if (st == GP_OK && grad) st = ep_lml_grad_dev(ep, dX, d, thetas + (size_t)b * P, strict, grad + (size_t)b * P);
                if (info) info[b] = 0;
            }
            if (sweeps) sweeps[b] = j;
        }
        if (ep) gp_ep_destroy(ep);
        sts[k] = st;
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < nw; ++k) threads.emplace_back(run, k);
    run(0);
    for (auto &t : threads) t.join();
    for (int k = 0; k < nw; ++k)
        if (sts[k] != GP_OK) {
            if (k > 0) GP_SET_ERR(ctx, "%s", ctxs[k]->err);
            return sts[k];
        }
    return GP_OK;
}

#ifdef EP_STAMPS
gp_status gp_debug_ep_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(ep_stamps), sizeof(unsigned long long) * 4 * GP_NB) == hipSuccess ? GP_OK : GP_EHIP;
}
#endif

void gp_ep_destroy(gp_ep *ep) {
    if (!ep) return;
    if (!ep->owns) { delete ep; return; }
    if (ep->ctx) {
        (void)hipSetDevice(ep->ctx->device);
        for (hipStream_t st : {ep->ctx->stream, ep->ctx->side, ep->ctx->side2, ep->ctx->side3}) if (st) (void)hipStreamSynchronize(st);
    }
    for (hipEvent_t ev : ep->ev) if (ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : {ep->ev_chol, ep->ev_parta, ep->ev_partb, ep->ev_w, ep->ev_pipe}) if (ev) (void)hipEventDestroy(ev);
    void *ptrs[] = {ep->K, ep->Sig, ep->Sig2, ep->L, ep->dinv, ep->S, ep->Sc, ep->blk, ep->vec, ep->cvec, ep->y, ep->flags, ep->uflags};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    delete ep;
}

}  // extern "C"
