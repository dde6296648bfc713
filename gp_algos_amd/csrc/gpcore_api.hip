// libgpcore.so -- C-ABI implementation (host orchestration over the HIP kernels).  gfx950 only.
// Reference call stacks being replaced: SURVEY.md section 3; per-function citations in include/gpcore.h.
#include "gpcore_internal.h"

#include <algorithm>
#include <atomic>
#include <queue>
#include <functional>
#include <new>
#include <thread>

// ------------------------------------------------------------------------------------------------
// profiling: HIP events on the context's stream around one kernel class
// ------------------------------------------------------------------------------------------------
void gp_prof_begin(gp_ctx *ctx, int cls, hipStream_t s) {
    if (!(ctx->prof_which & (1 << cls))) return;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return;
    (void)hipEventRecord(e0, s ? s : ctx->stream);
    ctx->prof[cls].ev.push_back(e0);
    ctx->prof[cls].ev.push_back(e1);
}
void gp_prof_end(gp_ctx *ctx, int cls, double work, hipStream_t s) {
    if (!(ctx->prof_which & (1 << cls))) return;
    gp_prof_slot &p = ctx->prof[cls];
    if (p.ev.size() < 2) return;
    (void)hipEventRecord(p.ev.back(), s ? s : ctx->stream);
    p.launches += 1;
    p.work += work;
}

namespace {

// tag: what a caller that wants to find its own invariants again next time wrote there (0 = nothing promised; ws_get resets it)
struct ws_slot { void *p = nullptr; size_t bytes = 0; unsigned long long tag = 0; };

struct mega_task { int type, k0, kb, i, j, q, dep[10]; };
struct mega_plan { int np = 0, extra = 0, out_blocks = 0, workers = 0; double model_us = 0.0; std::vector<mega_task> tasks; };
// the persistent-launch factorisation's state: the plan of the shape used last, its copy on the device, the flags
struct mega_state { mega_plan plan; int *d_tasks = nullptr, *d_done = nullptr, *d_ctl = nullptr; double *d_sums = nullptr; int cap = 0, epoch = 0, sums_cap = 0, seen_np = 0, seen_extra = -1; };

struct ctx_ext { ws_slot ws[WS_COUNT]; std::vector<gp_ctx *> children; mega_state mega; };

// gp_ctx owns a ctx_ext through this side table (keeps the header struct POD-ish)
ctx_ext *ext_of(gp_ctx *ctx);

}  // namespace

struct gp_ctx_full : gp_ctx { ctx_ext ext; };
namespace {
ctx_ext *ext_of(gp_ctx *ctx) { return &static_cast<gp_ctx_full *>(ctx)->ext; }

gp_status ws_get(gp_ctx *ctx, int slot, size_t bytes, double **out) {
    ws_slot &w = ext_of(ctx)->ws[slot];
    w.tag = 0;
    if (w.bytes < bytes) {
        if (w.p) { GP_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(w.p); w.p = nullptr; w.bytes = 0; }
        size_t want = bytes + bytes / 8;
        hipError_t e = hipMalloc(&w.p, want);
        if (e != hipSuccess) { w.p = nullptr; GP_SET_ERR(ctx, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return GP_ENOMEM; }
        w.bytes = want;
        GP_HIP(ctx, hipMemsetAsync(w.p, 0, want, ctx->stream));
    }
    *out = static_cast<double *>(w.p);
    return GP_OK;
}

// ws_get for a caller whose buffer keeps an invariant between calls (zeros that no kernel of its path ever overwrites): *kept says
// whether the slot still carries `tag` from the same caller -- nobody else has asked for the slot since and it was not reallocated.
gp_status ws_get_keep(gp_ctx *ctx, int slot, size_t bytes, double **out, unsigned long long tag, bool *kept) {
    ws_slot &w = ext_of(ctx)->ws[slot];
    const unsigned long long before = w.tag;
    const void *pb = w.p;
    GP_TRY(ws_get(ctx, slot, bytes, out));
    *kept = tag != 0 && before == tag && pb == w.p;
    w.tag = tag;
    return GP_OK;
}
void ws_forget(gp_ctx *ctx, int slot) { ext_of(ctx)->ws[slot].tag = 0; }

gp_status upload_2d(gp_ctx *ctx, double *dst, int ldd, const double *src, int lds, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return GP_OK;
    GP_HIP(ctx, hipMemcpy2DAsync(dst, (size_t)ldd * 8, src, (size_t)lds * 8, (size_t)rows * 8, cols, hipMemcpyHostToDevice, ctx->stream));
    return GP_OK;
}
gp_status download_2d(gp_ctx *ctx, double *dst, int ldd, const double *src, int lds, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return GP_OK;
    GP_HIP(ctx, hipMemcpy2DAsync(dst, (size_t)ldd * 8, src, (size_t)lds * 8, (size_t)rows * 8, cols, hipMemcpyDeviceToHost, ctx->stream));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GP_OK;
}

gp_status read_info(gp_ctx *ctx, int *info) {
    int h = 0, merr = 0;
    GP_HIP(ctx, hipMemcpyAsync(&h, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    mega_state &ms = ext_of(ctx)->mega;
    if (ms.d_ctl) GP_HIP(ctx, hipMemcpyAsync(&merr, ms.d_ctl + 1, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (merr != 0) {   // a workgroup of chol_mega_kernel gave up waiting for a task it depends on: never expected, reported rather than hung
        (void)hipMemsetAsync(ms.d_ctl + 1, 0, sizeof(int), ctx->stream);
        if (info) *info = 0;
        GP_SET_ERR(ctx, "Cholesky (single launch): task %d waited for a dependency beyond its bound", merr - 1);
        return GP_EHIP;
    }
    if (info) *info = h;
    return GP_OK;
}

double trapezoid_flops(int M, int N, int K) {  // lower-trapezoid tiles (bi >= bj), full diagonal tiles
    double nbm = M / (double)GP_NB, nbn = N / (double)GP_NB;
    return (nbn * nbm - nbn * (nbn - 1) / 2.0) * 2.0 * GP_NB * GP_NB * (double)K;
}
double syrk_flops(int r, int K) { return trapezoid_flops(r, r, K); }

bool small_panel_update() {   // GPCORE_PANEL_SMALL=0: the general 128 x 128-tile kernel for the in-panel updates too
    const char *e = getenv("GPCORE_PANEL_SMALL");
    return !e || atoi(e) != 0;
}

double small_update_tiles() {   // outer updates with fewer 128 x 128 tiles than this use the 64 x 64 kernel (GPCORE_SMALL_UPDATE_TILES)
    const char *e = getenv("GPCORE_SMALL_UPDATE_TILES");
    return e ? atof(e) : 600.0;   // measured at 0 / 150 / 300 / 600 / 1200: n = 8192 fit 7.04 / 6.65 / 6.64 / 6.63 / 6.72 ms
}

// C (M x N, rectangular) -= A (M x K) B (N x K)^T for a SINGLE problem: launches with few 128 x 128 tiles go to the 64 x 64 kernel
void gemm_sub_single(hipStream_t s, int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc) {
    if ((double)(M / GP_NB) * (N / GP_NB) < small_update_tiles() && small_panel_update()) gpk_gemm_k128_sub(s, M, N, A, lda, B, ldb, C, ldc, 0, K);
    else gpk_gemm_nt(s, M, N, K, -1.0, A, lda, B, ldb, 1.0, C, ldc, 0);
}

// Two-level blocked right-looking Cholesky of the padded np x np matrix A (lower):
//   inner (nb = 128), confined to one outer panel of GP_OUTER = 512 columns:
//       potrf_diag128(Akk) -> Lkk + 16x16 tile inverses;  A21 <- A21 Lkk^-T (MFMA trsm panel, all rows below);
//       the rest of the OUTER PANEL's columns -= A21 A21^T (narrow MFMA update, K = 128)
//   outer: trailing matrix -= P P^T with K = 512 (the high-intensity MFMA syrk this design is built around), one launch.
//       Look-ahead (ctx->lookahead; by default ON for single factorisations with 4096 <= rows and np <= 16384, see below): split
//       into (a) the next outer panel's columns on the main stream and (b) everything to the right of it on the CU-masked side
//       stream, under the next panel's chain of single-workgroup kernels (which need a whole CU's LDS: the mask keeps CUs free).
// `extra` (0 or GP_NB) rows below the matrix ride along: with y^T in row np this leaves (L^-1 y)^T there.
// count > 1: a lockstep batch -- problem g lives at A + g*strideA, dinv + g*strideDinv, info + g; every launch covers all of them.
// ---- the factorisation of ONE matrix as one persistent launch (chol_mega_kernel, kernels_diag.hip) ----
// Task list of the two-level right-looking factorisation (inner 128, outer panels of `out_blocks` block columns), in a LIST-SCHEDULE
// order: the tasks are generated in the order chol_blocked launches them, a critical-path-first schedule of that DAG on `workers`
// workgroups is simulated with a cost model (us per task on one CU, measured with tools/mega_trace.py: the constants below; a hand-over
// between workgroups 3), and the list is the tasks in the order the simulation starts them.  Workgroups claim tasks in list order, so
// on the chip the dependencies of a claimed task are mostly met already.  Every dependency points BACKWARDS in the list (checked):
// no deadlock.  The model is accurate -- with the measured task costs its makespan at n = 8192 is 4.79 ms against 4.7-4.85 measured --
// so it also answers what-ifs without a GPU (gp_chol_plan_info with GPCORE_MEGA_COSTS): K = 512 tile at 65 instead of 77.7 us 4.33 ms,
// the link at half its time 4.71 ms, tiles for free 3.86 ms.
// What the static order leaves on the table, and what did not recover it (round 4; the traces are tools/mega_trace.py's, the scheduling
// variants are kept as tools/lab/mega_claim_ahead.patch):
//   - the mean step from one diagonal block to the next is ~75 us against ~57 of task bodies: in some phases of the factorisation the
//     chip is behind the model and the chain's tasks are claimed 10-60 us after their dependencies were met, in others it is ahead and
//     they are claimed up to 400 us early (profiles/r04_i_mega_chain_detail.log);
//   - the chain's tasks a fixed lead (0..200 us of model time) earlier in the list: 4.99-5.09 ms, the lag is not a constant
//     (profiles/r04_k_fit_mega_lead_sweep.log);
//   - the chain's tasks in a queue of their own whose head is taken, by compare-and-swap, by the first workgroup that finds it READY:
//     nobody spins, but the head moves one task per poll of one workgroup (~2 us) and the 33 quarters behind a panel boundary are taken
//     over 66 us: 5.63 ms (profiles/r04_j_*);
//   - that queue claimed a bounded number of tasks AHEAD (4..64 outstanding) by workgroups that then wait: 5.27-6.03 ms, every
//     outstanding task is a CU spinning and the late dependencies are one level further out, in the bulk (profiles/r04_l_*, r04_m_*:
//     with every in-panel task of the next 2 / 4 / 8 block rows in the chain queue 5.47 / 5.53 / 5.65 ms);
//   - every block's link given an early phase INSIDE its task (the previous panel's earlier block columns applied before block k-1's
//     factor arrives): 4.92-5.10 ms, no better (profiles/r04_i_fit_mega_lead_order.log).  What is kept instead is that early part as
//     tasks of their own (type 5, below): the gap across a panel boundary fell from 85 to 15 us, the gaps inside panels rose from 8 to
//     18 us, the mean step stayed -- 2-7 % at n = 5120 ... 14336 (profiles/r04_q_*);
//   - pairs of tiles as 256 x 128 tasks: in the model 5.35 ms at n = 8192 (coarser tasks stand in the chain's way), 4.69 when only
//     columns a panel or more to the right are paired; not built.
bool chol_mega_plan(int np, int extra, int out_blocks, int workers, mega_plan &plan) {
    const int nb = np / GP_NB, nrow = (np + extra) / GP_NB;
    struct node { mega_task t; double cost; std::vector<int> deps; };
    std::vector<node> g;
    g.reserve((size_t)nb * nb);
    std::vector<int> trsm_of((size_t)nb * nrow, -1), potrf_of(nb, -1);
    // last writers of a tile: the task(s) whose result the next reader / writer of tile (i, j) must wait for (up to 4 quarters)
    std::vector<std::vector<int>> last((size_t)nrow * nb);
    auto tile = [&](int i, int j) -> std::vector<int> & { return last[(size_t)i * nb + j]; };
    // us per task on one CU, from tools/mega_trace.py (profiles/r04_q_mega_trace_boundary_sums.log): body + publish
    // (the list is not sensitive to them: with every cost 10 % lower or higher, or the link at 45 instead of 30, n = 8192 refits in
    // 4.80-4.90 ms, profiles/r04_r_fit_mega_cost_model_sweep.log)
    double c_potrf = 27.5, c_trsm = 20.8, c_k128 = 25.1, c_k512 = 77.7, c_q128 = 9.0, c_q512 = 27.8, c_link = 30.0;
    if (const char *e = getenv("GPCORE_MEGA_COSTS"))      // lab: "potrf,trsm,k128,k512,q128,q512,link" -- what-if runs of the model (gp_chol_plan_info)
        sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf,%lf", &c_potrf, &c_trsm, &c_k128, &c_k512, &c_q128, &c_q512, &c_link);
    auto add = [&](int type, int k0, int kb, int i, int j, int q, double cost, std::vector<int> deps) {
        node nd;
        nd.t = mega_task{type, k0, kb, i, j, q, {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1}};
        nd.cost = cost;
        std::sort(deps.begin(), deps.end());
        deps.erase(std::unique(deps.begin(), deps.end()), deps.end());
        nd.deps = deps;
        g.push_back(nd);
        return (int)g.size() - 1;
    };
    std::vector<int> pre_x;       // writers of tile (k+1, k) before its rows are solved: what the fused task of block k+1 waits for
    std::vector<int> early;       // the type-5 quarters of the next panel's first diagonal tile
    for (int c0 = 0; c0 < nb; c0 += out_blocks) {
        const int c1 = std::min(nb, c0 + out_blocks);
        for (int k = c0; k < c1; ++k) {
            if (k == 0) potrf_of[k] = add(0, k, 1, k, k, -1, c_potrf, tile(k, k));
            else {
                // every later block: the link to block k-1 (solve of rows k of block column k-1, the update of tile (k, k) that those rows
                // complete) is the prologue of the diagonal block's task; the place holder made in step k-1 stands for the solved rows.
                // Inside a panel that update is the in-panel one (K = 128, summed from zero).  For a panel's FIRST block it is the previous
                // panel's outer update of the tile: its earlier block columns were summed ahead of time by the type-5 quarters made in
                // step k-2 (slot = the panel's number), the task continues that sum (field j = the slot).
                std::vector<int> d = tile(k, k);
                d.insert(d.end(), pre_x.begin(), pre_x.end());
                d.push_back(potrf_of[k - 1]);
                int slot = -1;
                if (k == c0 && !early.empty()) { slot = c0 / out_blocks; d.insert(d.end(), early.begin(), early.end()); early.clear(); }
                potrf_of[k] = add(4, k, 1, k, slot, trsm_of[(size_t)(k - 1) * nrow + k], c_link + c_potrf, d);
                g[trsm_of[(size_t)(k - 1) * nrow + k]].t.kb = potrf_of[k];      // the place holder remembers who announces it
            }
            tile(k, k) = {potrf_of[k]};
            for (int i = k + 1; i < nrow; ++i) {
                if (i == k + 1 && i < nb) {
                    pre_x = tile(i, k);
                    trsm_of[(size_t)k * nrow + i] = add(3, k, -1, i, k, -1, -1.0, {});
                    tile(i, k) = {trsm_of[(size_t)k * nrow + i]};
                    continue;
                }
                std::vector<int> d = tile(i, k);
                d.push_back(potrf_of[k]);
                trsm_of[(size_t)k * nrow + i] = add(1, k, 1, i, k, -1, c_trsm, d);
                tile(i, k) = {trsm_of[(size_t)k * nrow + i]};
            }
            if (k == c1 - 2 && c1 < nb) {
                // the next panel's first diagonal tile: all but the last block column of ITS outer update can be summed now (rows c1 of
                // the block columns c0 .. c1-2 are solved), into the panel's scratch tile -- three quarters, (0, 1) is above the diagonal
                const int kb = c1 - 1 - c0;
                std::vector<int> d;
                for (int kk = c0; kk <= k; ++kk) d.push_back(trsm_of[(size_t)kk * nrow + c1]);
                for (int q = 0; q < 4; ++q)
                    if (q != 2) early.push_back(add(5, c0, kb, c1, c1 / out_blocks, q, c_q128 + (c_q512 - c_q128) * (kb - 1) / 3.0, d));
            }
            for (int j = k + 1; j < c1; ++j)
                for (int i = j; i < nrow; ++i) {
                    if (i == k + 1 && j == k + 1) continue;      // the next diagonal tile's update is part of that block's fused task
                    std::vector<int> d = tile(i, j);
                    d.push_back(trsm_of[(size_t)k * nrow + i]);
                    d.push_back(trsm_of[(size_t)k * nrow + j]);
                    tile(i, j) = {add(2, k, 1, i, j, -1, c_k128, d)};
                }
        }
        for (int j = c1; j < nb; ++j)
            for (int i = j; i < nrow; ++i) {
                std::vector<int> d = tile(i, j);
                d.push_back(trsm_of[(size_t)(c1 - 1) * nrow + i]);      // block column c1-1 of row block i is the last of the panel to be solved
                d.push_back(trsm_of[(size_t)(c1 - 1) * nrow + j]);
                const int kb = c1 - c0;
                const double cost = c_k128 + (c_k512 - c_k128) * (kb - 1) / 3.0, qcost = c_q128 + (c_q512 - c_q128) * (kb - 1) / 3.0;
                if (i == c1 && j == c1) continue;      // the next panel's first diagonal tile: type-5 quarters + that block's fused task
                if (i < c1 + out_blocks && j < c1 + out_blocks && i < nb) {      // the next outer panel's diagonal block: quarters
                    std::vector<int> w;
                    for (int q = 0; q < 4; ++q)
                        if (!(i == j && q == 2)) w.push_back(add(2, c0, kb, i, j, q, qcost, d));   // q = (row half) + 2 (column half); (0, 1) is above the diagonal
                    tile(i, j) = w;
                } else tile(i, j) = {add(2, c0, kb, i, j, -1, cost, d)};
            }
    }
    const int n = (int)g.size();
    for (const node &nd : g) if (nd.deps.size() > 10) return false;
    // critical-path priorities and the simulated list schedule
    const double hop = 3.0;
    std::vector<std::vector<int>> succ(n);
    std::vector<int> indeg(n, 0);
    for (int t = 0; t < n; ++t) { indeg[t] = (int)g[t].deps.size(); for (int d : g[t].deps) succ[d].push_back(t); }
    std::vector<double> prio(n, 0.0), est(n, 0.0), start(n, 0.0);
    for (int pass = 0; pass < 2; ++pass)      // twice: a fused task inherits the urgency of what waits for the rows it announces half-way
        for (int t = n - 1; t >= 0; --t) {
            double m = 0.0;
            for (int s2 : succ[t]) m = std::max(m, prio[s2]);
            prio[t] = std::max(g[t].cost, 0.0) + hop + m;
            if (g[t].t.type == 4) prio[t] = std::max(prio[t], c_link + hop + prio[g[t].t.q]);
        }
    typedef std::pair<double, int> pdi;
    std::priority_queue<pdi> ready;                                               // (priority, task): dependencies done
    std::priority_queue<pdi, std::vector<pdi>, std::greater<pdi>> waiting;        // (earliest start, task): done, but the hand-over is still in flight
    std::priority_queue<pdi, std::vector<pdi>, std::greater<pdi>> running;        // (finish time, task)
    for (int t = 0; t < n; ++t) if (indeg[t] == 0 && g[t].t.type != 3) ready.push(pdi(prio[t], t));
    int freew = workers, finished = 0;
    double now = 0.0;
    while (finished < n) {
        while (!waiting.empty() && waiting.top().first <= now + 1e-9) { const int t = waiting.top().second; waiting.pop(); ready.push(pdi(prio[t], t)); }
        while (freew > 0 && !ready.empty()) {
            const int t = ready.top().second;
            ready.pop();
            start[t] = now;
            running.push(pdi(now + g[t].cost, t));
            --freew;
            if (g[t].t.type == 4) {      // its place holder: announced c_link into the task, costs nobody a workgroup
                const int ph = g[t].t.q;
                start[ph] = now + 1e-3;
                running.push(pdi(now + c_link, ph));
            }
        }
        double next = running.empty() ? 1e300 : running.top().first;
        if (freew > 0 && !waiting.empty()) next = std::min(next, waiting.top().first);
        if (next >= 1e300) return false;      // not reached for a valid DAG
        now = next;
        while (!running.empty() && running.top().first <= now + 1e-9) {
            const int t = running.top().second;
            running.pop();
            ++finished;
            if (g[t].t.type != 3) ++freew;
            for (int s2 : succ[t]) {
                est[s2] = std::max(est[s2], now + hop);
                if (--indeg[s2] == 0) waiting.push(pdi(est[s2], s2));
            }
        }
    }
    std::vector<int> order(n), pos(n);
    for (int t = 0; t < n; ++t) order[t] = t;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return start[a] < start[b] || (start[a] == start[b] && prio[a] > prio[b]); });
    for (int p2 = 0; p2 < n; ++p2) pos[order[p2]] = p2;
    plan.tasks.resize(n);
    for (int p2 = 0; p2 < n; ++p2) {
        const node &nd = g[order[p2]];
        mega_task t = nd.t;
        for (size_t d = 0; d < nd.deps.size(); ++d) {
            t.dep[d] = pos[nd.deps[d]];
            if (t.dep[d] >= p2) return false;      // a dependency that does not point backwards: never for a schedule (a task starts after its dependencies end)
        }
        if (t.type == 4) t.q = pos[t.q];           // the place holder it announces, by its place in the list
        if (t.type == 3) t.kb = pos[t.kb];
        plan.tasks[p2] = t;
    }
    plan.np = np, plan.extra = extra, plan.out_blocks = out_blocks, plan.workers = workers, plan.model_us = now;
    return true;
}

// Runs the factorisation as one persistent launch on the context's stream; false = not available (allocation, planning): the caller
// falls back to the launch-per-step form.  d_ctl: [0] the claim counter (zeroed in stream order before every launch), [1] the error word.
// A plan costs host time once per shape (list schedule of the whole DAG: 4 / 13 / 68 ms at n = 5120 / 8192 / 14336) and the context keeps
// ONE: a shape is planned when it comes the second time in a row (the first factorisation of a shape, and shapes that alternate,
// take the launch-per-step form, which is 1 ms slower, not 13); GPCORE_CHOL_MEGA=1 plans at once.
bool chol_mega_run(gp_ctx *ctx, double *A, int np, int lda, double *dinv, int extra, int out_blocks, bool plan_at_once) {
    mega_state &ms = ext_of(ctx)->mega;
    const int workers = ctx->num_cu;
    if (ms.plan.np != np || ms.plan.extra != extra || ms.plan.out_blocks != out_blocks || ms.plan.workers != workers) {
        if (!plan_at_once && !(ms.seen_np == np && ms.seen_extra == extra)) {
            ms.seen_np = np, ms.seen_extra = extra;
            return false;
        }
        mega_plan pl;
        if (!chol_mega_plan(np, extra, out_blocks, workers, pl)) return false;
        const int n = (int)pl.tasks.size();
        if (n > ms.cap) {
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
            if (ms.d_tasks) (void)hipFree(ms.d_tasks);
            if (ms.d_done) (void)hipFree(ms.d_done);
            ms.d_tasks = ms.d_done = nullptr, ms.cap = 0;
            if (hipMalloc(&ms.d_tasks, sizeof(mega_task) * (size_t)n) != hipSuccess || hipMalloc(&ms.d_done, sizeof(int) * (size_t)n) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            if (hipMemset(ms.d_done, 0, sizeof(int) * (size_t)n) != hipSuccess) return false;
            ms.cap = n;
        }
        if (!ms.d_ctl && (hipMalloc(&ms.d_ctl, sizeof(int) * 4) != hipSuccess || hipMemset(ms.d_ctl, 0, sizeof(int) * 4) != hipSuccess)) { (void)hipGetLastError(); return false; }
        const int slots = np / GP_NB / out_blocks + 1;      // one scratch tile per outer panel: the started sum of its first diagonal tile
        if (slots > ms.sums_cap) {
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
            if (ms.d_sums) (void)hipFree(ms.d_sums);
            ms.d_sums = nullptr, ms.sums_cap = 0;
            if (hipMalloc(&ms.d_sums, sizeof(double) * GP_NB * GP_NB * (size_t)slots) != hipSuccess) { (void)hipGetLastError(); return false; }
            ms.sums_cap = slots;
        }
        // (synchronous copy: the previous launch may still be reading the old list)
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
        if (hipMemcpy(ms.d_tasks, pl.tasks.data(), sizeof(mega_task) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return false;
        ms.plan = std::move(pl);
    }
    const int n = (int)ms.plan.tasks.size();
    if (++ms.epoch <= 0) { ms.epoch = 1; (void)hipMemsetAsync(ms.d_done, 0, sizeof(int) * (size_t)ms.cap, ctx->stream); }
    (void)hipMemsetAsync(ms.d_ctl, 0, sizeof(int), ctx->stream);
    // lab: GPCORE_MEGA_TRACE=<file> -- per-task time stamps (claimed / dependencies met / body done / published, 100 MHz ticks) and the
    // task list of THIS launch written to <file> (tools/mega_trace.py reads it); synchronises, so never set it for a measurement
    const char *trace = getenv("GPCORE_MEGA_TRACE");
    unsigned long long *d_st = nullptr;
    if (trace && hipMalloc(&d_st, sizeof(unsigned long long) * 4 * (size_t)n) != hipSuccess) { (void)hipGetLastError(); d_st = nullptr; }
    if (d_st) (void)hipMemsetAsync(d_st, 0, sizeof(unsigned long long) * 4 * (size_t)n, ctx->stream);
    gp_prof_begin(ctx, GP_PROF_SYRK);
    gpk_chol_mega(ctx->stream, workers, A, lda, dinv, ctx->d_info, ms.d_tasks, n, ms.d_done, ms.d_ctl, ms.epoch, ms.d_ctl + 1, ms.d_sums, d_st);
    gp_prof_end(ctx, GP_PROF_SYRK, (double)np * np * np / 3.0);
    if (d_st) {
        std::vector<unsigned long long> h(4 * (size_t)n);
        if (hipStreamSynchronize(ctx->stream) == hipSuccess && hipMemcpy(h.data(), d_st, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(trace, "wb")) {
                const int hdr[4] = {n, np, extra, out_blocks};
                fwrite(hdr, sizeof(int), 4, f);
                fwrite(ms.plan.tasks.data(), sizeof(mega_task), (size_t)n, f);
                fwrite(h.data(), sizeof(unsigned long long), h.size(), f);
                fclose(f);
            }
        }
        (void)hipFree(d_st);
    }
    return true;
}

void chol_blocked(gp_ctx *ctx, double *A, int np, int lda, double *dinv, int extra, int count = 1, size_t strideA = 0, size_t strideDinv = 0,
                  int *info = nullptr) {
    hipStream_t s = ctx->stream, s2 = ctx->side;
    const int rows = np + extra;
    if (!info) info = ctx->d_info;
    gp_batch bdiag, btrsm, bgemm;
    bdiag.count = btrsm.count = bgemm.count = count;
    bdiag.s0 = strideA, bdiag.s1 = strideDinv;
    btrsm.s0 = strideA, btrsm.s1 = strideA, btrsm.s2 = strideDinv;
    bgemm.s0 = bgemm.s1 = bgemm.s2 = strideA;
    // Look-ahead (far part of the outer update on the CU-masked side stream, under the next panel's chain): -1 = by size.
    // Measured with the side stream's reserved CUs spread over the XCDs (gp_ctx_create): n = 8192 fit 7.89 -> 7.40 ms, n = 4096
    // 2.79 -> 2.96 ms (the chain kernels slow down 2-4x while a full-chip GEMM runs beside them, so short factorisations lose),
    // n = 32768 198 -> 196..203 ms; the EP refactorisation (4096 rows of V riding along) 96.6 -> 99.5 sweeps/s.
    // (re-measured with the 25 us diagonal kernel: n = 4096 1.961 -> 1.947 ms, n = 5120 2.775 -> 2.683, n = 6144 3.757 -> 3.569, n = 16384 30.85 -> 30.32, n = 24576 87.98 -> 89.39: on for 4096 rows .. np = 16384)
    const bool lookahead = count == 1 && (ctx->lookahead > 0 || (ctx->lookahead < 0 && rows >= 4096 && np <= 16384));
    bool side_busy = false;
    // outer panel width: 512 (K = 512 trailing updates); 1024 measured 3 % faster at n = 32768 (205 -> 198.5 ms), equal at 8192
    const int outer_env = gp_env_blocks("GPCORE_OUTER");
    const int OUTER = outer_env ? outer_env : (np > 16384 ? 2 * GP_OUTER : GP_OUTER);
    {   // One persistent launch instead of ~250 (chol_mega_kernel): single factorisations that would run with look-ahead, where it wins --
        // refit ms, launch-per-step / single launch (profiles/r04_final_fit_mega.log): n = 4096 1.92 / 1.98, 5120 2.70 / 2.55, 6144 3.42 / 3.20,
        // 8192 5.74 / 4.81, 10240 9.17 / 8.17, 12288 14.27 / 13.45, 14336 21.30 / 20.55, 16384 30.05 / 30.03: below ~5000 rows the chain
        // is the same length either way, above ~14000 the factorisation is bound by GEMM throughput and one workgroup per CU on
        // 128 x 128 tiles (77 us per K = 512 tile) is no faster than two.  GPCORE_CHOL_MEGA = 0 / 1 forces (read per call: the tests
        // run every form in one process).
        const char *me = getenv("GPCORE_CHOL_MEGA");
        const bool forced = me && atoi(me) != 0, off = me && atoi(me) == 0;
        const bool by_size = np >= 5120 && np <= 14336;
        if (!off && (forced || by_size) && lookahead && info == ctx->d_info && np >= 4 * OUTER && extra <= GP_NB &&
            chol_mega_run(ctx, A, np, lda, dinv, extra, OUTER / GP_NB, forced)) return;
    }
    // (Round 4 tried the chain on the outer panel's OWN rows only, with the rows under the panel -- panel solve and their share of the
    // in-panel update -- on a second stream gated by one event per step: bit-identical, and slower at every size (n = 8192 refit 6.86
    // against 5.93 ms, profiles/r04_b_chol_split_fit.log).  Those launches are throughput work the step has to do anyway; moved off
    // the chain they queue behind the far update's workgroups like every other launch, and each step pays two more launches and a
    // cross-stream event.  Removed again; what the chip idles through is the 24.5 us of the single-workgroup diagonal block.)
    for (int K0 = 0; K0 < np; K0 += OUTER) {
        const int wcols = std::min(OUTER, np - K0);
        for (int k0 = K0; k0 < K0 + wcols; k0 += GP_NB) {
            double *Akk = A + (size_t)k0 + (size_t)k0 * lda;
            double *dk = dinv + (size_t)k0 * 16;
            gp_prof_begin(ctx, GP_PROF_POTRF_DIAG);
            gpk_potrf_diag128(s, Akk, lda, dk, info, k0, bdiag);
            gp_prof_end(ctx, GP_PROF_POTRF_DIAG, (double)count * GP_NB * GP_NB * GP_NB / 3.0);
            const int r = rows - (k0 + GP_NB);
            if (r <= 0) continue;
            double *A21 = Akk + GP_NB;
            gp_prof_begin(ctx, GP_PROF_TRSM);
            gpk_trsm_panel128(s, A21, r, lda, Akk, lda, dk, nullptr, nullptr, nullptr, btrsm);
            gp_prof_end(ctx, GP_PROF_TRSM, (double)count * r * GP_NB * GP_NB);
            const int wc = K0 + wcols - (k0 + GP_NB);
            if (wc > 0) {
                gp_prof_begin(ctx, GP_PROF_PANEL_UPD);
                // a single factorisation: quarter-size tiles (the launch is bound by the latency of one tile, not by throughput);
                // a lockstep batch fills the chip with the general 128 x 128 kernel
                if (count == 1 && small_panel_update()) gpk_gemm_k128_sub(s, r, wc, A21, lda, A21, lda, A21 + (size_t)GP_NB * lda, lda, 1);
                else gpk_gemm_nt(s, r, wc, GP_NB, -1.0, A21, lda, A21, lda, 1.0, A21 + (size_t)GP_NB * lda, lda, 1, 0, bgemm);
                gp_prof_end(ctx, GP_PROF_PANEL_UPD, count * trapezoid_flops(r, wc, GP_NB));
            }
        }
        const int c1 = K0 + wcols;
        const int R = np - c1;
        if (R <= 0) break;
        if (side_busy) { (void)hipStreamWaitEvent(s, ctx->ev_b, 0); side_busy = false; }
        const double *P = A + (size_t)c1 + (size_t)K0 * lda;
        const int nnext = lookahead ? std::min(OUTER, R) : R;
        if (R > nnext) {
            (void)hipEventRecord(ctx->ev_a, s);          // outer panel K0 is complete at this point
            (void)hipStreamWaitEvent(s2, ctx->ev_a, 0);
            const int c2 = c1 + nnext;
            const double *P2 = A + (size_t)c2 + (size_t)K0 * lda;
            gp_prof_begin(ctx, GP_PROF_SYRK, s2);
            // (the far update on the 64 x 64-tile kernel, whose workgroups retire four times as often: n = 8192 refit 6.32 against 5.94 ms,
            //  n = 12288 16.4 against 14.5; the main stream at the device's highest priority: 5.93 against 5.94 -- profiles/r04_c_fit_priority_far_small.log)
            gpk_gemm_nt(s2, rows - c2, np - c2, wcols, -1.0, P2, lda, P2, lda, 1.0, A + (size_t)c2 + (size_t)c2 * lda, lda, 1);
            gp_prof_end(ctx, GP_PROF_SYRK, trapezoid_flops(rows - c2, np - c2, wcols), s2);
            (void)hipEventRecord(ctx->ev_b, s2);
            side_busy = true;
        }
        gp_prof_begin(ctx, GP_PROF_SYRK);
        // the last outer updates of a single factorisation have fewer 128 x 128 tiles than the chip has slots: quarter-size tiles
        const double tiles128 = trapezoid_flops(rows - c1, nnext, wcols) / (2.0 * GP_NB * GP_NB * wcols);
        if (count == 1 && tiles128 < small_update_tiles() && small_panel_update())
            gpk_gemm_k128_sub(s, rows - c1, nnext, P, lda, P, lda, A + (size_t)c1 + (size_t)c1 * lda, lda, 1, wcols);
        else
            gpk_gemm_nt(s, rows - c1, nnext, wcols, -1.0, P, lda, P, lda, 1.0, A + (size_t)c1 + (size_t)c1 * lda, lda, 1, 0, bgemm);
        gp_prof_end(ctx, GP_PROF_SYRK, count * trapezoid_flops(rows - c1, nnext, wcols));
    }
    if (side_busy) (void)hipStreamWaitEvent(s, ctx->ev_b, 0);
}

// alpha <- L^-T z: block backward substitution, one fused launch per block step (z is consumed).
void back_solve_vec(gp_ctx *ctx, const double *L, int np, int ldl, const double *dinv, double *z, double *alpha) {
    hipStream_t s = ctx->stream;
    for (int k = np / GP_NB - 1; k >= 0; --k)
        gpk_bwd_step(s, L, ldl, dinv + (size_t)k * GP_NB * 16, z, alpha, k * GP_NB);
}

// Vt (mp x np, ld mp) <- Vt * L^-T, block column by block column (forwardSolve(L, K*^T) transposed).
// sumsq (length mp) accumulates row sums of squares of the result when non-null.
// Row count (in 128-row tiles) from which the left-looking form is used
int rows_left_min() {
    const char *e = getenv("GPCORE_ROWS_LEFT_MIN");   // read per call: the tests lower it to run the large-batch form at small sizes
    const int x = e ? atoi(e) : 0;
    return x > 0 ? x : 192;
}

// Lw (np x np, lower, ld np): block row i = L_ii^-1 [ -L_i,<i | I ].  With it the left-looking step for block column i,
//   Vt_i <- (Vt_i - Vt_<i L_i,<i^T) L_ii^-T,  is ONE product  Vt_i <- Vt[:, 0:(i+1)*128] Lw[i-block, 0:(i+1)*128]^T
// (the panel solve becomes 128 more k-steps of the GEMM that is running anyway, and its row reductions move into the GEMM
// epilogue).  Built from L by: negated block transpose into upper form -> one batched row-panel solve (block column i
// against L_ii, all np/128 of them in one launch) -> transpose back.  `scratch` is np x np.
void build_lw(gp_ctx *ctx, double *Lw, double *scratch, const double *L, int np, int ldl, const double *dinv) {
    hipStream_t s = ctx->stream;
    gpk_lw_transpose(s, scratch, np, L, ldl, np, 1);
    gp_batch bt;
    bt.count = np / GP_NB;
    bt.s0 = (size_t)GP_NB * np;             // next block column of the upper form
    bt.s1 = (size_t)GP_NB * (ldl + 1);      // next diagonal block of L
    bt.s2 = (size_t)GP_NB * 16;
    bt.tri = 1;
    gp_prof_begin(ctx, GP_PROF_TRSM);
    gpk_trsm_panel128(s, scratch, np, np, L, ldl, dinv, nullptr, nullptr, nullptr, bt);
    gp_prof_end(ctx, GP_PROF_TRSM, (double)np * np / 2.0 * GP_NB);
    gpk_lw_transpose(s, Lw, np, scratch, np, np, 0);
}

void solve_rows_lower(gp_ctx *ctx, double *Vt, int mp, const double *L, int np, int ldl, const double *dinv, double *sumsq,
                      const double *tvec = nullptr, double *dots = nullptr, const double *Lw = nullptr) {
    hipStream_t s = ctx->stream;
    const int nblk = np / GP_NB;
    if (Lw && sumsq && (mp / GP_NB) >= rows_left_min()) {
        for (int i = 0; i < nblk; ++i) {
            const int K = (i + 1) * GP_NB;
            gp_prof_begin(ctx, GP_PROF_GEMM);
            gpk_gemm_nt_rowred(s, mp, GP_NB, K, Vt, mp, Lw + (size_t)i * GP_NB, np, Vt + (size_t)i * GP_NB * mp, mp, sumsq,
                               tvec ? tvec + (size_t)i * GP_NB : nullptr, dots);
            gp_prof_end(ctx, GP_PROF_GEMM, 2.0 * mp * GP_NB * (double)K);
        }
        return;
    }
    // Few rows (mp/128 tiles per step would leave most of the 256 CUs idle): right-looking -- after block column i is
    // solved, ALL later block columns are updated by one wide GEMM (K = 128).  Many rows (the posterior batches):
    // left-looking -- each block column is hit once by a long-K GEMM, the most efficient shape for the MFMA kernel.
    // GPCORE_ROWS_LEFT_MIN (row tiles) moves the switch point; the tests use it to run both forms at small sizes.
    const bool right_looking = (mp / GP_NB) < rows_left_min();
    if (right_looking) {
        // two-level: inside an outer block of OB columns the K = 128 updates touch only that block's later columns; everything
        // to the right of the block is updated once per outer block with K = OB (fewer passes over the later columns of Vt)
        const int ob_env = gp_env_blocks("GPCORE_ROWS_OUTER");
        const int OB = ob_env ? ob_env : 2 * GP_OUTER;   // 1024: measured +1.4 % on the EP refactorisation (n = 4096) over 128, 512 in between
        for (int c0 = 0; c0 < np; c0 += OB) {
            const int c1 = std::min(np, c0 + OB);
            for (int k0 = c0; k0 < c1; k0 += GP_NB) {
                double *Vk = Vt + (size_t)k0 * mp;
                gp_prof_begin(ctx, GP_PROF_TRSM);
                gpk_trsm_panel128(s, Vk, mp, mp, L + (size_t)k0 + (size_t)k0 * ldl, ldl, dinv + (size_t)(k0 / GP_NB) * GP_NB * 16, sumsq,
                                  tvec ? tvec + k0 : nullptr, dots);
                gp_prof_end(ctx, GP_PROF_TRSM, (double)mp * GP_NB * GP_NB);
                const int rest = c1 - (k0 + GP_NB);
                if (rest > 0) {
                    gp_prof_begin(ctx, GP_PROF_GEMM);
                    gemm_sub_single(s, mp, rest, GP_NB, Vk, mp, L + (size_t)(k0 + GP_NB) + (size_t)k0 * ldl, ldl, Vt + (size_t)(k0 + GP_NB) * mp, mp);
                    gp_prof_end(ctx, GP_PROF_GEMM, 2.0 * mp * (double)rest * GP_NB);
                }
            }
            const int right = np - c1;
            if (right > 0) {
                gp_prof_begin(ctx, GP_PROF_GEMM);
                gemm_sub_single(s, mp, right, c1 - c0, Vt + (size_t)c0 * mp, mp, L + (size_t)c1 + (size_t)c0 * ldl, ldl, Vt + (size_t)c1 * mp, mp);
                gp_prof_end(ctx, GP_PROF_GEMM, 2.0 * mp * (double)right * (c1 - c0));
            }
        }
        return;
    }
    for (int i = 0; i < nblk; ++i) {
        double *Vi = Vt + (size_t)i * GP_NB * mp;
        if (i > 0) {
            gp_prof_begin(ctx, GP_PROF_GEMM);
            gpk_gemm_nt(s, mp, GP_NB, i * GP_NB, -1.0, Vt, mp, L + (size_t)i * GP_NB, ldl, 1.0, Vi, mp, 0);
            gp_prof_end(ctx, GP_PROF_GEMM, 2.0 * mp * GP_NB * (double)i * GP_NB);
        }
        gp_prof_begin(ctx, GP_PROF_TRSM);
        gpk_trsm_panel128(s, Vi, mp, mp, L + (size_t)i * GP_NB + (size_t)i * GP_NB * ldl, ldl, dinv + (size_t)i * GP_NB * 16, sumsq,
                          tvec ? tvec + (size_t)i * GP_NB : nullptr, dots);
        gp_prof_end(ctx, GP_PROF_TRSM, (double)mp * GP_NB * GP_NB);
    }
}

// T (np x np) <- L^-T, i.e. the identity solved against L in row form, RIGHT-looking: after block column i of T is
// final (rows 0 .. (i+1)*128 only -- T is upper triangular), every later block column is updated at once:
//   T[0:(i+1)*128, j] -= T_i * L[j-block, i-block]^T   for all j > i    (M = (i+1)*128, N = np-(i+1)*128, K = 128).
// Same n^3/3 flops as a structure-exploiting left-looking solve, but each step is one wide GEMM (up to n^2/4/128^2
// tiles) instead of an (np/128)-tile one -- at n = 4096 the left-looking form keeps 32 of 256 CUs busy.
// For a single problem a two-level form (K = 128 updates kept inside a 512-wide outer block, one K = 512 update to the
// right of it per outer block) measured 2-4 % slower at n = 4096 (the long-K GEMM multiplies the zero lower part of the
// outer block and the K = 128 updates lose their width); with a lockstep batch the launches are wide enough either way and
// the K = 128 form is bound by re-reading and re-writing the later columns of T, so batches use the two-level form.
// lower_is_zero: the caller guarantees zeros below T's diagonal blocks (they are never written here), so only the blocks on and above
// the diagonal are reset
void inverse_transpose_lower(gp_ctx *ctx, double *T, const double *L, int np, int ldl, const double *dinv, int count = 1, size_t strideT = 0,
                             size_t strideL = 0, size_t strideDinv = 0, bool lower_is_zero = false) {
    hipStream_t s = ctx->stream;
    gp_batch btrsm, bgemm;
    btrsm.count = bgemm.count = count;
    btrsm.s0 = strideT, btrsm.s1 = strideL, btrsm.s2 = strideDinv;
    bgemm.s0 = strideT, bgemm.s1 = strideL, bgemm.s2 = strideT;
    // Outer block width OB: inside an outer block column the K = 128 updates touch only that block's own columns; everything
    // to the right of it is updated once per outer block with K = OB.  OB = 128 is the plain right-looking form.
    const int ob_env = gp_env_blocks("GPCORE_TINV_OUTER");
    const int OB = ob_env ? ob_env : (count >= 4 ? GP_OUTER : GP_NB);
    for (int g = 0; g < count; ++g) {
        if (lower_is_zero) gpk_set_identity_upper(s, T + g * strideT, np, np);
        else gpk_set_identity(s, T + g * strideT, np, np);
    }
    for (int c0 = 0; c0 < np; c0 += OB) {
        const int c1 = std::min(np, c0 + OB);
        for (int k0 = c0; k0 < c1; k0 += GP_NB) {
            const int rows = k0 + GP_NB;   // non-zero rows of this block column
            double *Tk = T + (size_t)k0 * np;
            gp_prof_begin(ctx, GP_PROF_TRSM);
            gpk_trsm_panel128(s, Tk, rows, np, L + (size_t)k0 + (size_t)k0 * ldl, ldl, dinv + (size_t)(k0 / GP_NB) * GP_NB * 16, nullptr, nullptr, nullptr, btrsm);
            gp_prof_end(ctx, GP_PROF_TRSM, (double)count * rows * GP_NB * GP_NB);
            const int rest = c1 - rows;
            if (rest > 0) {
                gp_prof_begin(ctx, GP_PROF_GEMM);
                if (count == 1) gemm_sub_single(s, rows, rest, GP_NB, Tk, np, L + (size_t)rows + (size_t)k0 * ldl, ldl, T + (size_t)rows * np, np);
                else gpk_gemm_nt(s, rows, rest, GP_NB, -1.0, Tk, np, L + (size_t)rows + (size_t)k0 * ldl, ldl, 1.0, T + (size_t)rows * np, np, 0, 0, bgemm);
                gp_prof_end(ctx, GP_PROF_GEMM, 2.0 * count * rows * (double)rest * GP_NB);
            }
        }
        const int right = np - c1;
        if (right > 0) {   // T[0:c1, c1:] -= T[0:c1, c0:c1] * L[c1:, c0:c1]^T
            gp_prof_begin(ctx, GP_PROF_GEMM);
            gpk_gemm_nt(s, c1, right, c1 - c0, -1.0, T + (size_t)c0 * np, np, L + (size_t)c1 + (size_t)c0 * ldl, ldl, 1.0, T + (size_t)c1 * np, np, 0, 0, bgemm);
            gp_prof_end(ctx, GP_PROF_GEMM, 2.0 * count * c1 * (double)right * (c1 - c0));
        }
    }
}

// Vt (mp x np) <- Vt * U^-T with U (np x np) UPPER triangular: backward over block columns.
void solve_rows_upper(gp_ctx *ctx, double *Vt, int mp, const double *U, int np, int ldu) {
    hipStream_t s = ctx->stream;
    const int nblk = np / GP_NB;
    for (int i = nblk - 1; i >= 0; --i) {
        double *Vi = Vt + (size_t)i * GP_NB * mp;
        const int kr = (nblk - 1 - i) * GP_NB;
        if (kr > 0) {
            gp_prof_begin(ctx, GP_PROF_GEMM);
            gpk_gemm_nt(s, mp, GP_NB, kr, -1.0, Vt + (size_t)(i + 1) * GP_NB * mp, mp,
                        U + (size_t)i * GP_NB + (size_t)(i + 1) * GP_NB * ldu, ldu, 1.0, Vi, mp, 0);
            gp_prof_end(ctx, GP_PROF_GEMM, 2.0 * mp * GP_NB * (double)kr);
        }
        gpk_trsm_panel_upper(s, Vi, mp, mp, U + (size_t)i * GP_NB + (size_t)i * GP_NB * ldu, ldu);
    }
}

gp_status model_alloc(gp_ctx *ctx, int n, int d, bool has_x, gp_model **out) {
    gp_model *m = new (std::nothrow) gp_model();
    if (!m) return GP_ENOMEM;
    m->ctx = ctx; m->n = n; m->d = d; m->np = gp_pad(n); m->ldl = m->np + GP_NB; m->has_x = has_x;
    const size_t np = m->np;
    hipError_t e = hipSuccess;
    if (has_x) e = hipMalloc(&m->dX, sizeof(double) * (size_t)n * d);
    if (has_x && e == hipSuccess) e = hipMalloc(&m->dcen, sizeof(double) * 64);
    if (e == hipSuccess) e = hipMalloc(&m->dy, sizeof(double) * np);
    if (e == hipSuccess) e = hipMalloc(&m->dL, sizeof(double) * (np + GP_NB) * np);
    if (e == hipSuccess) e = hipMalloc(&m->dalpha, sizeof(double) * np);
    if (e == hipSuccess) e = hipMalloc(&m->dlml, sizeof(double) * 8);
    if (e == hipSuccess) e = hipMalloc(&m->ddinv, sizeof(double) * np * 16);
    if (e == hipSuccess) e = hipMalloc(&m->dtmp, sizeof(double) * 2 * np);
    if (e == hipSuccess) e = hipMemsetAsync(m->dL, 0, sizeof(double) * (np + GP_NB) * np, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(m->dy, 0, sizeof(double) * np, ctx->stream);
    if (e != hipSuccess) {
        GP_SET_ERR(ctx, "model allocation (n=%d) failed: %s", n, hipGetErrorString(e));
        gp_model_destroy(m);
        return GP_ENOMEM;
    }
    *out = m;
    return GP_OK;
}

// factor the matrix already sitting in model->dL (lower), then alpha and LML.
// y^T rides through the factorisation in row np of dL (forward solve for free), the backward solve follows.
void model_factor(gp_model *m) {
    gp_ctx *ctx = m->ctx;
    hipStream_t s = ctx->stream;
    (void)hipMemsetAsync(ctx->d_info, 0, sizeof(int), s);
    gpk_pad_identity(s, m->dL, m->n, m->np, m->ldl);
    gpk_copy_strided(s, m->dL + m->np, (size_t)m->ldl, m->dy, 1, m->np);
    chol_blocked(ctx, m->dL, m->np, m->ldl, m->ddinv, GP_NB);
    gpk_copy_strided(s, m->dtmp, 1, m->dL + m->np, (size_t)m->ldl, m->np);     // t = L^-1 y
    m->alpha_valid = false;
    m->lw_valid = false;
    gpk_lml(s, m->dL, m->n, m->ldl, m->dtmp, m->dlml);
}

// alpha = L^-T t, on demand: the posterior (mean = V^T t) and the LML (y.alpha = |t|^2) never need it
void ensure_alpha(gp_model *m) {
    if (m->alpha_valid) return;
    hipStream_t s = m->ctx->stream;
    double *z = m->dtmp + m->np;
    (void)hipMemcpyAsync(z, m->dtmp, sizeof(double) * m->np, hipMemcpyDeviceToDevice, s);
    back_solve_vec(m->ctx, m->dL, m->np, m->ldl, m->ddinv, z, m->dalpha);
    m->alpha_valid = true;
}

}  // namespace

// external-linkage shims for the other host translation units (gpcore_ep.hip)
gp_status gpi_ws_get(gp_ctx *ctx, int slot, size_t bytes, double **out) { return ws_get(ctx, slot, bytes, out); }
gp_status gpi_upload_2d(gp_ctx *ctx, double *dst, int ldd, const double *src, int lds, int rows, int cols) { return upload_2d(ctx, dst, ldd, src, lds, rows, cols); }
gp_status gpi_download_2d(gp_ctx *ctx, double *dst, int ldd, const double *src, int lds, int rows, int cols) { return download_2d(ctx, dst, ldd, src, lds, rows, cols); }
gp_status gpi_read_info(gp_ctx *ctx, int *info) { return read_info(ctx, info); }
void gpi_chol_blocked(gp_ctx *ctx, double *A, int np, int lda, double *dinv, int extra) { chol_blocked(ctx, A, np, lda, dinv, extra); }

// Step k0 of chol_blocked (single problem, GP_OUTER-wide outer panels, no look-ahead) on stream s.  Calling it for
// k0 = 0, 128, ..., np - 128 in order IS the factorisation; the caller may change columns >= k0 between steps (the EP sweep
// scales block k0's rows and columns when its site precisions become final).  `solved` (optional) is recorded when block column k0 of the factor (and of the rows riding along) is final.
// With a `far` stream (and `solved`, `far_done`) the step that closes an outer panel only updates the NEXT outer panel's columns on s
// and leaves everything to the right of it to `far`, which starts at `solved` and records `far_done`; the closing step of the next
// outer panel waits for that event before it touches those columns (the look-ahead of chol_blocked, one step at a time).
void gpi_chol_panel_step(gp_ctx *ctx, hipStream_t s, double *A, int np, int lda, double *dinv, int extra, int k0, hipEvent_t solved,
                         hipStream_t far, hipEvent_t far_done, int count, size_t strideA, size_t strideDinv, int *info) {
    const int rows = np + extra;
    const int K0 = k0 / GP_OUTER * GP_OUTER, wcols = std::min(GP_OUTER, np - K0);
    double *Akk = A + (size_t)k0 + (size_t)k0 * lda;
    double *dk = dinv + (size_t)k0 * 16;
    gp_batch bdiag, btrsm, bgemm;
    bdiag.count = btrsm.count = bgemm.count = count;
    bdiag.s0 = strideA, bdiag.s1 = strideDinv;
    btrsm.s0 = strideA, btrsm.s1 = strideA, btrsm.s2 = strideDinv;
    bgemm.s0 = bgemm.s1 = bgemm.s2 = strideA;
    const bool single = count == 1;      // the quarter-tile kernel serves single problems; a batch fills the chip with 128 x 128 tiles
    gpk_potrf_diag128(s, Akk, lda, dk, info ? info : ctx->d_info, k0, bdiag);
    const int r = rows - (k0 + GP_NB);
    double *A21 = Akk + GP_NB;
    if (r > 0) gpk_trsm_panel128(s, A21, r, lda, Akk, lda, dk, nullptr, nullptr, nullptr, btrsm);
    if (solved) (void)hipEventRecord(solved, s);
    if (r <= 0) return;
    const int c1 = K0 + wcols, wc = c1 - (k0 + GP_NB);
    if (wc > 0) {
        gp_prof_begin(ctx, GP_PROF_PANEL_UPD, s);
        if (single && small_panel_update()) gpk_gemm_k128_sub(s, r, wc, A21, lda, A21, lda, A21 + (size_t)GP_NB * lda, lda, 1);
        else gpk_gemm_nt(s, r, wc, GP_NB, -1.0, A21, lda, A21, lda, 1.0, A21 + (size_t)GP_NB * lda, lda, 1, 0, bgemm);
        gp_prof_end(ctx, GP_PROF_PANEL_UPD, count * trapezoid_flops(r, wc, GP_NB), s);
        return;
    }
    const int R = np - c1;
    if (R <= 0) return;
    const bool split = far && solved && far_done;
    const int near = split ? std::min(GP_OUTER, R) : R;
    if (split && K0 > 0) (void)hipStreamWaitEvent(s, far_done, 0);   // the previous outer panel's far update wrote the columns updated below
    if (R > near) {
        (void)hipStreamWaitEvent(far, solved, 0);
        const int c2 = c1 + near;
        const double *P2 = A + (size_t)c2 + (size_t)K0 * lda;
        const double tiles = trapezoid_flops(rows - c2, np - c2, wcols) / (2.0 * GP_NB * GP_NB * wcols);
        gp_prof_begin(ctx, GP_PROF_SYRK, far);
        if (single && tiles < small_update_tiles() && small_panel_update())
            gpk_gemm_k128_sub(far, rows - c2, np - c2, P2, lda, P2, lda, A + (size_t)c2 + (size_t)c2 * lda, lda, 1, wcols);
        else
            gpk_gemm_nt(far, rows - c2, np - c2, wcols, -1.0, P2, lda, P2, lda, 1.0, A + (size_t)c2 + (size_t)c2 * lda, lda, 1, 0, bgemm);
        gp_prof_end(ctx, GP_PROF_SYRK, count * trapezoid_flops(rows - c2, np - c2, wcols), far);
    }
    if (split) (void)hipEventRecord(far_done, far);
    const double *P = A + (size_t)c1 + (size_t)K0 * lda;
    const double tiles128 = trapezoid_flops(rows - c1, near, wcols) / (2.0 * GP_NB * GP_NB * wcols);
    gp_prof_begin(ctx, GP_PROF_SYRK, s);
    if (single && tiles128 < small_update_tiles() && small_panel_update())
        gpk_gemm_k128_sub(s, rows - c1, near, P, lda, P, lda, A + (size_t)c1 + (size_t)c1 * lda, lda, 1, wcols);
    else
        gpk_gemm_nt(s, rows - c1, near, wcols, -1.0, P, lda, P, lda, 1.0, A + (size_t)c1 + (size_t)c1 * lda, lda, 1, 0, bgemm);
    gp_prof_end(ctx, GP_PROF_SYRK, count * trapezoid_flops(rows - c1, near, wcols), s);
}
void gpi_solve_rows_lower(gp_ctx *ctx, double *Vt, int mp, const double *L, int np, int ldl, const double *dinv, double *sumsq,
                          const double *tvec, double *dots) { solve_rows_lower(ctx, Vt, mp, L, np, ldl, dinv, sumsq, tvec, dots); }
void gpi_inverse_transpose_lower(gp_ctx *ctx, double *T, const double *L, int np, int ldl, const double *dinv) { inverse_transpose_lower(ctx, T, L, np, ldl, dinv); }
void gpi_back_solve_vec(gp_ctx *ctx, const double *L, int np, int ldl, const double *dinv, double *z, double *alpha) { back_solve_vec(ctx, L, np, ldl, dinv, z, alpha); }
gp_status gpi_model_alloc(gp_ctx *ctx, int n, int d, bool has_x, gp_model **out) { return model_alloc(ctx, n, d, has_x, out); }
// alpha of a fitted model (computed on demand) copied into a caller's device buffer of n doubles
gp_status gpi_model_alpha(gp_model *m, double *dst) {
    ensure_alpha(m);
    GP_HIP(m->ctx, hipMemcpyAsync(dst, m->dalpha, sizeof(double) * m->n, hipMemcpyDeviceToDevice, m->ctx->stream));
    return GP_OK;
}
// k-th helper context of `ctx` (same device, own stream and workspaces), created on first use and destroyed with it
// The two extra streams of the EP sweep's streamed refactorisation, created on first use: every stream a context owns competes for
// the process's few hardware queues, and contexts that never run such a sweep (the workers of the batched LML path, small EP
// problems) are better off without them.
gp_status gpi_ctx_ep_streams(gp_ctx *ctx) {
    if (ctx->side2 && ctx->side3) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipSuccess, em;
    int reserved = 32;
    if (const char *rc = getenv("GPCORE_RESERVED_CUS")) reserved = atoi(rc);
    // Streams of the EP refactorisation that runs under the site loop, beside the site loop's own side work (gp_ep_sweep).
    // side3 carries its long GEMMs (far trailing updates, next covariance); GPCORE_RESERVED_CUS_EP (default 96 = 12 per XCD)
    // CUs are kept free of them, or the site loop's side stream -- whose work the serial chain waits for one block later --
    // is starved whenever they run (n = 4096 sweeps/s with 32 / 64 / 96 / 128 reserved: 161.3 / 171.3 / 177.2 / 163.9).
    // (round 3, with the fused chain kernel and the urgent tiles the chain needs less shelter: 40 / 48 / 56 / 64 / 72 / 80 / 96 reserved:
    // 196.6 / 196.4 / 200.7 / 201.8 / 192.4 / 189.6 / 188 sweeps/s at n = 4096, n = 8192 32.4 at 64 against 29.1 at 96 -> 64)
    int reserved_ep = reserved > 0 ? 64 : 0;
    if (const char *rc = getenv("GPCORE_RESERVED_CUS_EP")) reserved_ep = atoi(rc);
    // side2 carries the factorisation chain: its workgroups are short-lived (quarter-size tiles) but its single-workgroup
    // diagonal kernel needs a whole CU's LDS, which it only finds quickly on the CUs the masked streams leave alone -- so side2
    // itself is NOT masked.  (Keeping k CUs free of side2 as well was measured with the 25 us diagonal kernel, k = 0 / 8 / 16 / 24 / 32:
    // 163.5 / 162.7 / 163.2 / 162.3 / 164.3 sweeps/s at n = 4096 -- no effect, so the slow-down of the site loop's small kernels
    // while the other streams' GEMMs run is not a matter of finding a free CU; the switch is gone.)
    const int chain_reserved = 0;
    for (hipStream_t *sp : {&ctx->side2, &ctx->side3}) {
        if (e != hipSuccess) break;
        em = hipErrorInvalidValue;
        const int rsv = (sp == &ctx->side2) ? chain_reserved : reserved_ep;
        if (rsv > 0 && ctx->num_cu >= 64) {
            const int words = (ctx->num_cu + 31) / 32;
            std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
            if (ctx->num_cu % 32) mask[words - 1] = (1u << (ctx->num_cu % 32)) - 1u;
            for (int i = 0; i < rsv && i < ctx->num_cu; ++i) mask[i / 32] &= ~(1u << (i % 32));
            em = hipExtStreamCreateWithCUMask(sp, (uint32_t)words, mask.data());
        }
        if (em != hipSuccess) { (void)hipGetLastError(); e = hipStreamCreateWithFlags(sp, hipStreamNonBlocking); }
    }
    if (e != hipSuccess) { GP_SET_ERR(ctx, "EP streams: %s", hipGetErrorString(e)); return GP_EHIP; }
    return GP_OK;
}

static thread_local bool g_child_ctx = false;   // gp_ctx_create called for a helper context: no EP streams up front (gpi_ctx_ep_streams makes them if ever needed)

gp_ctx *gpi_child_ctx(gp_ctx *ctx, int k) {
    ctx_ext *x = ext_of(ctx);
    while ((int)x->children.size() <= k) {
        gp_ctx *c = nullptr;
        g_child_ctx = true;
        const gp_status cst = gp_ctx_create(ctx->device, nullptr, &c);
        g_child_ctx = false;
        if (cst != GP_OK) return nullptr;
        x->children.push_back(c);
    }
    return x->children[k];
}
// z <- L^-1 t (t is consumed): one fused launch per block step
void gpi_forward_solve_vec(gp_ctx *ctx, const double *L, int np, int ldl, const double *dinv, double *t, double *z) {
    for (int k = 0; k < np / GP_NB; ++k)
        gpk_fwd_step(ctx->stream, L, ldl, dinv + (size_t)k * GP_NB * 16, t, z, k * GP_NB, np - (k + 1) * GP_NB);
}

extern "C" {

const char *gp_version(void) { return "gpcore 0.1 (gfx950, fp64 MFMA)"; }

gp_status gp_ctx_create(int device, void *stream, gp_ctx **out) {
    if (!out) return GP_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return GP_EHIP;
    gp_ctx_full *ctx = new (std::nothrow) gp_ctx_full();
    if (!ctx) return GP_ENOMEM;
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) {
        ctx->num_cu = prop.multiProcessorCount;
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete ctx; return GP_EHIP; }  // gfx950 code objects only
    }
    if (e == hipSuccess) {
        if (stream) { ctx->stream = static_cast<hipStream_t>(stream); ctx->own_stream = false; }
        else { e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking); ctx->own_stream = true; }
    }
    if (e == hipSuccess) {
        // Side stream for the far trailing update of the Cholesky.  It is CU-masked to leave a few CUs free of its
        // long-running GEMM workgroups: the single-workgroup diagonal-block kernel on the main stream needs ~150 KB
        // of LDS, i.e. a whole CU, and otherwise waits hundreds of microseconds for two GEMM workgroups on one CU
        // to retire together.  GPCORE_RESERVED_CUS (default 32 = 4 per XCD, 0 = no mask) sets how many CUs stay reserved.
        int reserved = 32;
        if (const char *rc = getenv("GPCORE_RESERVED_CUS")) reserved = atoi(rc);
        hipError_t em = hipErrorInvalidValue;
        if (reserved > 0 && ctx->num_cu >= 64) {
            const int words = (ctx->num_cu + 31) / 32;
            std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
            if (ctx->num_cu % 32) mask[words - 1] = (1u << (ctx->num_cu % 32)) - 1u;
            // Bit b of the mask is CU (b / 8) of XCD (b % 8): the driver deals the flat mask round-robin over the 8 XCDs, and the
            // dispatcher hands every XCD the same share of a grid whatever its CU count.  So the reserved CUs must be spread
            // EVENLY over the XCDs -- bits 0 .. reserved-1 -- or the XCD that lost more CUs finishes last (measured: 8 reserved CUs
            // all in XCD 0, bits 0, 32, 64, ..., made a 406-tile GEMM take 1.56 ms instead of 1.08 ms).
            for (int i = 0; i < reserved && i < ctx->num_cu; ++i) mask[i / 32] &= ~(1u << (i % 32));
            em = hipExtStreamCreateWithCUMask(&ctx->side, (uint32_t)words, mask.data());
        }
        if (em != hipSuccess) { (void)hipGetLastError(); e = hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking); }
    }
    // A caller's context gets the EP sweep's two extra streams right away (created together with the first two they land on
    // hardware queues of their own: made later, on first use, a sweep at n = 4096 ran at 104 instead of 180 sweeps/s); helper
    // contexts (batched LML workers, concurrent small EP runs) do without: every stream competes for the few hardware queues.
    if (e == hipSuccess && !g_child_ctx && gpi_ctx_ep_streams(ctx) != GP_OK) e = hipErrorInvalidValue;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming);
    if (const char *la = getenv("GPCORE_LOOKAHEAD")) ctx->lookahead = atoi(la) != 0 ? 1 : 0;
    if (e == hipSuccess) e = hipMalloc(&ctx->d_scalars, sizeof(double) * 256);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_info, sizeof(int) * 8 + sizeof(double) * 64);      // 8 status ints, then gp_gram_center's 64 doubles
    if (e == hipSuccess) e = hipMemset(ctx->d_info, 0, sizeof(int) * 8 + sizeof(double) * 64);
    if (e == hipSuccess && (gpk_init_diag_kernels() != 0 || gpk_init_gemm_kernels() != 0)) e = hipErrorInvalidValue;
    if (e != hipSuccess) { delete ctx; return GP_EHIP; }
    *out = ctx;
    return GP_OK;
}

void gp_ctx_destroy(gp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int c = 0; c < GP_PROF_NCLASSES; ++c)
        for (hipEvent_t e : ctx->prof[c].ev) (void)hipEventDestroy(e);
    ctx_ext *x = ext_of(ctx);
    for (gp_ctx *c : x->children) gp_ctx_destroy(c);
    x->children.clear();
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < WS_COUNT; ++i) if (x->ws[i].p) (void)hipFree(x->ws[i].p);
    if (x->mega.d_tasks) (void)hipFree(x->mega.d_tasks);
    if (x->mega.d_done) (void)hipFree(x->mega.d_done);
    if (x->mega.d_ctl) (void)hipFree(x->mega.d_ctl);
    if (x->mega.d_sums) (void)hipFree(x->mega.d_sums);
    if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
    if (ctx->d_info) (void)hipFree(ctx->d_info);
    if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->side2) (void)hipStreamDestroy(ctx->side2);
    if (ctx->side3) (void)hipStreamDestroy(ctx->side3);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete static_cast<gp_ctx_full *>(ctx);
}

gp_status gp_ctx_trim(gp_ctx *ctx) {
    if (!ctx) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx_ext *x = ext_of(ctx);
    for (gp_ctx *c : x->children) GP_TRY(gp_ctx_trim(c));
    for (int i = 0; i < WS_COUNT; ++i)
        if (x->ws[i].p) { (void)hipFree(x->ws[i].p); x->ws[i].p = nullptr; x->ws[i].bytes = 0; }
    return GP_OK;
}

gp_status gp_ctx_sync(gp_ctx *ctx) {
    if (!ctx) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GP_OK;
}

const char *gp_last_error(const gp_ctx *ctx) { return ctx ? ctx->err : "null context"; }

gp_status gp_ctx_profile(gp_ctx *ctx, int mask) {
    if (!ctx || mask < 0 || mask >= (1 << GP_PROF_NCLASSES)) return GP_EINVAL;
    ctx->prof_which = mask;
    return GP_OK;
}

gp_status gp_ctx_set_lookahead(gp_ctx *ctx, int mode) {
    if (!ctx || mode < -1 || mode > 1) return GP_EINVAL;
    ctx->lookahead = mode;
    return GP_OK;
}

gp_status gp_ctx_profile_read(gp_ctx *ctx, int which, int64_t *launches, double *total_ms, double *work) {
    if (!ctx || which <= 0 || which >= GP_PROF_NCLASSES) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    gp_prof_slot &p = ctx->prof[which];
    double ms = 0.0;
    for (size_t i = 0; i + 1 < p.ev.size(); i += 2) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, p.ev[i], p.ev[i + 1]) == hipSuccess) ms += t;
    }
    for (hipEvent_t e : p.ev) (void)hipEventDestroy(e);
    if (launches) *launches = p.launches;
    if (total_ms) *total_ms = ms;
    if (work) *work = p.work;
    p.ev.clear(); p.launches = 0; p.work = 0.0;
    return GP_OK;
}

gp_status gp_probe_mfma_f64(gp_ctx *ctx, double *tflops) {
    if (!ctx || !tflops) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double v = gpk_probe_mfma(ctx->stream, ctx->num_cu, 2, nullptr, nullptr);
    if (v < 0) { GP_SET_ERR(ctx, "probe allocation failed"); return GP_ENOMEM; }
    *tflops = v;
    return GP_OK;
}

gp_status gp_chol_plan_info(int n, int extra_rows, int workgroups, int *ntasks, double *model_us) {
    if (n < GP_NB || n % GP_NB || extra_rows < 0 || extra_rows > GP_NB || extra_rows % GP_NB || workgroups < 1 || !ntasks) return GP_EINVAL;
    mega_plan pl;
    const int ob = GP_OUTER / GP_NB;
    if (!chol_mega_plan(n, extra_rows, ob, workgroups, pl)) return GP_EINVAL;
    // independent check of the list: replay it and count, per tile, the updates in the order the two-level scheme applies them
    const int nb = n / GP_NB, nrow = (n + extra_rows) / GP_NB;
    std::vector<int> upd((size_t)nrow * nb, 0), quarters((size_t)nrow * nb, 0), solved((size_t)nrow * nb, 0), fact(nb, 0), started(nb, 0);
    auto expected = [&](int i, int j) { const int J = j / ob; return J + (j - J * ob); };      // outer updates of the panels before j's, then the in-panel ones
    for (size_t p2 = 0; p2 < pl.tasks.size(); ++p2) {
        const mega_task &t = pl.tasks[p2];
        for (int d = 0; d < 10; ++d) if (t.dep[d] >= (int)p2) return GP_EINVAL;
        if (t.type == 3) continue;                       // a place holder: announced by the fused task that names it
        if (t.type == 0) { if (upd[(size_t)t.k0 * nb + t.k0] != expected(t.k0, t.k0) || fact[t.k0]++) return GP_EINVAL; }
        else if (t.type == 5) {
            // a quarter of the started sum of a panel's first diagonal tile: rows t.i of the previous panel's block columns but the last,
            // all solved; three quarters per slot, each once
            const int k = t.i;
            if (k % ob != 0 || k == 0 || t.j != k / ob || t.k0 != k - ob || t.kb != ob - 1 || t.q < 0 || t.q > 3 || t.q == 2 || fact[k]) return GP_EINVAL;
            for (int kk = t.k0; kk < t.k0 + t.kb; ++kk) if (!solved[(size_t)k * nb + kk]) return GP_EINVAL;
            if (started[k] & (1 << t.q)) return GP_EINVAL;
            started[k] |= 1 << t.q;
        }
        else if (t.type == 4) {
            // link + diagonal block: solves rows k of block column k-1 (after all ITS updates) and completes the update of tile (k, k) those
            // rows belong to: the in-panel one, or -- first block of a panel -- the previous panel's outer one, continued from its slot
            const int k = t.k0;
            if (k == 0 || !fact[k - 1] || upd[(size_t)k * nb + (k - 1)] != expected(k, k - 1) || solved[(size_t)k * nb + (k - 1)]++) return GP_EINVAL;
            if (k % ob == 0 && ob > 1 ? (t.j != k / ob || started[k] != 0xB) : (t.j != -1)) return GP_EINVAL;
            if (t.q < 0 || t.q >= (int)pl.tasks.size() || pl.tasks[t.q].type != 3 || (size_t)t.q <= p2) return GP_EINVAL;   // its place holder follows it in the list
            if (++upd[(size_t)k * nb + k] != expected(k, k) || fact[k]++) return GP_EINVAL;
        }
        else if (t.type == 1) { if (!fact[t.k0] || upd[(size_t)t.i * nb + t.k0] != expected(t.i, t.k0) || solved[(size_t)t.i * nb + t.k0]++) return GP_EINVAL; }
        else {
            for (int kk = t.k0; kk < t.k0 + t.kb; ++kk) if (!solved[(size_t)t.i * nb + kk] || (t.j < nb && t.i != t.j && !solved[(size_t)t.j * nb + kk])) return GP_EINVAL;
            int &qd = quarters[(size_t)t.i * nb + t.j];
            const int need = t.q < 0 ? 1 : (t.i == t.j ? 3 : 4);
            if (++qd == need) { qd = 0; ++upd[(size_t)t.i * nb + t.j]; }
        }
    }
    for (int k = 0; k < nb; ++k) if (!fact[k]) return GP_EINVAL;
    for (int j = 0; j < nb; ++j) for (int i = j + 1; i < nrow; ++i) if (!solved[(size_t)i * nb + j]) return GP_EINVAL;
    *ntasks = (int)pl.tasks.size();
    if (model_us) *model_us = pl.model_us;
    return GP_OK;
}

gp_status gp_probe_mfma_f64_ex(gp_ctx *ctx, int waves_per_simd, double *tflops, double *clock_mhz, double *cycles_per_mfma) {
    if (!ctx || !tflops || waves_per_simd < 1 || waves_per_simd > 2) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double v = gpk_probe_mfma(ctx->stream, ctx->num_cu, waves_per_simd, clock_mhz, cycles_per_mfma);
    if (v < 0) { GP_SET_ERR(ctx, "probe allocation failed"); return GP_ENOMEM; }
    *tflops = v;
    return GP_OK;
}

gp_status gp_dev_alloc(gp_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e != hipSuccess) { GP_SET_ERR(ctx, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return GP_ENOMEM; }
    return GP_OK;
}
gp_status gp_dev_free(gp_ctx *ctx, void *dptr) {
    if (!ctx) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (dptr) GP_HIP(ctx, hipFree(dptr));
    return GP_OK;
}
gp_status gp_dev_upload(gp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!ctx) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_HIP(ctx, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GP_OK;
}
gp_status gp_dev_download(gp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes) {
    if (!ctx) return GP_EINVAL;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    GP_HIP(ctx, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GP_OK;
}

gp_status gp_hp_get_at_position(const double *theta, int d, int pos, double *out) {
    if (!theta || !out || d < 0) return GP_EINVAL;
    if (pos < 1 || pos > d + 2) return GP_ERANGE;
    *out = theta[pos - 1];  // [sf, l_1..l_d, sn], 1-based
    return GP_OK;
}

// ---- Gram ---------------------------------------------------------------------------------------
gp_status gp_gram_rbf_dev(gp_ctx *ctx, const double *dX, int n, int d, int ldx, const double *theta, double *dK, int ldk, int uplo) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, dX && theta && dK, "null pointer");
    GP_REQUIRE(ctx, n >= 0 && d >= 1 && d <= 64 && ldx >= n && ldk >= n, "bad dimensions (1 <= d <= 64)");
    if (n == 0) return GP_OK;
    gp_prof_begin(ctx, GP_PROF_GRAM);
    // (one-shot call: the first point as centre, no extra launch; a fitted model and the batched paths use the centroid, computed once)
    gpk_gram_sym(ctx->stream, dX, n, d, ldx, theta, dK, ldk, uplo == GP_FULL, 0.0, gp_gram_flag(ctx));
    gp_prof_end(ctx, GP_PROF_GRAM, uplo == GP_FULL ? 8.0 * n * (double)n + 8.0 * n * d : 8.0 * n * (n + 1.0) / 2.0 + 8.0 * n * d);
    return GP_OK;
}

gp_status gp_gram_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *theta, double *K, int ldk, int uplo) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, X && theta && K, "null pointer");
    GP_REQUIRE(ctx, n >= 0 && d >= 1 && d <= 64 && ldx >= n && ldk >= n, "bad dimensions (1 <= d <= 64)");
    if (n == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dX, *dK;
    GP_TRY(ws_get(ctx, WS_A, sizeof(double) * (size_t)n * d, &dX));
    GP_TRY(ws_get(ctx, WS_B, sizeof(double) * (size_t)n * n, &dK));
    GP_TRY(upload_2d(ctx, dX, n, X, ldx, n, d));
    if (uplo != GP_FULL) GP_TRY(upload_2d(ctx, dK, n, K, ldk, n, n));  // keep the caller's strict upper triangle
    GP_TRY(gp_gram_rbf_dev(ctx, dX, n, d, n, theta, dK, n, uplo));
    return download_2d(ctx, K, ldk, dK, n, n, n);
}

gp_status gp_dgram_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *theta, int pos, double *D, int ldd) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, X && theta && D, "null pointer");
    GP_REQUIRE(ctx, n >= 0 && d >= 1 && d <= 64 && ldx >= n && ldd >= n, "bad dimensions (1 <= d <= 64)");
    if (pos < 1 || pos > d + 2) { GP_SET_ERR(ctx, "hyper-parameter position %d outside 1..%d", pos, d + 2); return GP_ERANGE; }   // MatchError
    if (n == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dX, *dD;
    GP_TRY(ws_get(ctx, WS_A, sizeof(double) * (size_t)n * d, &dX));
    GP_TRY(ws_get(ctx, WS_B, sizeof(double) * (size_t)n * n, &dD));
    GP_TRY(upload_2d(ctx, dX, n, X, ldx, n, d));
    gp_prof_begin(ctx, GP_PROF_GRAM);
    gpk_dgram_sym(ctx->stream, dX, n, d, n, theta, pos, dD, n);
    gp_prof_end(ctx, GP_PROF_GRAM, 8.0 * n * (double)n + 8.0 * n * d);
    return download_2d(ctx, D, ldd, dD, n, n, n);
}

gp_status gp_cross_gram_rbf(gp_ctx *ctx, const double *Xs, int m, int ldxs, const double *X, int n, int ldx, int d, const double *theta, double *Ks, int ldks) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, Xs && X && theta && Ks, "null pointer");
    GP_REQUIRE(ctx, m >= 0 && n >= 0 && d >= 1 && d <= 64 && ldxs >= m && ldx >= n && ldks >= m, "bad dimensions (1 <= d <= 64)");
    if (m == 0 || n == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dXs, *dX, *dK;
    GP_TRY(ws_get(ctx, WS_A, sizeof(double) * (size_t)m * d, &dXs));
    GP_TRY(ws_get(ctx, WS_C, sizeof(double) * (size_t)n * d, &dX));
    GP_TRY(ws_get(ctx, WS_B, sizeof(double) * (size_t)m * n, &dK));
    GP_TRY(upload_2d(ctx, dXs, m, Xs, ldxs, m, d));
    GP_TRY(upload_2d(ctx, dX, n, X, ldx, n, d));
    gpk_gram_cross(ctx->stream, dXs, m, m, dX, n, n, d, theta, dK, m, gp_gram_flag(ctx));
    return download_2d(ctx, Ks, ldks, dK, m, m, n);
}

// ---- factorisation / solves ---------------------------------------------------------------------
gp_status gp_potrf_lower(gp_ctx *ctx, double *A, int n, int lda, int *info) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, A && n >= 0 && lda >= n, "bad matrix");
    if (info) *info = 0;
    if (n == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    const int np = gp_pad(n);
    double *dA;
    GP_TRY(ws_get(ctx, WS_B, sizeof(double) * (size_t)np * np, &dA));
    GP_TRY(upload_2d(ctx, dA, np, A, lda, n, n));
    gpk_pad_identity(ctx->stream, dA, n, np, np);
    GP_HIP(ctx, hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
    double *dinv;
    GP_TRY(ws_get(ctx, WS_E, sizeof(double) * (size_t)np * 16, &dinv));
    chol_blocked(ctx, dA, np, np, dinv, 0);
    gpk_zero_upper(ctx->stream, dA, n, np);
    int h = 0;
    GP_TRY(read_info(ctx, &h));
    if (h) { if (info) *info = h; GP_SET_ERR(ctx, "matrix not positive definite at pivot %d", h); return GP_ENOTPD; }
    return download_2d(ctx, A, lda, dA, np, n, n);
}

static gp_status trsm_impl(gp_ctx *ctx, int trans, const double *L, int n, int ldl, double *B, int nrhs, int ldb);

gp_status gp_trsm_lower(gp_ctx *ctx, int trans, const double *L, int n, int ldl, double *B, int nrhs, int ldb) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, L && B && n >= 0 && nrhs >= 0 && ldl >= n && ldb >= n, "bad arguments");
    if (n == 0 || nrhs == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    return trsm_impl(ctx, trans, L, n, ldl, B, nrhs, ldb);
}

gp_status gp_inv_lower(gp_ctx *ctx, const double *L, int n, int ldl, double *Linv, int ldi) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, L && Linv && n >= 0 && ldl >= n && ldi >= n, "bad arguments");
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Linv[i + (size_t)j * ldi] = (i == j) ? 1.0 : 0.0;
    return gp_trsm_lower(ctx, 0, L, n, ldl, Linv, n, ldi);
}

// ---- regression ---------------------------------------------------------------------------------
gp_status gp_model_refit_dev(gp_model *m, const double *theta, double sigma_noise) {
    if (!m) return GP_EINVAL;
    gp_ctx *ctx = m->ctx;
    GP_REQUIRE(ctx, m->has_x && theta, "model has no training inputs (built from a Gram matrix)");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    m->theta.assign(theta, theta + (m->kind == 1 ? 11 : m->d + 2));
    m->sigma_noise = sigma_noise;
    gp_prof_begin(ctx, GP_PROF_GRAM);
    if (m->kind == 1) gpk_co2_gram(ctx->stream, m->dX, m->n, m->dX, m->n, theta, 0, m->dL, m->ldl, 1, 0, std::isnan(sigma_noise) ? 0.0 : sigma_noise);
    else gpk_gram_sym(ctx->stream, m->dX, m->n, m->d, m->n, theta, m->dL, m->ldl, 0, std::isnan(sigma_noise) ? 0.0 : sigma_noise, gp_gram_flag(ctx), m->dcen);
    gp_prof_end(ctx, GP_PROF_GRAM, 8.0 * m->n * (m->n + 1.0) / 2.0 + 8.0 * m->n * m->d);
    model_factor(m);
    GP_LAUNCH_CHECK(ctx);
    return GP_OK;
}

gp_status gp_model_status(gp_model *m, int *info) {
    if (!m) return GP_EINVAL;
    GP_HIP(m->ctx, hipSetDevice(m->ctx->device));
    int h = 0;
    GP_TRY(read_info(m->ctx, &h));
    m->last_info = h;
    if (info) *info = h;
    if (h) { GP_SET_ERR(m->ctx, "matrix not positive definite at pivot %d", h); return GP_ENOTPD; }
    return GP_OK;
}

gp_status gp_fit_rbf_dev(gp_ctx *ctx, const double *dX, int n, int d, int ldx, const double *dy, const double *theta, double sigma_noise, gp_model **out, int *info) {
    if (!ctx || !out) return GP_EINVAL;
    *out = nullptr;
    GP_REQUIRE(ctx, dX && dy && theta, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && ldx >= n, "bad dimensions (n >= 1, 1 <= d <= 64)");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    gp_model *m = nullptr;
    GP_TRY(model_alloc(ctx, n, d, true, &m));
    gpk_copy_2d(ctx->stream, m->dX, n, dX, ldx, n, d);
    gpk_centroid(ctx->stream, m->dX, n, d, n, m->dcen);
    hipError_t e = hipMemcpyAsync(m->dy, dy, sizeof(double) * n, hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) { gp_model_destroy(m); GP_SET_ERR(ctx, "copy y: %s", hipGetErrorString(e)); return GP_EHIP; }
    gp_status st = gp_model_refit_dev(m, theta, sigma_noise);
    if (st == GP_OK) st = gp_model_status(m, info);
    if (st != GP_OK) { gp_model_destroy(m); return st; }
    *out = m;
    return GP_OK;
}

gp_status gp_fit_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *y, const double *theta, double sigma_noise, gp_model **out, int *info) {
    if (!ctx || !out) return GP_EINVAL;
    *out = nullptr;
    if (info) *info = 0;
    GP_REQUIRE(ctx, X && y && theta, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && ldx >= n, "bad dimensions (n >= 1, 1 <= d <= 64)");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    gp_model *m = nullptr;
    GP_TRY(model_alloc(ctx, n, d, true, &m));
    gp_status st = upload_2d(ctx, m->dX, n, X, ldx, n, d);
    if (st == GP_OK) gpk_centroid(ctx->stream, m->dX, n, d, n, m->dcen);
    if (st == GP_OK) st = upload_2d(ctx, m->dy, n, y, n, n, 1);
    if (st == GP_OK) st = gp_model_refit_dev(m, theta, sigma_noise);
    if (st == GP_OK) st = gp_model_status(m, info);
    if (st != GP_OK) { gp_model_destroy(m); return st; }
    *out = m;
    return GP_OK;
}

gp_status gp_fit_from_gram(gp_ctx *ctx, const double *K, int n, int ldk, const double *y, gp_model **out, int *info) {
    if (!ctx || !out) return GP_EINVAL;
    *out = nullptr;
    if (info) *info = 0;
    GP_REQUIRE(ctx, K && y && n >= 1 && ldk >= n, "bad arguments");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    gp_model *m = nullptr;
    GP_TRY(model_alloc(ctx, n, 0, false, &m));
    gp_status st = upload_2d(ctx, m->dL, m->ldl, K, ldk, n, n);
    if (st == GP_OK) st = upload_2d(ctx, m->dy, n, y, n, n, 1);
    if (st == GP_OK) {
        model_factor(m);
        gpk_zero_upper(ctx->stream, m->dL, n, m->ldl);
        st = gp_model_status(m, info);
    }
    if (st != GP_OK) { gp_model_destroy(m); return st; }
    *out = m;
    return GP_OK;
}

// GpPredictor.logLikelihoodWithDerivatives (gp/regression/GpPredictor.scala:60-80) for ANY KernelFunc: the host evaluates the kernel
// matrix and the P derivative matrices with the reference's own loops (buildKernelMatrix / buildMatrixWithFunc(trainingData)(
// derAfterHyperParam(i)), :62,74), the device does everything that is O(n^3) or O(P n^2): factorisation, alpha, K^-1 = L^-T L^-1
// (:66-67) and g_p = 1/2 tr((alpha alpha^T - K^-1) dK_p) (:76).  sigma_noise (NaN = None) is added un-squared to K's diagonal (:116).
gp_status gp_lml_grad_from_gram(gp_ctx *ctx, const double *K, int n, int ldk, const double *y, const double *const *dK, int nparams, int lddk,
                                double sigma_noise, double *lml, double *grad, int *info) {
    if (!ctx) return GP_EINVAL;
    if (info) *info = 0;
    GP_REQUIRE(ctx, K && y && lml && n >= 1 && ldk >= n && nparams >= 0 && (nparams == 0 || (dK && grad && lddk >= n)), "bad arguments");
    for (int p = 0; p < nparams; ++p) GP_REQUIRE(ctx, dK[p], "a derivative matrix pointer is NULL");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    gp_model *m = nullptr;
    GP_TRY(model_alloc(ctx, n, 0, false, &m));
    const int np = m->np;
    int h = 0;
    gp_status st = upload_2d(ctx, m->dL, m->ldl, K, ldk, n, n);
    if (st == GP_OK) st = upload_2d(ctx, m->dy, n, y, n, n, 1);
    if (st == GP_OK) {
        if (!std::isnan(sigma_noise)) gpk_add_diag(s, m->dL, n, m->ldl, sigma_noise);
        model_factor(m);
        st = gp_model_status(m, &h);
    }
    if (info) *info = h;
    if (st == GP_OK) st = download_2d(ctx, lml, 1, m->dlml, 1, 1, 1);
    if (st == GP_OK && nparams > 0) {
        double *T = nullptr, *Kinv = nullptr, *D = nullptr, *small = nullptr;
        st = ws_get(ctx, WS_VT, sizeof(double) * (size_t)np * np, &T);
        if (st == GP_OK) st = ws_get(ctx, WS_D, sizeof(double) * (size_t)np * np, &Kinv);
        if (st == GP_OK) st = ws_get(ctx, WS_B, sizeof(double) * (size_t)np * np, &D);
        if (st == GP_OK) st = ws_get(ctx, WS_C, sizeof(double) * ((size_t)2 * np + nparams), &small);   // alpha | partial | g[nparams]
        double *alpha = small, *partial = small + np, *g = partial + np;
        if (st == GP_OK) st = gpi_model_alpha(m, alpha);
        if (st == GP_OK) {
            inverse_transpose_lower(ctx, T, m->dL, np, m->ldl, m->ddinv);                           // T = L^-T
            gpk_gemm_nt(s, np, np, np, 1.0, T, np, T, np, 0.0, Kinv, np, 1, 1);                      // K^-1 = T T^T (lower)
        }
        for (int p = 0; p < nparams && st == GP_OK; ++p) {
            st = upload_2d(ctx, D, np, dK[p], lddk, n, n);        // stream-ordered: the trace of p - 1 has read D before this lands
            if (st == GP_OK) gpk_co2_trace(s, n, alpha, Kinv, np, D, np, partial, g + p);
        }
        if (st == GP_OK) st = download_2d(ctx, grad, nparams, g, nparams, nparams, 1);
    }
    if (st == GP_OK) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { GP_SET_ERR(ctx, "launch failed: %s", hipGetErrorString(e)); st = GP_EHIP; } }
    gp_model_destroy(m);
    return st;
}

gp_status gp_model_get(gp_model *m, int what, double *out, int ld) {
    if (!m || !out) return GP_EINVAL;
    gp_ctx *ctx = m->ctx;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    switch (what) {
        case GP_GET_L:
            GP_REQUIRE(ctx, ld >= m->n, "ld < n");
            return download_2d(ctx, out, ld, m->dL, m->ldl, m->n, m->n);
        case GP_GET_ALPHA:
            ensure_alpha(m);
            return download_2d(ctx, out, m->n, m->dalpha, m->np, m->n, 1);
        case GP_GET_LML:
            return download_2d(ctx, out, 1, m->dlml, 1, 1, 1);
        default:
            GP_SET_ERR(ctx, "unknown selector %d", what);
            return GP_EINVAL;
    }
}

void gp_model_destroy(gp_model *m) {
    if (!m) return;
    if (m->ctx) { (void)hipSetDevice(m->ctx->device); (void)hipStreamSynchronize(m->ctx->stream); }
    if (m->dX) (void)hipFree(m->dX);
    if (m->dcen) (void)hipFree(m->dcen);
    if (m->dy) (void)hipFree(m->dy);
    if (m->dL) (void)hipFree(m->dL);
    if (m->dalpha) (void)hipFree(m->dalpha);
    if (m->ddinv) (void)hipFree(m->ddinv);
    if (m->dtmp) (void)hipFree(m->dtmp);
    if (m->dlml) (void)hipFree(m->dlml);
    if (m->dLw) (void)hipFree(m->dLw);
    delete m;
}

// posterior at m test points already in HBM: mean, diagonal variance; Vt stays in the workspace
static gp_status predict_core(gp_model *mdl, const double *dXs, int m, int ldxs, double *dmean, double *dvar, double **vt_out, int *mp_out) {
    gp_ctx *ctx = mdl->ctx;
    hipStream_t s = ctx->stream;
    const int n = mdl->n, np = mdl->np, mp = gp_pad(m);
    const int nchunk = 16;
    double *Vt, *partial, *sumsq;
    GP_TRY(ws_get(ctx, WS_VT, sizeof(double) * (size_t)mp * np, &Vt));
    GP_TRY(ws_get(ctx, WS_PARTIAL, sizeof(double) * (size_t)nchunk * mp, &partial));
    GP_TRY(ws_get(ctx, WS_SUMSQ, sizeof(double) * (size_t)mp, &sumsq));
    if (np > n) gpk_fill(s, Vt + (size_t)n * mp, (size_t)mp * (np - n), 0.0);
    if (mp > m) {  // pad rows: keep them finite (rows never mix, but NaNs would slow nothing and help nobody)
        GP_HIP(ctx, hipMemset2DAsync(Vt + m, (size_t)mp * 8, 0, (size_t)(mp - m) * 8, n, s));
    }
    gp_prof_begin(ctx, GP_PROF_GRAM);
    if (mdl->kind == 1) gpk_co2_gram(s, dXs, m, mdl->dX, n, mdl->theta.data(), 0, Vt, mp, 0, 1, 0.0);
    else gpk_gram_cross(s, dXs, m, ldxs, mdl->dX, n, n, mdl->d, mdl->theta.data(), Vt, mp, gp_gram_flag(ctx), mdl->dcen);
    gp_prof_end(ctx, GP_PROF_GRAM, 8.0 * m * (double)n + 8.0 * (m + n) * mdl->d);
    // sumsq and the mean accumulate inside the row-panel solves: var = kss - |v|^2, mean = v . (L^-1 y)  (= K* alpha)
    double *dots = partial;
    GP_HIP(ctx, hipMemsetAsync(sumsq, 0, sizeof(double) * mp, s));
    GP_HIP(ctx, hipMemsetAsync(dots, 0, sizeof(double) * mp, s));
    const double *Lw = nullptr;
    const bool use_lw = [] { const char *e = getenv("GPCORE_POSTERIOR_LW"); return !e || atoi(e) != 0; }();
    if (use_lw && mp / GP_NB >= rows_left_min()) {   // large batch: fold the panel solves into the GEMMs (see build_lw)
        if (!mdl->dLw) {
            hipError_t e = hipMalloc(&mdl->dLw, sizeof(double) * (size_t)np * np);
            if (e != hipSuccess) { mdl->dLw = nullptr; GP_SET_ERR(ctx, "hipMalloc(Lw, n=%d) failed: %s", n, hipGetErrorString(e)); return GP_ENOMEM; }
        }
        if (!mdl->lw_valid) {
            double *scratch;
            GP_TRY(ws_get(ctx, WS_D, sizeof(double) * (size_t)np * np, &scratch));
            build_lw(ctx, mdl->dLw, scratch, mdl->dL, np, mdl->ldl, mdl->ddinv);
            mdl->lw_valid = true;
        }
        Lw = mdl->dLw;
    }
    solve_rows_lower(ctx, Vt, mp, mdl->dL, np, mdl->ldl, mdl->ddinv, sumsq, mdl->dtmp, dots, Lw);
    GP_HIP(ctx, hipMemcpyAsync(dmean, dots, sizeof(double) * m, hipMemcpyDeviceToDevice, s));
    if (dvar) {
        const double sf = mdl->theta[0], sn = mdl->kind == 1 ? 0.0 : mdl->theta[mdl->d + 1];
        gpk_var_finish(s, dvar, sumsq, m, mdl->kind == 1 ? gpk_co2_kss(mdl->theta.data()) : sf * sf + sn * sn);
    }
    if (vt_out) *vt_out = Vt;
    if (mp_out) *mp_out = mp;
    GP_LAUNCH_CHECK(ctx);
    return GP_OK;
}

gp_status gp_predict_dev(gp_model *mdl, const double *dXs, int m, int ldxs, double *dmean, double *dvar) {
    if (!mdl) return GP_EINVAL;
    gp_ctx *ctx = mdl->ctx;
    GP_REQUIRE(ctx, mdl->has_x, "model was built from a Gram matrix");
    GP_REQUIRE(ctx, dXs && dmean && m >= 1 && ldxs >= m, "bad arguments");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    // The (batch x n) workspace Vt is never materialised for all m at once (config C5: 10^6 x 32768 doubles = 262 GB):
    // test points go through in batches whose Vt stays under ~32 GiB; 65 536 rows per batch already fill the chip (one round
    // of 512 resident tiles), 131 072 (two rounds) measured +0.9 % at n = 8192, more rows nothing (GPCORE_PREDICT_BATCH).
    const size_t budget = (size_t)32 << 30;
    int batch = (int)std::min<size_t>((size_t)m, std::max<size_t>(GP_NB, (budget / ((size_t)mdl->np * 8)) / GP_NB * GP_NB));
    static const int batch_cap = [] { const char *e = getenv("GPCORE_PREDICT_BATCH"); int v = e ? atoi(e) : 0; return v >= GP_NB ? v / GP_NB * GP_NB : 131072; }();
    batch = std::min(batch, batch_cap);
    for (int lo = 0; lo < m; lo += batch) {
        const int mb = std::min(batch, m - lo);
        GP_TRY(predict_core(mdl, dXs + lo, mb, ldxs, dmean + lo, dvar ? dvar + lo : nullptr, nullptr, nullptr));
    }
    return GP_OK;
}

gp_status gp_predict(gp_model *mdl, const double *Xs, int m, int ldxs, double *mean, double *var_diag, double *cov, int ldc) {
    if (!mdl) return GP_EINVAL;
    gp_ctx *ctx = mdl->ctx;
    GP_REQUIRE(ctx, mdl->has_x, "model was built from a Gram matrix");
    GP_REQUIRE(ctx, Xs && mean && m >= 0 && ldxs >= m && (!cov || ldc >= m), "bad arguments");
    if (m == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    const int d = mdl->d, mp = gp_pad(m);
    double *dXs, *dout;
    GP_TRY(ws_get(ctx, WS_A, sizeof(double) * (size_t)m * d, &dXs));
    GP_TRY(ws_get(ctx, WS_C, sizeof(double) * (size_t)2 * mp, &dout));
    GP_TRY(upload_2d(ctx, dXs, m, Xs, ldxs, m, d));
    double *Vt = nullptr;
    int mp2 = 0;
    GP_TRY(predict_core(mdl, dXs, m, m, dout, dout + mp, &Vt, &mp2));
    GP_TRY(download_2d(ctx, mean, m, dout, m, m, 1));
    if (var_diag) GP_TRY(download_2d(ctx, var_diag, m, dout + mp, m, m, 1));
    if (cov) {
        // Sigma* = buildKernelMatrix(X*) - V^T V  (GpPredictor.scala:36): symmetric Gram of the test points
        // (sn^2 on its diagonal) minus Vt Vt^T on the MFMA syrk, then mirrored on the host copy-out.
        double *dC;
        GP_TRY(ws_get(ctx, WS_D, sizeof(double) * (size_t)mp * mp, &dC));
        if (mdl->kind == 1) gpk_co2_gram(ctx->stream, dXs, m, dXs, m, mdl->theta.data(), 0, dC, mp, 1, 1, 0.0);
        else gpk_gram_sym(ctx->stream, dXs, m, d, m, mdl->theta.data(), dC, mp, 1, 0.0, gp_gram_flag(ctx));
        gpk_pad_identity(ctx->stream, dC, m, mp, mp);
        gp_prof_begin(ctx, GP_PROF_SYRK);
        gpk_gemm_nt(ctx->stream, mp, mp, mdl->np, -1.0, Vt, mp, Vt, mp, 1.0, dC, mp, 1);
        gp_prof_end(ctx, GP_PROF_SYRK, syrk_flops(mp, mdl->np));
        GP_TRY(download_2d(ctx, cov, ldc, dC, mp, m, m));
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < j; ++i) cov[i + (size_t)j * ldc] = cov[j + (size_t)i * ldc];
    }
    return GP_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Posterior from a factor the CALLER holds: GpPredictor.computePosterior(trainingData, testData, l, alphaVec[, kernelFunc])
// gp/regression/GpPredictor.scala:45-58 -- the entry GP-UCB (gp/optimization/GPOptimizer.scala:91) and the GP-UKF
// (dynamicalsystems/filtering/GPUnscentedKalmanFilter.scala:78-87,141-142) call.  Shared core: Vt (mp x np, one row per test
// point) arrives holding K*; mean = K* alpha is taken first (gemv, columns ascending like Breeze's dgemv), then
// Vt <- Vt L^-T (= V^T) with the row sums of squares for the diagonal variance.
// ------------------------------------------------------------------------------------------------
namespace {

struct host_factor {          // (L, alpha) uploaded into padded device buffers
    double *dL = nullptr, *dinv = nullptr, *dalpha = nullptr;
    int n = 0, np = 0;
};

gp_status upload_factor(gp_ctx *ctx, const double *L, int n, int ldl, const double *alpha, host_factor *f) {
    hipStream_t s = ctx->stream;
    const int np = gp_pad(n);
    f->n = n, f->np = np;
    GP_TRY(ws_get(ctx, WS_B, sizeof(double) * (size_t)np * np, &f->dL));
    GP_TRY(ws_get(ctx, WS_E, sizeof(double) * (size_t)np * 16, &f->dinv));
    GP_TRY(ws_get(ctx, WS_F, sizeof(double) * (size_t)np, &f->dalpha));
    GP_HIP(ctx, hipMemsetAsync(f->dL, 0, sizeof(double) * (size_t)np * np, s));
    GP_HIP(ctx, hipMemsetAsync(f->dalpha, 0, sizeof(double) * (size_t)np, s));
    GP_TRY(upload_2d(ctx, f->dL, np, L, ldl, n, n));
    gpk_zero_upper(s, f->dL, n, np);          // callers pass breeze's cholesky output, but a full matrix must not leak in
    gpk_pad_identity(s, f->dL, n, np, np);
    gpk_tile_inverses(s, f->dL, np, np, f->dinv);
    GP_TRY(upload_2d(ctx, f->dalpha, n, alpha, n, n, 1));
    return GP_OK;
}

// Vt holds K* (m x n valid, pads zero) on entry and V^T on exit; dout = [mean (mp) | var or sumsq (mp)]
gp_status posterior_rows(gp_ctx *ctx, double *Vt, int m, int mp, const host_factor &f, double *dmean, double *dsumsq) {
    hipStream_t s = ctx->stream;
    double *partial;
    GP_TRY(ws_get(ctx, WS_PARTIAL, sizeof(double) * (size_t)16 * mp, &partial));
    gpk_gemv_rows(s, Vt, m, f.n, mp, f.dalpha, dmean, partial, 16);
    GP_HIP(ctx, hipMemsetAsync(dsumsq, 0, sizeof(double) * mp, s));
    solve_rows_lower(ctx, Vt, mp, f.dL, f.np, f.np, f.dinv, dsumsq);
    GP_LAUNCH_CHECK(ctx);
    return GP_OK;
}

// cov (m x m, host) = dC - Vt Vt^T, dC (mp x mp, lower valid, pad identity) already holding the test Gram matrix
gp_status cov_finish(gp_ctx *ctx, double *dC, int m, int mp, const double *Vt, int np, double *cov, int ldc) {
    gp_prof_begin(ctx, GP_PROF_SYRK);
    gpk_gemm_nt(ctx->stream, mp, mp, np, -1.0, Vt, mp, Vt, mp, 1.0, dC, mp, 1);
    gp_prof_end(ctx, GP_PROF_SYRK, syrk_flops(mp, np));
    GP_TRY(download_2d(ctx, cov, ldc, dC, mp, m, m));
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < j; ++i) cov[i + (size_t)j * ldc] = cov[j + (size_t)i * ldc];
    return GP_OK;
}

// V (n x m, host) = (Vt)^T
gp_status v_download(gp_ctx *ctx, const double *Vt, int m, int mp, int n, int np, double *V, int ldv) {
    double *dV;
    GP_TRY(ws_get(ctx, WS_G, sizeof(double) * (size_t)np * mp, &dV));
    gpk_transpose(ctx->stream, dV, np, Vt, mp, mp, np);
    return download_2d(ctx, V, ldv, dV, np, n, m);
}

}  // namespace

extern "C" gp_status gp_posterior_from_factor(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *theta, const double *L, int ldl,
                                              const double *alpha, const double *Xs, int m, int ldxs, double *mean, double *var_diag,
                                              double *cov, int ldc, double *V, int ldv) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, X && theta && L && alpha && Xs && mean, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && ldx >= n && ldl >= n && m >= 0 && ldxs >= m, "bad dimensions (n >= 1, 1 <= d <= 64)");
    GP_REQUIRE(ctx, (!cov || ldc >= m) && (!V || ldv >= n), "bad leading dimension of an output");
    if (m == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    host_factor f;
    GP_TRY(upload_factor(ctx, L, n, ldl, alpha, &f));
    const int np = f.np, mp = gp_pad(m);
    double *dX, *dXs, *Vt, *dout;
    GP_TRY(ws_get(ctx, WS_A, sizeof(double) * (size_t)m * d, &dXs));
    GP_TRY(ws_get(ctx, WS_H, sizeof(double) * (size_t)n * d, &dX));
    GP_TRY(ws_get(ctx, WS_VT, sizeof(double) * (size_t)mp * np, &Vt));
    GP_TRY(ws_get(ctx, WS_C, sizeof(double) * (size_t)2 * mp, &dout));
    GP_TRY(upload_2d(ctx, dXs, m, Xs, ldxs, m, d));
    GP_TRY(upload_2d(ctx, dX, n, X, ldx, n, d));
    GP_HIP(ctx, hipMemsetAsync(Vt, 0, sizeof(double) * (size_t)mp * np, s));
    gp_prof_begin(ctx, GP_PROF_GRAM);
    gpk_gram_cross(s, dXs, m, m, dX, n, n, d, theta, Vt, mp, gp_gram_flag(ctx));
    gp_prof_end(ctx, GP_PROF_GRAM, 8.0 * m * (double)n + 8.0 * (m + n) * d);
    GP_TRY(posterior_rows(ctx, Vt, m, mp, f, dout, dout + mp));
    GP_TRY(download_2d(ctx, mean, m, dout, m, m, 1));
    if (var_diag) {
        const double sf = theta[0], sn = theta[d + 1];
        gpk_var_finish(s, dout + mp, dout + mp, m, sf * sf + sn * sn);
        GP_TRY(download_2d(ctx, var_diag, m, dout + mp, m, m, 1));
    }
    if (cov) {   // buildKernelMatrix(kernelFunc, testData) - vMatrix.t * vMatrix   (:56)
        double *dC;
        GP_TRY(ws_get(ctx, WS_D, sizeof(double) * (size_t)mp * mp, &dC));
        gpk_gram_sym(s, dXs, m, d, m, theta, dC, mp, 1, 0.0, gp_gram_flag(ctx));
        gpk_pad_identity(s, dC, m, mp, mp);
        GP_TRY(cov_finish(ctx, dC, m, mp, Vt, np, cov, ldc));
    }
    if (V) GP_TRY(v_download(ctx, Vt, m, mp, n, np, V, ldv));
    return GP_OK;
}

// Same for ANY KernelFunc: the caller evaluated K* = buildKernelMatrix(kernelFunc, testData, trainingData) (m x n) and, for the
// covariance, K** = buildKernelMatrix(kernelFunc, testData) (m x m) -- or only its diagonal -- on the host.
extern "C" gp_status gp_posterior_from_gram(gp_ctx *ctx, const double *Ks, int m, int n, int ldks, const double *Kss, int ldkss,
                                            const double *kss_diag, const double *L, int ldl, const double *alpha, double *mean,
                                            double *var_diag, double *cov, int ldc, double *V, int ldv) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, Ks && L && alpha && mean, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && m >= 0 && ldks >= m && ldl >= n, "bad dimensions");
    GP_REQUIRE(ctx, (!cov || (Kss && ldkss >= m && ldc >= m)) && (!var_diag || Kss || kss_diag) && (!V || ldv >= n) && (!Kss || ldkss >= m),
               "cov needs Kss, var_diag needs Kss or kss_diag");
    if (m == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    host_factor f;
    GP_TRY(upload_factor(ctx, L, n, ldl, alpha, &f));
    const int np = f.np, mp = gp_pad(m);
    double *Vt, *dout;
    GP_TRY(ws_get(ctx, WS_VT, sizeof(double) * (size_t)mp * np, &Vt));
    GP_TRY(ws_get(ctx, WS_C, sizeof(double) * (size_t)2 * mp, &dout));
    GP_HIP(ctx, hipMemsetAsync(Vt, 0, sizeof(double) * (size_t)mp * np, s));
    GP_TRY(upload_2d(ctx, Vt, mp, Ks, ldks, m, n));
    GP_TRY(posterior_rows(ctx, Vt, m, mp, f, dout, dout + mp));
    GP_TRY(download_2d(ctx, mean, m, dout, m, m, 1));
    if (var_diag) {
        GP_TRY(download_2d(ctx, var_diag, m, dout + mp, m, m, 1));
        for (int i = 0; i < m; ++i) var_diag[i] = (Kss ? Kss[i + (size_t)i * ldkss] : kss_diag[i]) - var_diag[i];
    }
    if (cov) {
        double *dC;
        GP_TRY(ws_get(ctx, WS_D, sizeof(double) * (size_t)mp * mp, &dC));
        GP_HIP(ctx, hipMemsetAsync(dC, 0, sizeof(double) * (size_t)mp * mp, s));
        GP_TRY(upload_2d(ctx, dC, mp, Kss, ldkss, m, m));
        GP_TRY(cov_finish(ctx, dC, m, mp, Vt, np, cov, ldc));
    }
    if (V) GP_TRY(v_download(ctx, Vt, m, mp, n, np, V, ldv));
    return GP_OK;
}

// predict() for a model fitted with gp_fit_from_gram (GpPredictor.predict with a user KernelFunc such as Co2Kernel,
// gp/regression/Co2Prediction.scala:29-137): the factor and alpha are already resident, only K* / K** come from the host.
extern "C" gp_status gp_predict_from_gram(gp_model *mdl, const double *Ks, int m, int ldks, const double *Kss, int ldkss,
                                          const double *kss_diag, double *mean, double *var_diag, double *cov, int ldc) {
    if (!mdl) return GP_EINVAL;
    gp_ctx *ctx = mdl->ctx;
    GP_REQUIRE(ctx, Ks && mean && m >= 0 && ldks >= m, "bad arguments");
    GP_REQUIRE(ctx, (!cov || (Kss && ldkss >= m && ldc >= m)) && (!var_diag || Kss || kss_diag) && (!Kss || ldkss >= m),
               "cov needs Kss, var_diag needs Kss or kss_diag");
    if (m == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int n = mdl->n, np = mdl->np, mp = gp_pad(m);
    ensure_alpha(mdl);
    host_factor f;
    f.dL = mdl->dL, f.dinv = mdl->ddinv, f.dalpha = mdl->dalpha, f.n = n, f.np = np;
    double *Vt, *dout;
    GP_TRY(ws_get(ctx, WS_VT, sizeof(double) * (size_t)mp * np, &Vt));
    GP_TRY(ws_get(ctx, WS_C, sizeof(double) * (size_t)2 * mp, &dout));
    GP_HIP(ctx, hipMemsetAsync(Vt, 0, sizeof(double) * (size_t)mp * np, s));
    GP_TRY(upload_2d(ctx, Vt, mp, Ks, ldks, m, n));
    {   // posterior_rows with the model's leading dimension (ldl = np + 128)
        double *partial;
        GP_TRY(ws_get(ctx, WS_PARTIAL, sizeof(double) * (size_t)16 * mp, &partial));
        gpk_gemv_rows(s, Vt, m, n, mp, f.dalpha, dout, partial, 16);
        GP_HIP(ctx, hipMemsetAsync(dout + mp, 0, sizeof(double) * mp, s));
        solve_rows_lower(ctx, Vt, mp, mdl->dL, np, mdl->ldl, mdl->ddinv, dout + mp);
        GP_LAUNCH_CHECK(ctx);
    }
    GP_TRY(download_2d(ctx, mean, m, dout, m, m, 1));
    if (var_diag) {
        GP_TRY(download_2d(ctx, var_diag, m, dout + mp, m, m, 1));
        for (int i = 0; i < m; ++i) var_diag[i] = (Kss ? Kss[i + (size_t)i * ldkss] : kss_diag[i]) - var_diag[i];
    }
    if (cov) {
        double *dC;
        GP_TRY(ws_get(ctx, WS_D, sizeof(double) * (size_t)mp * mp, &dC));
        GP_HIP(ctx, hipMemsetAsync(dC, 0, sizeof(double) * (size_t)mp * mp, s));
        GP_TRY(upload_2d(ctx, dC, mp, Kss, ldkss, m, m));
        GP_TRY(cov_finish(ctx, dC, m, mp, Vt, np, cov, ldc));
    }
    return GP_OK;
}

// ------------------------------------------------------------------------------------------------
// generic triangular solves with many right-hand sides, through the row-panel machinery:
//   trans = 0:  L X = B    <=>  X^T = B^T L^-T      (forward over block columns)
//   trans = 1:  L^T X = B  <=>  X^T = B^T L^-1      (backward; needs the NN product, done on L^T copy)
// ------------------------------------------------------------------------------------------------

static gp_status trsm_impl(gp_ctx *ctx, int trans, const double *L, int n, int ldl, double *B, int nrhs, int ldb) {
    const int np = gp_pad(n), mp = gp_pad(nrhs);
    double *dL, *dB, *dVt;
    GP_TRY(ws_get(ctx, WS_B, sizeof(double) * (size_t)np * np, &dL));
    GP_TRY(ws_get(ctx, WS_D, sizeof(double) * (size_t)np * mp, &dB));
    GP_TRY(ws_get(ctx, WS_VT, sizeof(double) * (size_t)mp * np, &dVt));
    hipStream_t s = ctx->stream;
    GP_HIP(ctx, hipMemsetAsync(dL, 0, sizeof(double) * (size_t)np * np, s));
    GP_HIP(ctx, hipMemsetAsync(dB, 0, sizeof(double) * (size_t)np * mp, s));
    GP_TRY(upload_2d(ctx, dL, np, L, ldl, n, n));
    gpk_zero_upper(s, dL, n, np);
    gpk_pad_identity(s, dL, n, np, np);
    GP_TRY(upload_2d(ctx, dB, np, B, ldb, n, nrhs));
    gpk_transpose(s, dVt, mp, dB, np, np, mp);  // Vt = B^T  (mp x np)
    if (!trans) {
        double *dinv;
        GP_TRY(ws_get(ctx, WS_C, sizeof(double) * (size_t)np * 16, &dinv));
        gpk_tile_inverses(s, dL, np, np, dinv);
        solve_rows_lower(ctx, dVt, mp, dL, np, np, dinv, nullptr);
    } else {
        double *dU;
        GP_TRY(ws_get(ctx, WS_E, sizeof(double) * (size_t)np * np, &dU));
        gpk_transpose(s, dU, np, dL, np, np, np);  // U = L^T (upper), so X^T = B^T U^-T ... with U upper
        solve_rows_upper(ctx, dVt, mp, dU, np, np);
    }
    gpk_transpose(s, dB, np, dVt, mp, mp, np);
    return download_2d(ctx, B, ldb, dB, np, n, nrhs);
}

// ------------------------------------------------------------------------------------------------
// LML + gradient at B hyper-parameter settings (GpPredictor.logLikelihoodWithDerivatives :60-80):
//   fit -> (L, t = L^-1 y, LML);  T = L^-T (rows of I solved against L);  Kinv = T T^T (MFMA syrk, k >= row block);
//   alpha = T t;  fused traces over W = alpha alpha^T - Kinv.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int LML_GEMV_CHUNKS = 32;

// One worker = one context (own stream + workspaces) that evaluates settings G at a time IN LOCKSTEP: the G problems have
// identical shapes, so every step of the factorisation / triangular inversion is ONE launch covering all of them
// (blockIdx.y = problem).  The latency-bound diagonal-block chain is paid once per group instead of once per setting and
// the K = 128 updates launch G times the tiles, which is what fills 256 CUs at n = 4096.
struct lml_worker {
    gp_ctx *ctx = nullptr;
    int G = 1, n = 0, d = 0, np = 0, ldl = 0;
    double *dX = nullptr, *dy = nullptr, *L = nullptr, *T = nullptr, *Kinv = nullptr, *partials = nullptr, *gemv_part = nullptr, *small = nullptr;
    int *info = nullptr;
    size_t sL = 0, sT = 0, sSmall = 0, sPart = 0;
    std::vector<double> hres;
    std::vector<int> hinfo;
    gp_status st = GP_OK;
    // per-problem small block: dinv (np*16) | t (np) | alpha (np) | res (P+1, padded to 72)
    double *dinv(int g) const { return small + g * sSmall; }
    double *tvec(int g) const { return dinv(g) + (size_t)np * 16; }
    double *alpha(int g) const { return tvec(g) + np; }
    double *res(int g) const { return alpha(g) + np; }
};

size_t lml_bytes_per_setting(int np, bool with_grad) {
    return sizeof(double) * ((size_t)(np + GP_NB) * np + (with_grad ? 2 * (size_t)np * np : 0) + (size_t)np * 64);
}

gp_status lml_worker_setup(lml_worker &w, const double *X, int n, int d, int ldx, const double *y, int nparams) {
    gp_ctx *ctx = w.ctx;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    const int np = ((n + GP_NB - 1) / GP_NB) * GP_NB, G = w.G;
    w.n = n, w.d = d, w.np = np, w.ldl = np + GP_NB;
    w.sL = (size_t)w.ldl * np, w.sT = (size_t)np * np, w.sSmall = (size_t)np * 18 + 72;
    w.sPart = (size_t)gpk_lml_grad_partials_size(n, d);
    GP_TRY(ws_get(ctx, WS_A, sizeof(double) * ((size_t)n * d + np), &w.dX));
    w.dy = w.dX + (size_t)n * d;
    // L and T keep zeros that nothing in this path overwrites (the factors' upper triangles and the 127 rows under y^T; the blocks
    // below T's diagonal): a call that finds its own tag on the slot does not clear 4 GB per worker again
    const unsigned long long tag = 0x4c4d4c0000000000ull | ((unsigned long long)np << 16) | (unsigned long long)G;
    bool keptL = false, keptT = false;
    GP_TRY(ws_get_keep(ctx, WS_B, sizeof(double) * w.sL * G, &w.L, tag, &keptL));
    GP_TRY(ws_get(ctx, WS_C, sizeof(double) * w.sSmall * G, &w.small));
    double *ip = nullptr;
    GP_TRY(ws_get(ctx, WS_SUMSQ, sizeof(double) * (size_t)(G + 2), &ip));
    w.info = reinterpret_cast<int *>(ip);
    if (nparams > 0) {
        GP_TRY(ws_get_keep(ctx, WS_VT, sizeof(double) * w.sT * G, &w.T, tag, &keptT));
        if (!keptT) GP_HIP(ctx, hipMemsetAsync(w.T, 0, sizeof(double) * w.sT * G, ctx->stream));
        GP_TRY(ws_get(ctx, WS_D, sizeof(double) * w.sT * G, &w.Kinv));
        GP_TRY(ws_get(ctx, WS_PARTIAL, sizeof(double) * w.sPart * G, &w.partials));
        GP_TRY(ws_get(ctx, WS_E, sizeof(double) * (size_t)LML_GEMV_CHUNKS * np * G, &w.gemv_part));
    }
    GP_TRY(upload_2d(ctx, w.dX, n, X, ldx, n, d));
    GP_HIP(ctx, hipMemsetAsync(w.dy, 0, sizeof(double) * np, ctx->stream));
    GP_TRY(upload_2d(ctx, w.dy, n, y, n, n, 1));
    if (!keptL) GP_HIP(ctx, hipMemsetAsync(w.L, 0, sizeof(double) * w.sL * G, ctx->stream));   // upper triangles and the 127 spare rows under y^T
    w.hres.assign((size_t)72 * G, 0.0);
    w.hinfo.assign(G, 0);
    return GP_OK;
}

// settings thetas[0 .. g) (g <= G) -> lml[0 .. g), grad, info
gp_status lml_worker_eval(lml_worker &w, const double *thetas, int g, int nparams, double sigma_noise, double *lml, double *grad, int *info) {
    gp_ctx *ctx = w.ctx;
    hipStream_t s = ctx->stream;
    const int np = w.np, n = w.n, d = w.d, P = d + 2, ldl = w.ldl;
    const double extra = std::isnan(sigma_noise) ? 0.0 : sigma_noise;
    GP_HIP(ctx, hipMemsetAsync(w.info, 0, sizeof(int) * g, s));
    gpk_centroid(s, w.dX, n, d, n, gp_gram_center(ctx));      // one centre for all settings of the group (it does not depend on theta)
    for (int j = 0; j < g; ++j) {
        double *Lj = w.L + j * w.sL;
        gp_prof_begin(ctx, GP_PROF_GRAM);
        gpk_gram_sym(s, w.dX, n, d, n, thetas + (size_t)j * P, Lj, ldl, 0, extra, gp_gram_flag(ctx), gp_gram_center(ctx));
        gp_prof_end(ctx, GP_PROF_GRAM, 8.0 * n * (n + 1.0) / 2.0 + 8.0 * n * d);
        gpk_pad_identity(s, Lj, n, np, ldl);
        gpk_copy_strided(s, Lj + np, (size_t)ldl, w.dy, 1, np);       // y^T rides through the factorisation in row np
    }
    chol_blocked(ctx, w.L, np, ldl, w.small, GP_NB, g, w.sL, w.sSmall, w.info);
    for (int j = 0; j < g; ++j) {
        const double *Lj = w.L + j * w.sL;
        gpk_copy_strided(s, w.tvec(j), 1, Lj + np, (size_t)ldl, np);   // t = L^-1 y
        gpk_lml(s, Lj, n, ldl, w.tvec(j), w.res(j));
    }
    if (nparams > 0) {
        inverse_transpose_lower(ctx, w.T, w.L, np, ldl, w.small, g, w.sT, w.sL, w.sSmall, true);     // T = L^-T (upper triangular)
        gp_batch bk;
        bk.count = g, bk.s0 = bk.s1 = bk.s2 = w.sT;
        gp_prof_begin(ctx, GP_PROF_SYRK);
        gpk_gemm_nt(s, np, np, np, 1.0, w.T, np, w.T, np, 0.0, w.Kinv, np, 1, 1, bk);          // Kinv = T T^T, lower
        gp_prof_end(ctx, GP_PROF_SYRK, (double)g * np * np * np / 3.0);
        for (int j = 0; j < g; ++j) {
            // alpha = L^-T t = T t: one matrix-vector pass over the T already here, not np/128 substitution steps
            gpk_gemv_rows(s, w.T + j * w.sT, np, np, np, w.tvec(j), w.alpha(j), w.gemv_part + (size_t)j * LML_GEMV_CHUNKS * np, LML_GEMV_CHUNKS, 1);
            gpk_lml_grad_traces(s, w.dX, n, d, n, thetas + (size_t)j * P, w.alpha(j), w.Kinv + j * w.sT, np, w.partials + j * w.sPart, w.res(j) + 1);
        }
    }
    GP_LAUNCH_CHECK(ctx);
    GP_HIP(ctx, hipMemcpy2DAsync(w.hres.data(), 72 * sizeof(double), w.res(0), w.sSmall * sizeof(double), (P + 1) * sizeof(double), g,
                                 hipMemcpyDeviceToHost, s));
    GP_HIP(ctx, hipMemcpyAsync(w.hinfo.data(), w.info, sizeof(int) * g, hipMemcpyDeviceToHost, s));
    GP_HIP(ctx, hipStreamSynchronize(s));
    for (int j = 0; j < g; ++j) {
        const int h = w.hinfo[j];
        if (h) { ws_forget(ctx, WS_B); ws_forget(ctx, WS_VT); }    // a failed factorisation may have left NaN in the rows that ride along
        if (info) info[j] = h;
        lml[j] = h ? NAN : w.hres[(size_t)72 * j];
        for (int p = 0; p < nparams; ++p) grad[(size_t)j * nparams + p] = h ? NAN : w.hres[(size_t)72 * j + 1 + p];
    }
    return GP_OK;
}

}  // namespace

// Settings are independent and identically shaped.  They are evaluated in lockstep groups of G (one launch per algorithm
// step for the whole group); a second worker (own context, stream and host thread) runs another group concurrently so
// that one group's single-workgroup diagonal steps overlap the other's GEMMs.  GPCORE_LML_GROUP (default 32, capped by free
// HBM) and GPCORE_LML_WORKERS (default 2) set the shape; results do not depend on either.
extern "C" gp_status gp_lml_grad_rbf_batched(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *y, const double *thetas,
                                             int B, int nparams, double sigma_noise, double *lml, double *grad, int *info) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, X && y && thetas && lml, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && ldx >= n && B >= 0, "bad dimensions");
    const int P = d + 2;
    GP_REQUIRE(ctx, nparams >= 0 && nparams <= P && (nparams == 0 || grad), "0 <= nparams <= d+2");
    if (B == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    int nw = 2, G = 32;
    if (const char *e = getenv("GPCORE_LML_WORKERS")) nw = atoi(e);
    if (const char *e = getenv("GPCORE_LML_GROUP")) G = atoi(e);
    G = std::max(1, std::min(G, 32));
    nw = std::max(1, std::min(std::min(nw, 8), (B + G - 1) / G));
    G = std::min(G, (B + nw - 1) / nw);
    {   // keep the groups within half of the HBM that is free right now
        size_t free_b = 0, total_b = 0;
        GP_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
        const int np = ((n + GP_NB - 1) / GP_NB) * GP_NB;
        const size_t per = lml_bytes_per_setting(np, nparams > 0);
        while (G > 1 && (size_t)nw * G * per > free_b / 2) --G;
        while (nw > 1 && (size_t)nw * G * per > free_b / 2) --nw;
    }
    // helper contexts are cached on the caller's context
    ctx_ext *x = ext_of(ctx);
    while ((int)x->children.size() < nw - 1) {
        gp_ctx *c = nullptr;
        g_child_ctx = true;
        const gp_status cst = gp_ctx_create(ctx->device, nullptr, &c);
        g_child_ctx = false;
        if (cst != GP_OK) break;
        x->children.push_back(c);
    }
    nw = std::min(nw, (int)x->children.size() + 1);
    std::vector<lml_worker> ws(nw);
    for (int k = 0; k < nw; ++k) { ws[k].ctx = (k == 0) ? ctx : x->children[k - 1]; ws[k].G = G; }
    std::atomic<int> next{0};
    auto run = [&](int k) {
        lml_worker &w = ws[k];
        w.st = lml_worker_setup(w, X, n, d, ldx, y, nparams);
        while (w.st == GP_OK) {
            const int b = next.fetch_add(G);
            if (b >= B) break;
            const int g = std::min(G, B - b);
            w.st = lml_worker_eval(w, thetas + (size_t)b * P, g, nparams, sigma_noise, lml + b, grad ? grad + (size_t)b * nparams : nullptr,
                                   info ? info + b : nullptr);
        }
        // a worker that found no group left still has its uploads of the caller's X, y in flight: the host buffers may
        // be released as soon as this function returns
        (void)hipStreamSynchronize(w.ctx->stream);
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < nw; ++k) threads.emplace_back(run, k);
    run(0);
    for (auto &t : threads) t.join();
    for (int k = 0; k < nw; ++k)
        if (ws[k].st != GP_OK) {
            if (k > 0) GP_SET_ERR(ctx, "%s", ws[k].ctx->err);
            return ws[k].st;
        }
    return GP_OK;
}

// ------------------------------------------------------------------------------------------------
// GpPredictor.obtainOptimalHyperParams (gp/regression/GpPredictor.scala:126-142) driving
// BreezeLbfgsOptimizer.maximize (optimization/Optimization.scala:30-63): L-BFGS with `history` correction pairs and at
// most max_iter iterations on f(x) = -LML over the first nparams entries of theta (vector order), returning the best point
// any evaluation saw (:44-46,52-55).  Breeze's LBFGS is third-party code outside the reference tree, so its exact iterates
// are not reproducible: what is kept is the interface, the memory/iteration limits, the objective and the
// best-seen rule.  The line search is batched: the NC trial steps alpha0 * 2^-c of one iteration are ONE lockstep group on
// the device (training data stays resident for the whole run), so an iteration costs one group evaluation however many
// trial steps the Armijo test rejects.
// ------------------------------------------------------------------------------------------------
namespace {

struct lbfgs_pair { std::vector<double> s, y; double rho; };

double vdot(const std::vector<double> &a, const std::vector<double> &b) {
    double r = 0.0;
    for (size_t i = 0; i < a.size(); ++i) r += a[i] * b[i];
    return r;
}

}  // namespace

// Shared driver: maximise F over the first nparams entries of theta.  `evaluate(thetas, count, f, g, bad)` fills, for `count`
// settings (rows of P doubles), f = F, g = dF/dtheta (count x nparams) and bad[c] != 0 where F is undefined (not positive definite).
gp_status gpi_lbfgs_maximize(gp_ctx *ctx, int P, int nparams, const double *theta0, int max_iter, int history, int NC,
                             const std::function<gp_status(const double *, int, double *, double *, int *)> &evaluate_raw,
                             double *theta_out, double *f_out, int *iters_out, int *evals_out) {
    std::vector<double> theta(theta0, theta0 + P), thetas((size_t)NC * P), fl(NC), gl((size_t)NC * nparams);
    std::vector<int> infos(NC);
    int evals = 0;
    // f = -F, g = -grad at `count` settings; undefined settings come back as +inf
    auto evaluate = [&](int count) -> gp_status {
        GP_TRY(evaluate_raw(thetas.data(), count, fl.data(), gl.data(), infos.data()));
        evals += count;
        for (int c = 0; c < count; ++c) {
            bool bad = infos[c] != 0 || !std::isfinite(fl[c]);
            for (int p = 0; p < nparams && !bad; ++p) bad = !std::isfinite(gl[(size_t)c * nparams + p]);
            fl[c] = bad ? INFINITY : -fl[c];
            for (int p = 0; p < nparams; ++p) gl[(size_t)c * nparams + p] = bad ? 0.0 : -gl[(size_t)c * nparams + p];
        }
        return GP_OK;
    };
    std::copy(theta.begin(), theta.end(), thetas.begin());
    GP_TRY(evaluate(1));
    if (!std::isfinite(fl[0])) { GP_SET_ERR(ctx, "the starting hyper-parameters do not give a positive definite matrix (pivot %d)", infos[0]); return GP_ENOTPD; }
    std::vector<double> x(theta.begin(), theta.begin() + nparams), g(gl.begin(), gl.begin() + nparams);
    double f = fl[0];
    std::vector<double> best_x = x;
    double best_f = f;
    std::vector<lbfgs_pair> hist;
    int it = 0;
    for (; it < max_iter; ++it) {
        const double gnorm = std::sqrt(vdot(g, g));
        if (gnorm <= 1e-9 * std::max(1.0, std::fabs(f))) break;
        // two-loop recursion: p = -H g
        std::vector<double> q = g, al(hist.size());
        for (int k = (int)hist.size() - 1; k >= 0; --k) {
            al[k] = hist[k].rho * vdot(hist[k].s, q);
            for (int i = 0; i < nparams; ++i) q[i] -= al[k] * hist[k].y[i];
        }
        double scale = hist.empty() ? 1.0 / gnorm : vdot(hist.back().s, hist.back().y) / vdot(hist.back().y, hist.back().y);
        for (int i = 0; i < nparams; ++i) q[i] *= scale;
        for (size_t k = 0; k < hist.size(); ++k) {
            const double be = hist[k].rho * vdot(hist[k].y, q);
            for (int i = 0; i < nparams; ++i) q[i] += hist[k].s[i] * (al[k] - be);
        }
        std::vector<double> pdir(nparams);
        for (int i = 0; i < nparams; ++i) pdir[i] = -q[i];
        double slope = vdot(g, pdir);
        if (!(slope < 0.0)) {   // not a descent direction: drop the history, steepest descent
            hist.clear();
            for (int i = 0; i < nparams; ++i) pdir[i] = -g[i] / gnorm;
            slope = -gnorm;
        }
        // batched backtracking: up to 3 groups of NC halvings
        int chosen = -1;
        double alpha0 = 1.0, alpha = 0.0;
        for (int round = 0; round < 3 && chosen < 0; ++round, alpha0 *= std::ldexp(1.0, -NC)) {
            for (int c = 0; c < NC; ++c) {
                const double a = std::ldexp(alpha0, -c);
                std::copy(theta.begin(), theta.end(), thetas.begin() + (size_t)c * P);
                for (int i = 0; i < nparams; ++i) thetas[(size_t)c * P + i] = x[i] + a * pdir[i];
            }
            GP_TRY(evaluate(NC));
            for (int c = 0; c < NC; ++c) {
                if (fl[c] < best_f) { best_f = fl[c]; for (int i = 0; i < nparams; ++i) best_x[i] = thetas[(size_t)c * P + i]; }
                if (chosen < 0 && fl[c] <= f + 1e-4 * std::ldexp(alpha0, -c) * slope) { chosen = c; alpha = std::ldexp(alpha0, -c); }
            }
        }
        if (chosen < 0) break;   // no trial step decreases f: converged to the resolution of the line search
        lbfgs_pair pr;
        pr.s.resize(nparams), pr.y.resize(nparams);
        for (int i = 0; i < nparams; ++i) {
            pr.s[i] = alpha * pdir[i];
            pr.y[i] = gl[(size_t)chosen * nparams + i] - g[i];
            x[i] += pr.s[i];
            g[i] = gl[(size_t)chosen * nparams + i];
        }
        const double fnew = fl[chosen], sy = vdot(pr.s, pr.y);
        if (sy > 1e-12 * std::sqrt(vdot(pr.s, pr.s) * vdot(pr.y, pr.y))) {
            pr.rho = 1.0 / sy;
            if ((int)hist.size() == history) hist.erase(hist.begin());
            hist.push_back(pr);
        }
        const bool stalled = std::fabs(f - fnew) <= 1e-10 * std::max(1.0, std::fabs(f));
        f = fnew;
        if (stalled) { ++it; break; }
    }
    std::copy(theta.begin(), theta.end(), theta_out);
    for (int i = 0; i < nparams; ++i) theta_out[i] = best_x[i];
    if (f_out) *f_out = -best_f;
    if (iters_out) *iters_out = it;
    if (evals_out) *evals_out = evals;
    return GP_OK;
}

extern "C" gp_status gp_optimize_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *y, const double *theta0, int nparams,
                                     double sigma_noise, int max_iter, int history, double *theta_out, double *lml_out, int *iters_out,
                                     int *evals_out) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, X && y && theta0 && theta_out, "null pointer");
    GP_REQUIRE(ctx, n >= 1 && d >= 1 && d <= 64 && ldx >= n, "bad dimensions");
    const int P = d + 2;
    GP_REQUIRE(ctx, nparams >= 1 && nparams <= P && max_iter >= 0 && history >= 1, "1 <= nparams <= d+2, max_iter >= 0, history >= 1");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    constexpr int NC = 6;   // trial steps per iteration: alpha0, alpha0/2, ..., alpha0/32
    lml_worker w;
    w.ctx = ctx;
    w.G = NC;
    GP_TRY(lml_worker_setup(w, X, n, d, ldx, y, nparams));
    auto evaluate = [&](const double *thetas, int count, double *f, double *g, int *bad) -> gp_status {
        return lml_worker_eval(w, thetas, count, nparams, sigma_noise, f, g, bad);
    };
    return gpi_lbfgs_maximize(ctx, P, nparams, theta0, max_iter, history, NC, evaluate, theta_out, lml_out, iters_out, evals_out);
}

// ------------------------------------------------------------------------------------------------
// EP classification -- implemented in gpcore_ep.hip
// ------------------------------------------------------------------------------------------------
