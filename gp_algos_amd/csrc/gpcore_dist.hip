// Multi-GPU entry points of the C-ABI: one process per GPU, RCCL over xGMI.
//
// The hot path shards over INDEPENDENT units only -- hyper-parameter settings (config C3:
// GpPredictor.logLikelihoodWithDerivatives evaluated by obtainOptimalHyperParams / the mesh evaluator,
// gp/regression/GpPredictor.scala:60-80,126-142) and test points (config C5: GpPredictor.predict :24-43) -- so the only
// collective on the data path is ONE ncclAllGather of the per-rank results (preceded by an all-gather of one status word per
// rank, so that a rank whose local part failed -- GP_ENOMEM, GP_EHIP, a bad argument only it sees -- cannot leave its peers
// blocked inside the collective: every rank learns of the failure and returns, the failing rank with its own status, the
// others with GP_EPEER); the Cholesky itself stays single-GPU.
// Every rank passes the SAME inputs, evaluates the contiguous slice  [rank * ceil(U/G), ...)  of the U units on its own
// device and receives the assembled result.  Message sizes: (B/G) x (2 + nparams) doubles (C3: 8 x 12 x 8 B = 768 B per
// rank) or 2 m/G doubles (C5: 2 MB per rank) -- far below one xGMI link's 153 GB/s; no reduction, no ring of large tensors.
//
// RCCL is resolved at run time (a process that already carries an RCCL -- torch.distributed's -- is reused, otherwise
// librccl.so.1 is loaded), so libgpcore.so itself has no link-time dependency on it and single-GPU hosts never load it.
#include "gpcore_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <vector>

namespace {

// the slice of rccl.h this file uses (types are ABI-stable: an opaque 128-byte id passed by value, an opaque communicator)
struct rccl_unique_id { char internal[GP_DIST_ID_BYTES]; };
typedef void *rccl_comm_t;
constexpr int RCCL_FLOAT64 = 8;   // ncclFloat64 / ncclDouble

struct rccl_api {
    int (*GetUniqueId)(rccl_unique_id *) = nullptr;
    int (*CommInitRank)(rccl_comm_t *, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};

rccl_api &rccl() {
    static rccl_api api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        if (!dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
            h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
            if (!h) return;
        }
        void *from = h ? h : RTLD_DEFAULT;
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(from, "ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(from, "ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(from, "ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(from, "ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(from, "ncclGetErrorString"));
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.GetErrorString;
    });
    return api;
}

}  // namespace

struct gp_dist {
    gp_ctx *ctx = nullptr;
    rccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
    double *send = nullptr, *recv = nullptr;   // device staging of the all-gather
    size_t cap = 0;                            // doubles per rank the staging buffers hold
    double *st_send = nullptr, *st_recv = nullptr;   // status words (1 per rank), allocated with the communicator
    int fail_next = 0;                         // fault injection for the tests (gp_dist_inject_failure): next call fails locally
};

#define GP_RCCL(ctx, call) do { int r_ = (call); if (r_ != 0) { \
    GP_SET_ERR(ctx, "%s:%d %s -> %s", __FILE__, __LINE__, #call, rccl().GetErrorString(r_)); return GP_ERCCL; } } while (0)

namespace {

gp_status dist_reserve(gp_dist *d, size_t per_rank) {
    if (d->cap >= per_rank) return GP_OK;
    gp_ctx *ctx = d->ctx;
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (d->send) (void)hipFree(d->send);
    if (d->recv) (void)hipFree(d->recv);
    d->send = d->recv = nullptr;
    d->cap = 0;
    hipError_t e = hipMalloc(&d->send, sizeof(double) * per_rank);
    if (e == hipSuccess) e = hipMalloc(&d->recv, sizeof(double) * per_rank * (size_t)d->world);
    if (e != hipSuccess) { GP_SET_ERR(ctx, "hipMalloc of the all-gather staging failed: %s", hipGetErrorString(e)); return GP_ENOMEM; }
    d->cap = per_rank;
    return GP_OK;
}

// Every rank's status word to every rank, BEFORE the payload collective.  `local` is what this rank's own part returned
// (evaluation, staging allocation).  Uses only buffers that exist since gp_dist_init, so it cannot fail locally short of a
// dead device or communicator -- and then RCCL's own abort handling is what the peers see.  All ranks return non-zero or
// all return GP_OK: nobody enters the payload all-gather alone.
gp_status dist_agree(gp_dist *d, gp_status local) {
    gp_ctx *ctx = d->ctx;
    const double mine = (double)(int)local;
    std::vector<double> all((size_t)d->world, 0.0);
    GP_HIP(ctx, hipMemcpyAsync(d->st_send, &mine, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    GP_RCCL(ctx, rccl().AllGather(d->st_send, d->st_recv, 1, RCCL_FLOAT64, d->comm, ctx->stream));
    GP_HIP(ctx, hipMemcpyAsync(all.data(), d->st_recv, sizeof(double) * all.size(), hipMemcpyDeviceToHost, ctx->stream));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int bad = -1;
    const gp_status s = gp_dist_status_scan(all.data(), d->world, d->rank, &bad);
    if (s == GP_EPEER) GP_SET_ERR(ctx, "rank %d of the group failed with status %d; no result was exchanged", bad, (int)all[(size_t)bad]);
    return s;
}

// host slice (count doubles, zero-padded to per_rank) -> every rank's slices, rank-major, on the host.  The caller has
// reserved the staging (dist_reserve) and agreed with its peers (dist_agree) before.
gp_status dist_allgather(gp_dist *d, const double *mine, size_t count, size_t per_rank, std::vector<double> &all) {
    gp_ctx *ctx = d->ctx;
    std::vector<double> pad(per_rank, 0.0);
    std::copy(mine, mine + count, pad.begin());
    GP_HIP(ctx, hipMemcpyAsync(d->send, pad.data(), sizeof(double) * per_rank, hipMemcpyHostToDevice, ctx->stream));
    GP_RCCL(ctx, rccl().AllGather(d->send, d->recv, per_rank, RCCL_FLOAT64, d->comm, ctx->stream));
    all.resize(per_rank * (size_t)d->world);
    GP_HIP(ctx, hipMemcpyAsync(all.data(), d->recv, sizeof(double) * all.size(), hipMemcpyDeviceToHost, ctx->stream));
    GP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GP_OK;
}

}  // namespace

extern "C" {

gp_status gp_dist_unique_id(gp_ctx *ctx, unsigned char *id) {
    if (!ctx || !id) return GP_EINVAL;
    if (!rccl().ok) { GP_SET_ERR(ctx, "RCCL is not available (librccl.so.1 not found)"); return GP_ERCCL; }
    GP_HIP(ctx, hipSetDevice(ctx->device));
    rccl_unique_id u;
    GP_RCCL(ctx, rccl().GetUniqueId(&u));
    memcpy(id, u.internal, GP_DIST_ID_BYTES);
    return GP_OK;
}

gp_status gp_dist_init(gp_ctx *ctx, const unsigned char *id, int rank, int world, gp_dist **out) {
    if (!ctx || !out) return GP_EINVAL;
    *out = nullptr;
    GP_REQUIRE(ctx, id && world >= 1 && rank >= 0 && rank < world, "need an id, world >= 1 and 0 <= rank < world");
    if (!rccl().ok) { GP_SET_ERR(ctx, "RCCL is not available (librccl.so.1 not found)"); return GP_ERCCL; }
    GP_HIP(ctx, hipSetDevice(ctx->device));
    gp_dist *d = new (std::nothrow) gp_dist();
    if (!d) return GP_ENOMEM;
    d->ctx = ctx, d->rank = rank, d->world = world;
    rccl_unique_id u;
    memcpy(u.internal, id, GP_DIST_ID_BYTES);
    int r = rccl().CommInitRank(&d->comm, world, u, rank);
    if (r != 0) {
        GP_SET_ERR(ctx, "ncclCommInitRank(rank %d of %d) -> %s", rank, world, rccl().GetErrorString(r));
        delete d;
        return GP_ERCCL;
    }
    hipError_t e = hipMalloc(&d->st_send, sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d->st_recv, sizeof(double) * (size_t)world);
    if (e != hipSuccess) {
        GP_SET_ERR(ctx, "hipMalloc of the status words failed: %s", hipGetErrorString(e));
        gp_dist_destroy(d);
        return GP_ENOMEM;
    }
    *out = d;
    return GP_OK;
}

void gp_dist_destroy(gp_dist *d) {
    if (!d) return;
    if (d->ctx) { (void)hipSetDevice(d->ctx->device); (void)hipStreamSynchronize(d->ctx->stream); }
    if (d->comm) (void)rccl().CommDestroy(d->comm);
    if (d->send) (void)hipFree(d->send);
    if (d->recv) (void)hipFree(d->recv);
    if (d->st_send) (void)hipFree(d->st_send);
    if (d->st_recv) (void)hipFree(d->st_recv);
    delete d;
}

gp_status gp_dist_status_scan(const double *status, int world, int rank, int *bad_rank) {
    if (!status || world < 1 || rank < 0 || rank >= world) return GP_EINVAL;
    int first = -1;
    for (int r = 0; r < world; ++r)
        if (status[r] != 0.0) { first = r; break; }
    if (bad_rank) *bad_rank = first;
    if (first < 0) return GP_OK;
    if (status[rank] != 0.0) { if (bad_rank) *bad_rank = rank; return (gp_status)(int)status[rank]; }
    return GP_EPEER;
}

gp_status gp_dist_inject_failure(gp_dist *d) {
    if (!d) return GP_EINVAL;
    d->fail_next = 1;
    return GP_OK;
}

gp_status gp_dist_shard(const gp_dist *d, int total, int *lo, int *hi) {
    if (!d || !lo || !hi || total < 0) return GP_EINVAL;
    const int per = (total + d->world - 1) / d->world;
    *lo = std::min(total, d->rank * per);
    *hi = std::min(total, *lo + per);
    return GP_OK;
}

gp_status gp_dist_lml_grad_batched(gp_dist *d, const double *X, int n, int dd, int ldx, const double *y, const double *thetas, int B,
                                   int nparams, double sigma_noise, double *lml, double *grad, int *info) {
    if (!d) return GP_EINVAL;   // no communicator: nothing to exchange a status over
    gp_ctx *ctx = d->ctx;
    // Nothing between here and dist_agree returns.  A rank that fails locally -- a bad argument only it sees included -- still
    // takes part in the status exchange: its peers are already on their way into that all-gather.  (B and nparams size the
    // exchange itself, so the ranks must agree on them like on the communicator; a rank that disagrees is caught by RCCL.)
    gp_status local = GP_OK;
    if (!(X && y && thetas && lml && B >= 0 && nparams >= 0 && (nparams == 0 || grad))) {
        GP_SET_ERR(ctx, "invalid argument: bad arguments");
        local = GP_EINVAL;
    }
    const int Bs = std::max(B, 0), NPs = std::max(nparams, 0);
    if (local == GP_OK && B == 0) return GP_OK;   // an empty batch is empty on every rank: no collective at all
    const int P = dd + 2, per = (Bs + d->world - 1) / d->world;
    int lo = 0, hi = 0;
    if (gp_dist_shard(d, Bs, &lo, &hi) != GP_OK) lo = hi = 0;
    const int mine = hi - lo, W = 2 + NPs;   // per setting: lml | info | grad[nparams]
    std::vector<double> l(std::max(mine, 1)), g((size_t)std::max(mine, 1) * std::max(NPs, 1));
    std::vector<int> inf(std::max(mine, 1), 0);
    if (local == GP_OK) local = dist_reserve(d, (size_t)per * (2 + NPs));
    if (local == GP_OK && mine > 0)
        local = gp_lml_grad_rbf_batched(ctx, X, n, dd, ldx, y, thetas + (size_t)lo * P, mine, nparams, sigma_noise, l.data(), g.data(), inf.data());
    if (local == GP_OK && d->fail_next) local = GP_ENOMEM, d->fail_next = 0;
    GP_TRY(dist_agree(d, local));
    std::vector<double> pack((size_t)std::max(mine, 1) * W, 0.0), all;
    for (int b = 0; b < mine; ++b) {
        pack[(size_t)b * W] = l[b];
        pack[(size_t)b * W + 1] = (double)inf[b];
        for (int p = 0; p < nparams; ++p) pack[(size_t)b * W + 2 + p] = g[(size_t)b * nparams + p];
    }
    GP_TRY(dist_allgather(d, pack.data(), (size_t)mine * W, (size_t)per * W, all));
    for (int b = 0; b < B; ++b) {
        const int r = b / per, j = b - r * per;
        const double *row = all.data() + ((size_t)r * per + j) * W;
        lml[b] = row[0];
        if (info) info[b] = (int)row[1];
        for (int p = 0; p < nparams; ++p) grad[(size_t)b * nparams + p] = row[2 + p];
    }
    return GP_OK;
}

gp_status gp_dist_predict(gp_dist *d, gp_model *model, const double *Xs, int m, int ldxs, double *mean, double *var) {
    if (!d) return GP_EINVAL;   // no communicator: nothing to exchange a status over
    gp_ctx *ctx = d->ctx;
    // as above: every local check feeds the status exchange instead of returning ahead of it
    gp_status local = GP_OK;
    if (!(model && Xs && mean && var && m >= 0 && ldxs >= m)) {
        GP_SET_ERR(ctx, "invalid argument: bad arguments");
        local = GP_EINVAL;
    } else if (model->ctx != ctx) {
        GP_SET_ERR(ctx, "invalid argument: the model must live on the context the communicator was created on");
        local = GP_EINVAL;
    }
    if (local == GP_OK && m == 0) return GP_OK;   // an empty request is empty on every rank
    const int ms = std::max(m, 0), per = (ms + d->world - 1) / d->world;
    int lo = 0, hi = 0;
    if (gp_dist_shard(d, ms, &lo, &hi) != GP_OK) lo = hi = 0;
    const int mine = hi - lo;
    std::vector<double> pack((size_t)2 * std::max(mine, 1), 0.0), all;
    if (local == GP_OK) local = dist_reserve(d, (size_t)2 * per);
    if (local == GP_OK && mine > 0) local = gp_predict(model, Xs + lo, mine, ldxs, pack.data(), pack.data() + mine, nullptr, 0);
    if (local == GP_OK && d->fail_next) local = GP_ENOMEM, d->fail_next = 0;
    GP_TRY(dist_agree(d, local));
    // interleave as [mean slice | var slice] per rank: two contiguous runs of `mine` doubles, padded to `per` each
    std::vector<double> slot((size_t)2 * per, 0.0);
    std::copy(pack.begin(), pack.begin() + mine, slot.begin());
    std::copy(pack.begin() + mine, pack.begin() + 2 * (size_t)mine, slot.begin() + per);
    GP_TRY(dist_allgather(d, slot.data(), (size_t)2 * per, (size_t)2 * per, all));
    for (int r = 0; r < d->world; ++r) {
        const int rlo = std::min(m, r * per), rhi = std::min(m, rlo + per);
        const double *base = all.data() + (size_t)r * 2 * per;
        for (int i = rlo; i < rhi; ++i) { mean[i] = base[i - rlo]; var[i] = base[per + (i - rlo)]; }
    }
    return GP_OK;
}

}  // extern "C"
