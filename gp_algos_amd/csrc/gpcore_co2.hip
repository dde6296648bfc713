// C-ABI for the composite CO2 kernel (gp/regression/Co2Prediction.scala:29-137) -- a user-defined KernelFunc of the reference
// with 11 hyper-parameters on one-dimensional inputs -- so that GpPredictor with `co2Kernel` (config/spring-context.xml:29-31,
// 49-51; Co2Prediction.main :188-219, MasterThesisRelatedTasks.evaluateGpPredictionOnCo2Ds) runs on the device end to end:
// Gram / cross-Gram / derivative Gram, fit (a gp_model that gp_predict / gp_model_get accept), log marginal likelihood with all 11
// derivatives (GpPredictor.logLikelihoodWithDerivatives :60-80 with Co2Kernel.derAfterHyperParam) and the L-BFGS fit
// (obtainOptimalHyperParams :126-142).
#include "gpcore_internal.h"

#include <new>
#include <vector>

namespace {
constexpr int CO2_P = 11;
}

extern "C" {

gp_status gp_gram_co2(gp_ctx *ctx, const double *x, int n, const double *theta, double *K, int ldk, int uplo) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, x && theta && K && n >= 0 && ldk >= n, "bad arguments");
    if (n == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dx, *dK;
    GP_TRY(gpi_ws_get(ctx, WS_A, sizeof(double) * (size_t)n, &dx));
    GP_TRY(gpi_ws_get(ctx, WS_B, sizeof(double) * (size_t)n * n, &dK));
    GP_TRY(gpi_upload_2d(ctx, dx, n, x, n, n, 1));
    if (uplo != GP_FULL) GP_TRY(gpi_upload_2d(ctx, dK, n, K, ldk, n, n));   // keep the caller's strict upper triangle
    gpk_co2_gram(ctx->stream, dx, n, dx, n, theta, 0, dK, n, 1, uplo == GP_FULL, 0.0);
    return gpi_download_2d(ctx, K, ldk, dK, n, n, n);
}

gp_status gp_dgram_co2(gp_ctx *ctx, const double *x, int n, const double *theta, int pos, double *D, int ldd) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, x && theta && D && n >= 0 && ldd >= n, "bad arguments");
    if (pos < 1 || pos > CO2_P) { GP_SET_ERR(ctx, "hyper-parameter position %d outside 1..11", pos); return GP_ERANGE; }   // MatchError (:73-83)
    if (n == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dx, *dD;
    GP_TRY(gpi_ws_get(ctx, WS_A, sizeof(double) * (size_t)n, &dx));
    GP_TRY(gpi_ws_get(ctx, WS_B, sizeof(double) * (size_t)n * n, &dD));
    GP_TRY(gpi_upload_2d(ctx, dx, n, x, n, n, 1));
    gpk_co2_gram(ctx->stream, dx, n, dx, n, theta, pos, dD, n, 1, 1, 0.0);
    return gpi_download_2d(ctx, D, ldd, dD, n, n, n);
}

gp_status gp_cross_gram_co2(gp_ctx *ctx, const double *xs, int m, const double *x, int n, const double *theta, double *Ks, int ldks) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, xs && x && theta && Ks && m >= 0 && n >= 0 && ldks >= m, "bad arguments");
    if (m == 0 || n == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    double *dxs, *dx, *dK;
    GP_TRY(gpi_ws_get(ctx, WS_A, sizeof(double) * (size_t)m, &dxs));
    GP_TRY(gpi_ws_get(ctx, WS_C, sizeof(double) * (size_t)n, &dx));
    GP_TRY(gpi_ws_get(ctx, WS_B, sizeof(double) * (size_t)m * n, &dK));
    GP_TRY(gpi_upload_2d(ctx, dxs, m, xs, m, m, 1));
    GP_TRY(gpi_upload_2d(ctx, dx, n, x, n, n, 1));
    gpk_co2_gram(ctx->stream, dxs, m, dx, n, theta, 0, dK, m, 0, 1, 0.0);
    return gpi_download_2d(ctx, Ks, ldks, dK, m, m, n);
}

gp_status gp_fit_co2(gp_ctx *ctx, const double *x, int n, const double *y, const double *theta, double sigma_noise, gp_model **out, int *info) {
    if (!ctx || !out) return GP_EINVAL;
    *out = nullptr;
    if (info) *info = 0;
    GP_REQUIRE(ctx, x && y && theta && n >= 1, "bad arguments");
    GP_HIP(ctx, hipSetDevice(ctx->device));
    gp_model *m = nullptr;
    GP_TRY(gpi_model_alloc(ctx, n, 1, true, &m));
    m->kind = 1;
    gp_status st = gpi_upload_2d(ctx, m->dX, n, x, n, n, 1);
    if (st == GP_OK) st = gpi_upload_2d(ctx, m->dy, n, y, n, n, 1);
    if (st == GP_OK) st = gp_model_refit_dev(m, theta, sigma_noise);
    if (st == GP_OK) st = gp_model_status(m, info);
    if (st != GP_OK) { gp_model_destroy(m); return st; }
    *out = m;
    return GP_OK;
}

gp_status gp_lml_grad_co2_batched(gp_ctx *ctx, const double *x, int n, const double *y, const double *thetas, int B, int nparams,
                                  double sigma_noise, double *lml, double *grad, int *info) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, x && y && thetas && lml && n >= 1 && B >= 0 && nparams >= 0 && nparams <= CO2_P && (nparams == 0 || grad),
               "bad arguments (0 <= nparams <= 11)");
    if (B == 0) return GP_OK;
    GP_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    gp_model *m = nullptr;
    GP_TRY(gpi_model_alloc(ctx, n, 1, true, &m));
    m->kind = 1;
    gp_status st = gpi_upload_2d(ctx, m->dX, n, x, n, n, 1);
    if (st == GP_OK) st = gpi_upload_2d(ctx, m->dy, n, y, n, n, 1);
    const int np = m->np;
    double *T = nullptr, *Kinv = nullptr, *D = nullptr, *small = nullptr;
    if (st == GP_OK && nparams > 0) {
        st = gpi_ws_get(ctx, WS_VT, sizeof(double) * (size_t)np * np, &T);
        if (st == GP_OK) st = gpi_ws_get(ctx, WS_D, sizeof(double) * (size_t)np * np, &Kinv);
        if (st == GP_OK) st = gpi_ws_get(ctx, WS_B, sizeof(double) * (size_t)np * np, &D);
    }
    if (st == GP_OK) st = gpi_ws_get(ctx, WS_C, sizeof(double) * ((size_t)2 * np + CO2_P + 2), &small);   // alpha | partial | g[11] | lml
    for (int b = 0; b < B && st == GP_OK; ++b) {
        const double *th = thetas + (size_t)b * CO2_P;
        st = gp_model_refit_dev(m, th, sigma_noise);
        int h = 0;
        if (st == GP_OK) st = gp_model_status(m, &h);
        if (info) info[b] = h;
        if (st == GP_ENOTPD) {   // this setting only: NaN results, go on (lml = NaN there, like gp_lml_grad_rbf_batched)
            lml[b] = NAN;
            for (int p = 0; p < nparams; ++p) grad[(size_t)b * nparams + p] = NAN;
            st = GP_OK;
            continue;
        }
        if (st != GP_OK) break;
        st = gp_model_get(m, GP_GET_LML, lml + b, 1);
        if (st != GP_OK || nparams == 0) continue;
        double *alpha = small, *partial = small + np, *g = partial + np;
        st = gpi_model_alpha(m, alpha);
        if (st != GP_OK) break;
        gpi_inverse_transpose_lower(ctx, T, m->dL, np, m->ldl, m->ddinv);                       // T = L^-T
        gpk_gemm_nt(s, np, np, np, 1.0, T, np, T, np, 0.0, Kinv, np, 1, 1);                      // K^-1 = T T^T (lower)
        for (int p = 0; p < nparams; ++p) {                                                    // g_p = 1/2 tr((alpha alpha^T - K^-1) dK/dhp_p)  (:70-78)
            gpk_co2_gram(s, m->dX, n, m->dX, n, th, p + 1, D, np, 1, 0, 0.0);
            gpk_co2_trace(s, n, alpha, Kinv, np, D, np, partial, g + p);
        }
        st = gpi_download_2d(ctx, grad + (size_t)b * nparams, nparams, g, nparams, nparams, 1);
    }
    gp_model_destroy(m);
    return st;
}

gp_status gp_optimize_co2(gp_ctx *ctx, const double *x, int n, const double *y, const double *theta0, int nparams, double sigma_noise,
                          int max_iter, int history, double *theta_out, double *lml_out, int *iters_out, int *evals_out) {
    if (!ctx) return GP_EINVAL;
    GP_REQUIRE(ctx, x && y && theta0 && theta_out && n >= 1 && nparams >= 1 && nparams <= CO2_P && max_iter >= 0 && history >= 1, "bad arguments");
    constexpr int NC = 4;
    auto evaluate = [&](const double *thetas, int count, double *f, double *g, int *bad) -> gp_status {
        return gp_lml_grad_co2_batched(ctx, x, n, y, thetas, count, nparams, sigma_noise, f, g, bad);
    };
    return gpi_lbfgs_maximize(ctx, CO2_P, nparams, theta0, max_iter, history, NC, evaluate, theta_out, lml_out, iters_out, evals_out);
}

}  // extern "C"
