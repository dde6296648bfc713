// EP binary classification (EpParameterEstimator.scala:29-109, GpClassifier.scala:24-47).
#include "gpcore_internal.h"

extern "C" {

gp_status gp_ep_create(gp_ctx *ctx, const double *, int, int, const int32_t *, gp_ep **out) {
    if (out) *out = nullptr;
    GP_SET_ERR(ctx, "EP path not built yet");
    return GP_EINVAL;
}
gp_status gp_ep_sweep(gp_ep *, int, double *, double *, int *) { return GP_EINVAL; }
gp_status gp_ep_lml(gp_ep *, int, double *) { return GP_EINVAL; }
gp_status gp_ep_get(gp_ep *, int, double *, int) { return GP_EINVAL; }
gp_status gp_ep_predict(gp_ep *, const double *, int, int, const double *, double *) { return GP_EINVAL; }
void gp_ep_destroy(gp_ep *) {}

}  // extern "C"
