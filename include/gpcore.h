/*
 * gpcore.h -- flat C-ABI of libgpcore.so, the MI355X-native (gfx950) Gaussian-process hot path.
 *
 * This header is the drop-in boundary: a JNI / ctypes / cgo binding needs
 * nothing but these declarations.  No C++ or torch types cross it: plain pointers, ints, doubles.
 * Every entry point names the reference code it replaces (paths relative to
 * /root/reference/src/main/scala/).
 *
 * Conventions
 *  - All matrices are COLUMN-MAJOR with an explicit leading dimension, matching Breeze
 *    DenseMatrix (data, offset, majorStride): element (i,j) = ptr[i + j*ld].
 *  - theta = [sf, l_1 .. l_d, sn], length d+2, the order of GaussianRbfParams.toDenseVector
 *    (utils/KernelRequisites.scala:48-52).  sf and sn enter the kernel SQUARED.
 *  - Host-pointer entry points copy in/out and never retain host pointers past the call.
 *    `_dev` entry points take device pointers (memory already resident in HBM) and run
 *    asynchronously on the context's stream; call gp_ctx_sync() before reading results.
 *  - Every function returns a gp_status; gp_last_error(ctx) gives the message.  Nothing aborts,
 *    nothing prints.  There is NO CPU fallback: without a usable GPU gp_ctx_create fails.
 *  - A context is bound to one device and one stream; calls on one context are serialised by the
 *    caller; different contexts may be used from different threads.
 */
#ifndef GPCORE_H
#define GPCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gp_status {
    GP_OK = 0,
    GP_EINVAL = 1, /* bad argument (Scala: require/assert -> IllegalArgumentException)           */
    GP_ENOTPD = 2, /* matrix not positive definite; *info = 1-based failing pivot (Breeze throws)  */
    GP_ENOMEM = 3, /* host or device allocation failed                                            */
    GP_EHIP = 4,   /* HIP runtime error (message in gp_last_error)                                */
    GP_ERANGE = 5, /* 1-based hyper-parameter position out of range (Scala: MatchError)           */
    GP_ERCCL = 6,  /* RCCL missing or a collective failed (message in gp_last_error)               */
    GP_EPEER = 7   /* gp_dist_*: another rank of the group failed; nothing was exchanged, this rank is intact */
} gp_status;

typedef struct gp_ctx gp_ctx;     /* device + stream + workspaces                                  */
typedef struct gp_model gp_model; /* fitted regression model: X, L, alpha resident in HBM          */
typedef struct gp_ep gp_ep;       /* EP classification state: K, Sigma, L, site parameters in HBM  */
typedef struct gp_dist gp_dist;   /* RCCL communicator of one rank (one process per GPU)           */
typedef struct gp_small gp_small; /* G small GP models over one set of training inputs (GP-UCB, GP-UKF) */

/* ---- library / context ------------------------------------------------------------------- */
const char *gp_version(void);
/* device = HIP ordinal.  stream = NULL: the context creates its own stream; otherwise an existing
 * hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) is adopted, not owned. */
gp_status gp_ctx_create(int device, void *stream, gp_ctx **out);
void gp_ctx_destroy(gp_ctx *ctx);
gp_status gp_ctx_sync(gp_ctx *ctx);
/* Releases the context's cached workspaces (posterior batches, lockstep LML groups, ... can hold tens of GB between calls in a
 * long-lived process); models and EP states are untouched, the next call that needs a workspace allocates it again. */
gp_status gp_ctx_trim(gp_ctx *ctx);
const char *gp_last_error(const gp_ctx *ctx);
/* Per-kernel timing with HIP events on the context's stream.  `mask` is an OR of (1 << GP_PROF_x);
 * while set, every launch of those kernel classes is bracketed by events.  gp_ctx_profile_read
 * returns launches, total ms and the algorithmic work (flops or bytes) the launches of one class
 * carried, then resets that class's counters.  mask = 0 switches profiling off. */
enum { GP_PROF_OFF = 0, GP_PROF_GEMM = 1, GP_PROF_SYRK = 2, GP_PROF_GRAM = 3, GP_PROF_TRSM = 4, GP_PROF_POTRF_DIAG = 5, GP_PROF_PANEL_UPD = 6, GP_PROF_NCLASSES = 7 };
gp_status gp_ctx_profile(gp_ctx *ctx, int mask);
gp_status gp_ctx_profile_read(gp_ctx *ctx, int which, int64_t *launches, double *total_ms, double *work);
/* Look-ahead of the blocked Cholesky (far part of an outer trailing update on the context's CU-masked side stream, under the next
 * panel's factorisation): 1 on, 0 off, -1 (default) chosen by size.  Per-kernel timings (gp_ctx_profile) of overlapping launches
 * add up to more than the wall time, so a measurement of the trailing-update kernel alone switches it off. */
gp_status gp_ctx_set_lookahead(gp_ctx *ctx, int mode);
/* The task list behind the single-launch form of the factorisation (breeze.linalg.cholesky, gp/regression/GpPredictor.scala:120, for ONE
 * matrix of n (padded) rows with `extra_rows` riding along, outer panels of 512 columns): HOST ONLY, no device -- builds the list for
 * `workgroups` workers, verifies it (every tile of the two-level right-looking scheme updated by every panel, in order; every
 * dependency pointing backwards in the list) and returns its length and the makespan of the schedule under the cost model, in us.
 * GP_EINVAL for a shape the single-launch form does not take. */
gp_status gp_chol_plan_info(int n, int extra_rows, int workgroups, int *ntasks, double *model_us);
/* fp64 MFMA peak probe: runs a register-only v_mfma_f64_16x16x4_f64 loop on every CU and returns
 * the measured TFLOP/s (denominator check for roofline fractions). */
gp_status gp_probe_mfma_f64(gp_ctx *ctx, double *tflops);
/* Same with 1 or 2 waves per SIMD; also returns the shader clock held during the loop (s_memtime / s_memrealtime)
 * and the SIMD cycles per MFMA instruction, so a roofline fraction can be quoted against the ACHIEVABLE rate. */
gp_status gp_probe_mfma_f64_ex(gp_ctx *ctx, int waves_per_simd, double *tflops, double *clock_mhz, double *cycles_per_mfma);
/* device memory helpers for callers without their own allocator (the JNI shim) */
gp_status gp_dev_alloc(gp_ctx *ctx, size_t bytes, void **dptr);
gp_status gp_dev_free(gp_ctx *ctx, void *dptr);
gp_status gp_dev_upload(gp_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
gp_status gp_dev_download(gp_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- hyper-parameter indexing (integer work, bit-exact) ------------------------------------ */
/* GaussianRbfParams.getAtPosition, utils/KernelRequisites.scala:40-46: pos is 1-BASED;
 * pos outside 1..d+2 -> GP_ERANGE (the Scala throws MatchError). */
gp_status gp_hp_get_at_position(const double *theta, int d, int pos, double *out);

/* ---- Gram matrices -------------------------------------------------------------------------- */
enum { GP_LOWER = 0, GP_FULL = 1 };
/* MatrixUtils.buildKernelMatrix(kernel, X)  utils/MatrixUtils.scala:57-70 with
 * GaussianRbfKernel.apply utils/KernelRequisites.scala:66-72.  K is n x n; the diagonal is exactly
 * sf*sf + sn*sn.  uplo = GP_FULL mirrors (exactly symmetric), GP_LOWER leaves the strict upper
 * triangle untouched. */
gp_status gp_gram_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *theta, double *K, int ldk, int uplo);
gp_status gp_gram_rbf_dev(gp_ctx *ctx, const double *dX, int n, int d, int ldx, const double *theta, double *dK, int ldk, int uplo);
/* Derivative Gram matrix dK/dtheta_pos, full symmetric, pos 1-BASED in the vector order sf, l_1..l_d, sn:
 * MatrixUtils.buildMatrixWithFunc(X)(kernel.derAfterHyperParam(pos)), utils/MatrixUtils.scala:72-84 with
 * GaussianRbfKernel.derAfterHyperParam, utils/KernelRequisites.scala:76-86 (2 sf e | sf^2 e (x_k-y_k)^2 l_k^-3 | i==j ? 2 sn : 0).
 * The fused gradient entry points never materialise these; this is the reference's building block itself.  pos outside
 * 1..d+2 -> GP_ERANGE (MatchError). */
gp_status gp_dgram_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *theta, int pos, double *D, int ldd);
/* MatrixUtils.buildKernelMatrix(kernel, X*, X)  utils/MatrixUtils.scala:44-55,86-97 (never adds
 * noise).  Ks is m x n. */
gp_status gp_cross_gram_rbf(gp_ctx *ctx, const double *Xs, int m, int ldxs, const double *X, int n, int ldx, int d, const double *theta, double *Ks, int ldks);

/* ---- dense factorisation / solves ----------------------------------------------------------- */
/* breeze.linalg.cholesky (call sites gp/regression/GpPredictor.scala:120,
 * gp/classification/EpParameterEstimator.scala:58): A (n x n, lower triangle read) is replaced
 * by L with a ZERO strict upper triangle.  Not PD -> GP_ENOTPD and *info = 1-based pivot. */
gp_status gp_potrf_lower(gp_ctx *ctx, double *A, int n, int lda, int *info);
/* MatrixUtils.forwardSolve / backSolve, utils/MatrixUtils.scala:17-35,115-133.
 * trans = 0: solve L X = B (forwardSolve(L, B)); trans = 1: solve L^T X = B (backSolve(L.t, B)).
 * B (n x nrhs) is overwritten by X. */
gp_status gp_trsm_lower(gp_ctx *ctx, int trans, const double *L, int n, int ldl, double *B, int nrhs, int ldb);
/* MatrixUtils.invTriangular(L, isUpper=false), utils/MatrixUtils.scala:106-113 */
gp_status gp_inv_lower(gp_ctx *ctx, const double *L, int n, int ldl, double *Linv, int ldi);

/* ---- GP regression --------------------------------------------------------------------------- */
/* GpPredictor.preComputeComponents, gp/regression/GpPredictor.scala:104-124.
 * sigma_noise = NaN means None; a number is added UN-SQUARED to the diagonal (:116). */
gp_status gp_fit_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *y, const double *theta, double sigma_noise, gp_model **out, int *info);
gp_status gp_fit_rbf_dev(gp_ctx *ctx, const double *dX, int n, int d, int ldx, const double *dy, const double *theta, double sigma_noise, gp_model **out, int *info);
/* Same with a host-built Gram matrix (any KernelFunc, e.g. Co2Kernel gp/regression/Co2Prediction.scala:29).
 * Such a model supports gp_model_get and gp_predict_from_gram (the caller evaluates K* and K** with its own kernel). */
gp_status gp_fit_from_gram(gp_ctx *ctx, const double *K, int n, int ldk, const double *y, gp_model **out, int *info);
/* Re-fit an existing model in place (same n, d): no allocation, fully asynchronous. */
gp_status gp_model_refit_dev(gp_model *model, const double *theta, double sigma_noise);
gp_status gp_model_status(gp_model *model, int *info); /* syncs; GP_ENOTPD if the last (re)fit failed */
enum { GP_GET_L = 0, GP_GET_ALPHA = 1, GP_GET_LML = 2 };
/* afterLearningComponents = (L, alpha, _) GpPredictor.scala:157; LML = GpPredictor.logLikelihood :144-149.
 * GP_GET_L: out is n x n with ld (zero strict upper); GP_GET_ALPHA: n doubles; GP_GET_LML: 1 double. */
gp_status gp_model_get(gp_model *model, int what, double *out, int ld);
void gp_model_destroy(gp_model *model);
/* GpPredictor.predict / computePosterior, gp/regression/GpPredictor.scala:24-58:
 * mean = K* alpha; V = L \ K*^T; Sigma* = Gram(X*) - V^T V whose diagonal carries sf^2 + sn^2.
 * mean[m] required; var_diag[m] optional (diag only); cov (m x m, ldc) optional (full, small m). */
gp_status gp_predict(gp_model *model, const double *Xs, int m, int ldxs, double *mean, double *var_diag, double *cov, int ldc);
/* device-resident variant: dXs (m x d), dmean[m], dvar[m] in HBM; asynchronous. */
gp_status gp_predict_dev(gp_model *model, const double *dXs, int m, int ldxs, double *dmean, double *dvar);
/* GpPredictor.computePosterior(trainingData, testData, l, alphaVec), gp/regression/GpPredictor.scala:45-58, for a factor the
 * CALLER holds (the entry GP-UCB, gp/optimization/GPOptimizer.scala:91, and the GP-UKF,
 * dynamicalsystems/filtering/GPUnscentedKalmanFilter.scala:78-87,141-142, use): mean = K* alpha,
 * V = forwardSolve(L, K*^T), Sigma* = buildKernelMatrix(X*) - V^T V with the ARD-RBF kernel at theta.  L is n x n (lower
 * triangle read), alpha[n].  mean[m] required; var_diag[m], cov (m x m, ldc) and V (n x m, ldv -- the second member of the
 * Scala return tuple) are optional. */
gp_status gp_posterior_from_factor(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *theta, const double *L, int ldl, const double *alpha, const double *Xs, int m, int ldxs, double *mean, double *var_diag, double *cov, int ldc, double *V, int ldv);
/* The same overload with an explicit kernelFunc (GpPredictor.scala:50-58) for ANY KernelFunc (Co2Kernel,
 * gp/regression/Co2Prediction.scala:29): the caller evaluates Ks = buildKernelMatrix(kernelFunc, testData, trainingData)
 * (m x n) and, for the covariance, Kss = buildKernelMatrix(kernelFunc, testData) (m x m) on the host; kss_diag[m] suffices
 * for var_diag alone.  cov requires Kss; var_diag requires Kss or kss_diag. */
gp_status gp_posterior_from_gram(gp_ctx *ctx, const double *Ks, int m, int n, int ldks, const double *Kss, int ldkss, const double *kss_diag, const double *L, int ldl, const double *alpha, double *mean, double *var_diag, double *cov, int ldc, double *V, int ldv);
/* GpPredictor.predict (:24-43) for a model fitted with gp_fit_from_gram: factor and alpha stay resident, only the host-built
 * Ks (m x n) and Kss / kss_diag cross the boundary. */
gp_status gp_predict_from_gram(gp_model *model, const double *Ks, int m, int ldks, const double *Kss, int ldkss, const double *kss_diag, double *mean, double *var_diag, double *cov, int ldc);

/* ---- log marginal likelihood + gradient ------------------------------------------------------ */
/* GpPredictor.logLikelihoodWithDerivatives, gp/regression/GpPredictor.scala:60-80, evaluated at B
 * hyper-parameter settings (thetas is B x P row-major: setting b at thetas + b*P, P = d+2), the
 * batch driven by obtainOptimalHyperParams :126-142 / MeshHyperParamsLogLikelihoodEvaluator.scala:26-40.
 * nparams = optimizedParamsNum (<= P): only the first nparams gradient components.
 * lml[B], grad[B x nparams] row-major.  info[B]: 0 or failing pivot per setting (lml = NaN there). */
gp_status gp_lml_grad_rbf_batched(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *y, const double *thetas, int B, int nparams, double sigma_noise, double *lml, double *grad, int *info);

/* The same method (GpPredictor.scala:60-80) for ANY KernelFunc -- the route SURVEY.md 8(b) prescribes for kernels without a
 * device form ("any other kernel -> host-built Gram through gp_*_from_gram"): the caller evaluates, with the reference's own host
 * loops, K = buildKernelMatrix(kernelFunc, trainingData) (:62; n x n, ld ldk, the kernel's own noise already on its diagonal) and
 * dK[p] = buildMatrixWithFunc(trainingData)(kernelFunc.derAfterHyperParam(p + 1)) for p < nparams = optimizedParamsNum (:74; P
 * pointers to n x n matrices with leading dimension lddk, SYMMETRIC as that builder makes them -- the lower triangles are read);
 * the device factors K (+ sigma_noise on the diagonal, un-squared, :116; NaN = None), forms alpha and K^-1 = L^-T L^-1 (:66-67)
 * and returns *lml and grad[p] = 1/2 tr((alpha alpha^T - K^-1) dK[p]) (:76).  Not PD -> GP_ENOTPD, *info = 1-based pivot. */
gp_status gp_lml_grad_from_gram(gp_ctx *ctx, const double *K, int n, int ldk, const double *y, const double *const *dK, int nparams, int lddk, double sigma_noise, double *lml, double *grad, int *info);

/* GpPredictor.obtainOptimalHyperParams, gp/regression/GpPredictor.scala:126-142, through
 * BreezeLbfgsOptimizer.maximize, optimization/Optimization.scala:30-63 (L-BFGS, m = history = 4, maxIter = 20 there):
 * maximises the LML over the first nparams entries of theta (optimizeNoise=false <=> nparams = d+1), starting at theta0,
 * and returns the best point any evaluation saw (:44-46,52-55) in theta_out[d+2] with its LML.  The training data stay in
 * HBM for the whole run and the trial steps of one line search are evaluated as one lockstep batch.  iters_out / evals_out
 * (optional) report iterations and LML evaluations.  Iterates are not Breeze's (third-party, not in the reference tree). */
gp_status gp_optimize_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *y, const double *theta0, int nparams, double sigma_noise, int max_iter, int history, double *theta_out, double *lml_out, int *iters_out, int *evals_out);

/* ---- EP binary classification ---------------------------------------------------------------- */
/* EpParameterEstimator(kernelMatrix, targets, _), gp/classification/EpParameterEstimator.scala:11-12.
 * K is any ready-made n x n Gram matrix, y in {-1,+1}. */
gp_status gp_ep_create(gp_ctx *ctx, const double *K, int n, int ldk, const int32_t *y, gp_ep **out);
/* nsweeps sweeps of estimateSiteParams :40-62 (site loop + end-of-sweep refactorisation).
 * tau[n], nu[n] receive the site parameters after the last sweep (either may be NULL). */
gp_status gp_ep_sweep(gp_ep *ep, int nsweeps, double *tau, double *nu, int *info);
/* Load site parameters obtained earlier (GpClassifier.classify with learnParams = Some(..), GpClassifier.scala:26-28):
 * rebuilds L, Sigma and mu from them with one end-of-sweep refactorisation (EpParameterEstimator.scala:56-61). */
gp_status gp_ep_set_site_params(gp_ep *ep, const double *tau, const double *nu, int *info);
/* epMarginalLikelihood :71-96.  strict != 0: as compiled (term at :92 dropped); 0: intended formula. */
gp_status gp_ep_lml(gp_ep *ep, int strict, double *lml);
/* MarginalLikelihoodEvaluator.logLikelihoodDerivativesAfterHyperParams, gp/classification/MarginalLikelihoodEvaluator.scala:46-66:
 * gradient of the EP log marginal likelihood w.r.t. the d+2 ARD-RBF hyper-parameters, for the K this state was created
 * from (K = Gram(X, theta)).  strict != 0: as compiled (rMatrix = b b^T only, the `- backSolve(..)` line at :59 is a dropped
 * statement; b = nu - (S^1/2 L) \ (L^T \ (S^1/2 K nu))); strict == 0: Rasmussen & Williams Alg. 5.2. grad[d+2]. */
gp_status gp_ep_lml_grad_rbf(gp_ep *ep, const double *X, int d, int ldx, const double *theta, int strict, double *grad);
enum { GP_EP_GET_L = 0, GP_EP_GET_SIGMA = 1, GP_EP_GET_MU = 2, GP_EP_GET_CAV_TAU = 3, GP_EP_GET_CAV_NU = 4 };
gp_status gp_ep_get(gp_ep *ep, int what, double *out, int ld);
/* GpClassifier.classify, gp/classification/GpClassifier.scala:24-47: Ks is m x n (test-train),
 * kss_diag[m] the diagonal of the test Gram matrix; prob[m] = Phi(mu*_i / sqrt(1 + var*_i)). */
gp_status gp_ep_predict(gp_ep *ep, const double *Ks, int m, int ldks, const double *kss_diag, double *prob);
/* EP log marginal likelihood over B ARD-RBF hyper-parameter settings (thetas B x (d+2) row-major), by setting index:
 * MeshHyperParamsLogLikelihoodEvaluator.recEvaluate, gp/classification/MeshHyperParamsLogLikelihoodEvaluator.scala:26-40, calling
 * MarginalLikelihoodEvaluator.logLikelihoodWithoutGrad, gp/classification/MarginalLikelihoodEvaluator.scala:24-31, per setting
 * (Gram -> EpParameterEstimator.estimateSiteParams -> epMarginalLikelihood).  Sweeps run until
 * AvgBasedStopCriterion(stop_eps) holds (EpParameterEstimator.scala:187-202, checked before every sweep but the first) or
 * max_sweeps is reached; stop_eps < 0 runs exactly max_sweeps.  lml[B]; sweeps[B] and info[B] optional (info: 0, or the
 * failing pivot of I + S^1/2 K S^1/2 with lml = NaN).  The reference's mis-keyed result map is not replicated. */
gp_status gp_ep_lml_rbf_batched(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *thetas, int B, double stop_eps, int max_sweeps, int strict, double *lml, int *sweeps, int *info);
/* MarginalLikelihoodEvaluator.logLikelihood (gp/classification/MarginalLikelihoodEvaluator.scala:33-44) at B settings: the EP
 * log marginal likelihood of gp_ep_lml_rbf_batched AND its gradient w.r.t. all d+2 hyper-parameters (:46-66; strict as in
 * gp_ep_lml_grad_rbf), grad B x (d+2) row-major (NaN where info[b] != 0). */
gp_status gp_ep_lml_grad_rbf_batched(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *thetas, int B, double stop_eps, int max_sweeps, int strict, double *lml, double *grad, int *sweeps, int *info);
/* GradientHyperParamsOptimizer.optimizeHyperParams, gp/classification/HyperParamsOptimization.scala:31-55, with the
 * BreezeLbfgsOptimizer it is wired with (optimization/Optimization.scala:30-63: L-BFGS, m = history, maxIter = max_iter, best-seen
 * point): maximises the EP log marginal likelihood over all d+2 hyper-parameters starting at theta0.  Every objective evaluation is
 * a full EP run (Gram, sweeps until AvgBasedStopCriterion(stop_eps) -- stop_eps < 0: exactly max_sweeps --, LML and gradient);
 * the trial steps of a line search run concurrently.  theta_out[d+2], *lml_out = EP LML there.  Iterates are not Breeze's. */
gp_status gp_ep_optimize_rbf(gp_ctx *ctx, const double *X, int n, int d, int ldx, const int32_t *y, const double *theta0, double stop_eps, int max_sweeps, int strict, int max_iter, int history, double *theta_out, double *lml_out, int *iters_out, int *evals_out);
void gp_ep_destroy(gp_ep *ep);

/* ---- a composite user kernel on the device: Co2Kernel --------------------------------------------- */
/* gp/regression/Co2Prediction.scala:29-137: k = k1 (squared exponential) + k2 (decaying periodic) + k3 (rational quadratic) +
 * k4 (squared exponential + hp11^2 on the diagonal), 11 hyper-parameters hp1..hp11 in the order of Co2HyperParams (1-based
 * getAtPosition :24), ONE-dimensional inputs x[n].  Same contracts as the *_rbf entry points. */
gp_status gp_gram_co2(gp_ctx *ctx, const double *x, int n, const double *theta, double *K, int ldk, int uplo);
/* buildMatrixWithFunc(x)(Co2Kernel.derAfterHyperParam(pos)), :69-137; pos 1..11, else GP_ERANGE (MatchError) */
gp_status gp_dgram_co2(gp_ctx *ctx, const double *x, int n, const double *theta, int pos, double *D, int ldd);
gp_status gp_cross_gram_co2(gp_ctx *ctx, const double *xs, int m, const double *x, int n, const double *theta, double *Ks, int ldks);
/* GpPredictor(co2Kernel).preComputeComponents: the returned model works with gp_model_get, gp_predict (Xs = xs, ldxs >= m, the
 * kernel rebuilt on the device) and gp_model_refit_dev (theta = 11 values). */
gp_status gp_fit_co2(gp_ctx *ctx, const double *x, int n, const double *y, const double *theta, double sigma_noise, gp_model **out, int *info);
/* GpPredictor.logLikelihoodWithDerivatives (:60-80) with Co2Kernel at B settings (thetas B x 11 row-major); grad B x nparams. */
gp_status gp_lml_grad_co2_batched(gp_ctx *ctx, const double *x, int n, const double *y, const double *thetas, int B, int nparams, double sigma_noise, double *lml, double *grad, int *info);
/* GpPredictor.obtainOptimalHyperParams (:126-142) with Co2Kernel: L-BFGS (m = history, maxIter = max_iter, best-seen point). */
gp_status gp_optimize_co2(gp_ctx *ctx, const double *x, int n, const double *y, const double *theta0, int nparams, double sigma_noise, int max_iter, int history, double *theta_out, double *lml_out, int *iters_out, int *evals_out);

/* ---- batched small-n posteriors: GP-UCB and GP-UKF --------------------------------------------- */
/* The heaviest callers of GpPredictor.computePosterior (gp/regression/GpPredictor.scala:45-58) ask for ONE test point per call
 * against a few hundred training points: GPOptimizer.maximizeUCB (gp/optimization/GPOptimizer.scala:82-109) inside L-BFGS, and
 * the GP-UKF's transition / observation / noise functions (dynamicalsystems/filtering/GPUnscentedKalmanFilter.scala:72-93,138-147)
 * once per sigma point, state dimension and time step.  A gp_small holds G models that share their training inputs X (one GP
 * per output dimension, each with its own theta, L, L^-1 and alpha) so that all (model, test point) pairs of a call are ONE
 * launch.  n and capacity are at most GP_SMALL_MAX_N. */
#define GP_SMALL_MAX_N 2048
/* GPUnscentedKalmanFilter.learnInputOutput (:116-129) / GPOptimizer.maximize's preComputeComponents (:51): fit G models on the
 * shared inputs X (n x d); Y is n x G (column g = targets of model g), thetas G x (d+2) row-major.  `capacity` >= n reserves
 * room for gp_small_append. */
gp_status gp_small_fit(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *Y, int ldy, int G, const double *thetas, double sigma_noise, int capacity, gp_small **out, int *info);
/* The same from factors the caller holds: Ls = G matrices (n x n, leading dimension ldl, model g at Ls + g*ldl*n), alphas G x n.
 * Such a batch cannot be appended to (it carries no targets). */
gp_status gp_small_from_factors(gp_ctx *ctx, const double *X, int n, int d, int ldx, const double *thetas, int G, const double *Ls, int ldl, const double *alphas, int capacity, gp_small **out);
void gp_small_destroy(gp_small *small);
gp_status gp_small_size(const gp_small *small, int *n, int *capacity, int *G);
enum { GP_SMALL_GET_L = 0, GP_SMALL_GET_LINV = 1, GP_SMALL_GET_ALPHA = 2 };
gp_status gp_small_get(gp_small *small, int g, int what, double *out, int ld);
/* computePosterior(X, x*_i, L_g, alpha_g)._1.mean(0) and .sigma(0,0) for every model g and test point i in one launch:
 * mean[g*m + i], var[g*m + i] (var = sf^2 + sn^2 - |L^-1 k*|^2, the diagonal the symmetric builder gives, :56). */
gp_status gp_small_posterior(gp_small *small, const double *Xs, int m, int ldxs, double *mean, double *var);
/* GPOptimizer.maximizeUCB's objective (:88-106) for model g at m candidate points: value[i] = mean + kappa*sqrt(var) and its
 * gradient grad[i*d + k] w.r.t. the candidate, through GaussianRbfKernel.gradient (utils/KernelRequisites.scala:95-107). */
gp_status gp_small_ucb(gp_small *small, int g, const double *Xs, int m, int ldxs, double kappa, double *value, double *grad);
/* GPOptimizer.maximize (:47-72) appends the chosen point and REFITS (O(n^3)); this extends L, L^-1 and alpha of every model by
 * the new row in O(n^2): x_new[d], y_new[G].  GP_ENOTPD (*info = n+1) leaves the batch unchanged. */
gp_status gp_small_append(gp_small *small, const double *x_new, const double *y_new, int *info);
/* The c L-BFGS runs of one GP-UCB iteration (GPOptimizer.scala:55-63; BreezeLbfgsOptimizer m = history, maxIter = max_iter,
 * optimization/Optimization.scala:30-63) in lockstep: starts is c x d (leading dimension lds), all trial points of an iteration
 * are evaluated in one launch.  best_x[d], *best_val: the best point and UCB value any evaluation saw. */
gp_status gp_small_maximize_ucb(gp_small *small, int g, const double *starts, int c, int lds, double kappa, int max_iter, int history, double *best_x, double *best_val, int *evals_out);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ------------------------------------------- */
/* The reference has no distributed code; these drive its two embarrassingly parallel loops: the settings that
 * GpPredictor.obtainOptimalHyperParams (gp/regression/GpPredictor.scala:126-142) / MeshHyperParamsLogLikelihoodEvaluator
 * (gp/classification/MeshHyperParamsLogLikelihoodEvaluator.scala:26-40) visit one by one, and the rows of testData in
 * GpPredictor.predict (:24-43).  Units are sharded contiguously, unit u -> rank u / ceil(U / world); every rank passes the SAME
 * arguments and receives the complete result; the only collective is ONE all-gather of the per-rank results per call, preceded
 * by an all-gather of one status word per rank: a rank whose local part fails (GP_ENOMEM, GP_EHIP, ...) still takes part in
 * the status exchange, then EVERY rank returns without entering the result collective -- the failing rank with its own status,
 * its peers with GP_EPEER -- so one rank's failure never leaves the group blocked.  Argument checks that every rank sees
 * alike (GP_EINVAL) return before any collective.  The Cholesky itself stays single-GPU.  RCCL is loaded at run time;
 * without it these return GP_ERCCL.
 * Rendezvous: rank 0 obtains an id and hands its GP_DIST_ID_BYTES bytes to the other ranks over whatever channel the host
 * program has (a file, a socket, torch.distributed); then every rank calls gp_dist_init on the context of ITS device. */
#define GP_DIST_ID_BYTES 128
gp_status gp_dist_unique_id(gp_ctx *ctx, unsigned char *id /* [GP_DIST_ID_BYTES] */);
gp_status gp_dist_init(gp_ctx *ctx, const unsigned char *id, int rank, int world, gp_dist **out);
void gp_dist_destroy(gp_dist *dist);
/* The agreement rule of the status exchange, host only (no device, no communicator): status[world] = every rank's gp_status as
 * a double.  All zero -> GP_OK; status[rank] != 0 -> that status; otherwise GP_EPEER.  *bad_rank (may be NULL) = the first failing
 * rank (this rank if it failed itself), -1 if none.  Exported so that a host with its own transport (MPI, sockets) can apply the same rule. */
gp_status gp_dist_status_scan(const double *status, int world, int rank, int *bad_rank);
/* Test hook: the NEXT gp_dist_lml_grad_batched / gp_dist_predict on this communicator fails locally with GP_ENOMEM after
 * its evaluation (exercises the status exchange; tests/test_gpu_dist.py). */
gp_status gp_dist_inject_failure(gp_dist *dist);
/* [lo, hi) of `total` units owned by this rank */
gp_status gp_dist_shard(const gp_dist *dist, int total, int *lo, int *hi);
/* gp_lml_grad_rbf_batched over B settings sharded B/world per GPU (BASELINE config C3); lml[B], grad[B x nparams], info[B]
 * complete on every rank. */
gp_status gp_dist_lml_grad_batched(gp_dist *dist, const double *X, int n, int d, int ldx, const double *y, const double *thetas, int B, int nparams, double sigma_noise, double *lml, double *grad, int *info);
/* gp_predict (mean + diagonal variance) of ONE request of m test points sharded m/world per GPU (BASELINE config C5); `model`
 * is this rank's own fitted model (every rank fits redundantly: cheaper than shipping L); mean[m], var[m] complete on every rank. */
gp_status gp_dist_predict(gp_dist *dist, gp_model *model, const double *Xs, int m, int ldxs, double *mean, double *var);

#ifdef __cplusplus
}
#endif
#endif /* GPCORE_H */
