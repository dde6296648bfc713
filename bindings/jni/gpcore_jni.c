/*
 * gpcore_jni.c -- JNI glue between the Scala shim (bindings/scala/gpcore/Native.scala) and libgpcore.so.
 * NOT compiled in the build container (no JDK, no jni.h: SURVEY.md section 8c); it is the binding a
 * maintainer adds on a box with a JDK:
 *     gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *         bindings/jni/gpcore_jni.c -Lgp_algos_amd -lgpcore -o libgpcore_jni.so
 * Every function is a 1:1 forward to include/gpcore.h; all logic stays behind the C-ABI, which IS tested.
 * Breeze DenseMatrix(data, offset, majorStride) crosses as (double[] data, int offset, int ld).
 * GetPrimitiveArrayCritical is held only around the library's synchronous host<->device copies.
 */
#include <jni.h>
#include <stdint.h>
#include "gpcore.h"

static void throw_for(JNIEnv *env, gp_ctx *ctx, gp_status st, int info) {
    const char *cls = "java/lang/RuntimeException";
    if (st == GP_EINVAL) cls = "java/lang/IllegalArgumentException";          /* require / assert            */
    else if (st == GP_ENOTPD) cls = "breeze/linalg/NotConvergedException";    /* what breeze.linalg.cholesky throws */
    else if (st == GP_ERANGE) cls = "scala/MatchError";                       /* getAtPosition past the end  */
    else if (st == GP_ENOMEM) cls = "java/lang/OutOfMemoryError";
    (void)info;
    jclass c = (*env)->FindClass(env, cls);
    if (!c) c = (*env)->FindClass(env, "java/lang/RuntimeException");
    (*env)->ThrowNew(env, c, ctx ? gp_last_error(ctx) : "gpcore error");
}

#define CTX(h) ((gp_ctx *)(intptr_t)(h))
#define MODEL(h) ((gp_model *)(intptr_t)(h))
#define EP(h) ((gp_ep *)(intptr_t)(h))

JNIEXPORT jlong JNICALL Java_gpcore_Native_ctxCreate(JNIEnv *env, jclass k, jint device) {
    gp_ctx *ctx = NULL;
    gp_status st = gp_ctx_create(device, NULL, &ctx);
    if (st != GP_OK) { throw_for(env, NULL, st, 0); return 0; }
    return (jlong)(intptr_t)ctx;
}
JNIEXPORT void JNICALL Java_gpcore_Native_ctxDestroy(JNIEnv *env, jclass k, jlong h) { gp_ctx_destroy(CTX(h)); }

/* MatrixUtils.buildKernelMatrix(kernel, X) -> K (n x n, column-major) */
JNIEXPORT void JNICALL Java_gpcore_Native_gramRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                  jint ldx, jdoubleArray theta, jdoubleArray out) {
    double *X = (*env)->GetPrimitiveArrayCritical(env, x, NULL);
    double *T = (*env)->GetPrimitiveArrayCritical(env, theta, NULL);
    double *K = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    gp_status st = gp_gram_rbf(CTX(h), X + xoff, n, d, ldx, T, K, n, GP_FULL);
    (*env)->ReleasePrimitiveArrayCritical(env, out, K, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, theta, T, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, x, X, JNI_ABORT);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}

/* MatrixUtils.buildMatrixWithFunc(X)(kernel.derAfterHyperParam(pos)) -> dK/dtheta_pos (n x n, column-major), pos 1-based */
JNIEXPORT void JNICALL Java_gpcore_Native_dgramRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                   jint ldx, jdoubleArray theta, jint pos, jdoubleArray out) {
    double *X = (*env)->GetPrimitiveArrayCritical(env, x, NULL);
    double *T = (*env)->GetPrimitiveArrayCritical(env, theta, NULL);
    double *D = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    gp_status st = gp_dgram_rbf(CTX(h), X + xoff, n, d, ldx, T, pos, D, n);
    (*env)->ReleasePrimitiveArrayCritical(env, out, D, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, theta, T, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, x, X, JNI_ABORT);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);   /* GP_ERANGE -> MatchError */
}

/* releases the context's cached device workspaces (call between phases of a long-lived JVM) */
JNIEXPORT void JNICALL Java_gpcore_Native_ctxTrim(JNIEnv *env, jclass k, jlong h) {
    gp_status st = gp_ctx_trim(CTX(h));
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}

/* GpPredictor.preComputeComponents -> model handle (L, alpha, LML stay in HBM) */
JNIEXPORT jlong JNICALL Java_gpcore_Native_fitRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                  jint ldx, jdoubleArray y, jdoubleArray theta, jdouble sigmaNoiseOrNaN) {
    double *X = (*env)->GetPrimitiveArrayCritical(env, x, NULL);
    double *Y = (*env)->GetPrimitiveArrayCritical(env, y, NULL);
    double *T = (*env)->GetPrimitiveArrayCritical(env, theta, NULL);
    gp_model *m = NULL;
    int info = 0;
    gp_status st = gp_fit_rbf(CTX(h), X + xoff, n, d, ldx, Y, T, sigmaNoiseOrNaN, &m, &info);
    (*env)->ReleasePrimitiveArrayCritical(env, theta, T, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, y, Y, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, x, X, JNI_ABORT);
    if (st != GP_OK) { throw_for(env, CTX(h), st, info); return 0; }
    return (jlong)(intptr_t)m;
}
JNIEXPORT void JNICALL Java_gpcore_Native_modelGet(JNIEnv *env, jclass k, jlong h, jlong m, jint what, jdoubleArray out, jint ld) {
    double *O = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    gp_status st = gp_model_get(MODEL(m), what, O, ld);
    (*env)->ReleasePrimitiveArrayCritical(env, out, O, 0);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}
JNIEXPORT void JNICALL Java_gpcore_Native_modelDestroy(JNIEnv *env, jclass k, jlong m) { gp_model_destroy(MODEL(m)); }

/* GpPredictor.predict: mean[m], var[m] (may be null), cov[m*m] (may be null) */
JNIEXPORT void JNICALL Java_gpcore_Native_predict(JNIEnv *env, jclass k, jlong h, jlong m, jdoubleArray xs, jint xsoff, jint mm,
                                                  jint ldxs, jdoubleArray mean, jdoubleArray var, jdoubleArray cov) {
    double *XS = (*env)->GetPrimitiveArrayCritical(env, xs, NULL);
    double *ME = (*env)->GetPrimitiveArrayCritical(env, mean, NULL);
    double *VA = var ? (*env)->GetPrimitiveArrayCritical(env, var, NULL) : NULL;
    double *CO = cov ? (*env)->GetPrimitiveArrayCritical(env, cov, NULL) : NULL;
    gp_status st = gp_predict(MODEL(m), XS + xsoff, mm, ldxs, ME, VA, CO, mm);
    if (cov) (*env)->ReleasePrimitiveArrayCritical(env, cov, CO, 0);
    if (var) (*env)->ReleasePrimitiveArrayCritical(env, var, VA, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, mean, ME, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, xs, XS, JNI_ABORT);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}

/* GpPredictor.logLikelihoodWithDerivatives at B settings: out = [lml_0, grad_0[0..np), lml_1, ...] */
JNIEXPORT void JNICALL Java_gpcore_Native_lmlGradBatched(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint n, jint d, jint ldx,
                                                         jdoubleArray y, jdoubleArray thetas, jint B, jint nparams,
                                                         jdouble sigmaNoiseOrNaN, jdoubleArray lml, jdoubleArray grad, jintArray info) {
    double *X = (*env)->GetPrimitiveArrayCritical(env, x, NULL);
    double *Y = (*env)->GetPrimitiveArrayCritical(env, y, NULL);
    double *T = (*env)->GetPrimitiveArrayCritical(env, thetas, NULL);
    double *L = (*env)->GetPrimitiveArrayCritical(env, lml, NULL);
    double *G = (*env)->GetPrimitiveArrayCritical(env, grad, NULL);
    jint *I = (*env)->GetPrimitiveArrayCritical(env, info, NULL);
    gp_status st = gp_lml_grad_rbf_batched(CTX(h), X, n, d, ldx, Y, T, B, nparams, sigmaNoiseOrNaN, L, G, (int *)I);
    (*env)->ReleasePrimitiveArrayCritical(env, info, I, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, grad, G, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, lml, L, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, thetas, T, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, y, Y, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, x, X, JNI_ABORT);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}

/* GpPredictor.obtainOptimalHyperParams: thetaInOut holds theta0 on entry and the best-seen point on return; returns its LML */
JNIEXPORT jdouble JNICALL Java_gpcore_Native_optimizeRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint n, jint d, jint ldx,
                                                         jdoubleArray y, jdoubleArray thetaInOut, jint nparams, jdouble sigmaNoiseOrNaN,
                                                         jint maxIter, jint history) {
    double *X = (*env)->GetPrimitiveArrayCritical(env, x, NULL);
    double *Y = (*env)->GetPrimitiveArrayCritical(env, y, NULL);
    double *T = (*env)->GetPrimitiveArrayCritical(env, thetaInOut, NULL);
    double theta0[66], lml = 0.0;
    for (int i = 0; i < d + 2 && i < 66; ++i) theta0[i] = T[i];
    gp_status st = gp_optimize_rbf(CTX(h), X, n, d, ldx, Y, theta0, nparams, sigmaNoiseOrNaN, maxIter, history, T, &lml, NULL, NULL);
    (*env)->ReleasePrimitiveArrayCritical(env, thetaInOut, T, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, y, Y, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, x, X, JNI_ABORT);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
    return lml;
}

/* MeshHyperParamsLogLikelihoodEvaluator over the leaves of the grid: EP LML per setting, by setting index */
JNIEXPORT void JNICALL Java_gpcore_Native_epLmlRbfBatched(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint n, jint d, jint ldx,
                                                          jintArray y, jdoubleArray thetas, jint B, jdouble stopEps, jint maxSweeps,
                                                          jboolean strict, jdoubleArray lml, jintArray sweeps, jintArray info) {
    double *X = (*env)->GetPrimitiveArrayCritical(env, x, NULL);
    jint *Y = (*env)->GetPrimitiveArrayCritical(env, y, NULL);
    double *T = (*env)->GetPrimitiveArrayCritical(env, thetas, NULL);
    double *L = (*env)->GetPrimitiveArrayCritical(env, lml, NULL);
    jint *S = (*env)->GetPrimitiveArrayCritical(env, sweeps, NULL);
    jint *I = (*env)->GetPrimitiveArrayCritical(env, info, NULL);
    gp_status st = gp_ep_lml_rbf_batched(CTX(h), X, n, d, ldx, (const int32_t *)Y, T, B, stopEps, maxSweeps, strict ? 1 : 0, L, (int *)S, (int *)I);
    (*env)->ReleasePrimitiveArrayCritical(env, info, I, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, sweeps, S, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, lml, L, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, thetas, T, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, y, Y, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, x, X, JNI_ABORT);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}

/* breeze.linalg.cholesky / MatrixUtils.forwardSolve, backSolve, invTriangular */
JNIEXPORT void JNICALL Java_gpcore_Native_potrfLower(JNIEnv *env, jclass k, jlong h, jdoubleArray a, jint off, jint n, jint lda) {
    double *A = (*env)->GetPrimitiveArrayCritical(env, a, NULL);
    int info = 0;
    gp_status st = gp_potrf_lower(CTX(h), A + off, n, lda, &info);
    (*env)->ReleasePrimitiveArrayCritical(env, a, A, 0);
    if (st != GP_OK) throw_for(env, CTX(h), st, info);
}
JNIEXPORT void JNICALL Java_gpcore_Native_trsmLower(JNIEnv *env, jclass k, jlong h, jint trans, jdoubleArray l, jint loff, jint n,
                                                    jint ldl, jdoubleArray b, jint boff, jint nrhs, jint ldb) {
    double *L = (*env)->GetPrimitiveArrayCritical(env, l, NULL);
    double *B = (*env)->GetPrimitiveArrayCritical(env, b, NULL);
    gp_status st = gp_trsm_lower(CTX(h), trans, L + loff, n, ldl, B + boff, nrhs, ldb);
    (*env)->ReleasePrimitiveArrayCritical(env, b, B, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, l, L, JNI_ABORT);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}

/* EpParameterEstimator / GpClassifier */
JNIEXPORT jlong JNICALL Java_gpcore_Native_epCreate(JNIEnv *env, jclass k, jlong h, jdoubleArray km, jint off, jint n, jint ldk,
                                                    jintArray targets) {
    double *K = (*env)->GetPrimitiveArrayCritical(env, km, NULL);
    jint *Y = (*env)->GetPrimitiveArrayCritical(env, targets, NULL);
    gp_ep *ep = NULL;
    gp_status st = gp_ep_create(CTX(h), K + off, n, ldk, (const int32_t *)Y, &ep);
    (*env)->ReleasePrimitiveArrayCritical(env, targets, Y, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, km, K, JNI_ABORT);
    if (st != GP_OK) { throw_for(env, CTX(h), st, 0); return 0; }
    return (jlong)(intptr_t)ep;
}
JNIEXPORT void JNICALL Java_gpcore_Native_epSweep(JNIEnv *env, jclass k, jlong h, jlong e, jint nsweeps, jdoubleArray tau, jdoubleArray nu) {
    double *T = (*env)->GetPrimitiveArrayCritical(env, tau, NULL);
    double *N = (*env)->GetPrimitiveArrayCritical(env, nu, NULL);
    int info = 0;
    gp_status st = gp_ep_sweep(EP(e), nsweeps, T, N, &info);
    (*env)->ReleasePrimitiveArrayCritical(env, nu, N, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, tau, T, 0);
    if (st != GP_OK) throw_for(env, CTX(h), st, info);
}
JNIEXPORT jdouble JNICALL Java_gpcore_Native_epLml(JNIEnv *env, jclass k, jlong h, jlong e, jboolean strict) {
    double v = 0.0;
    gp_status st = gp_ep_lml(EP(e), strict ? 1 : 0, &v);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
    return v;
}
JNIEXPORT void JNICALL Java_gpcore_Native_epGet(JNIEnv *env, jclass k, jlong h, jlong e, jint what, jdoubleArray out, jint ld) {
    double *O = (*env)->GetPrimitiveArrayCritical(env, out, NULL);
    gp_status st = gp_ep_get(EP(e), what, O, ld);
    (*env)->ReleasePrimitiveArrayCritical(env, out, O, 0);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}
JNIEXPORT void JNICALL Java_gpcore_Native_epPredict(JNIEnv *env, jclass k, jlong h, jlong e, jdoubleArray ks, jint off, jint m, jint ldks,
                                                    jdoubleArray kssDiag, jdoubleArray prob) {
    double *KS = (*env)->GetPrimitiveArrayCritical(env, ks, NULL);
    double *KD = (*env)->GetPrimitiveArrayCritical(env, kssDiag, NULL);
    double *P = (*env)->GetPrimitiveArrayCritical(env, prob, NULL);
    gp_status st = gp_ep_predict(EP(e), KS + off, m, ldks, KD, P);
    (*env)->ReleasePrimitiveArrayCritical(env, prob, P, 0);
    (*env)->ReleasePrimitiveArrayCritical(env, kssDiag, KD, JNI_ABORT);
    (*env)->ReleasePrimitiveArrayCritical(env, ks, KS, JNI_ABORT);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}
JNIEXPORT void JNICALL Java_gpcore_Native_epDestroy(JNIEnv *env, jclass k, jlong e) { gp_ep_destroy(EP(e)); }
