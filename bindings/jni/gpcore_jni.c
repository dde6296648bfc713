/*
 * gpcore_jni.c -- JNI glue between the Scala shim (bindings/scala/gpcore/Native.scala) and libgpcore.so.
 * NOT compiled in the build container (no JDK, no jni.h: SURVEY.md section 8c); it is the binding a
 * maintainer adds on a box with a JDK:
 *     gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *         bindings/jni/gpcore_jni.c -Lgp_algos_amd -lgpcore -o libgpcore_jni.so
 * Every function is a 1:1 forward to include/gpcore.h; all logic stays behind the C-ABI, which IS tested.
 * Breeze DenseMatrix(data, offset, majorStride) crosses as (double[] data, int offset, int ld).
 *
 * Array discipline (SURVEY.md 8b "threading"): NO JNI critical region is ever open while the library runs.  Inputs
 * are copied out of the Java arrays with Get<Type>ArrayRegion into malloc'd host buffers BEFORE the gpcore call,
 * outputs are written back with Set<Type>ArrayRegion AFTER it, so kernel launches, stream synchronisation, the
 * worker threads of the batched entry points and the multi-iteration optimiser never run with the garbage
 * collector locked out.  Every allocation and every region copy is checked; on failure the pending Java exception
 * (OutOfMemoryError / ArrayIndexOutOfBoundsException) is left in place and the call returns.
 *
 * Exceptions: GP_EINVAL -> IllegalArgumentException (require/assert in the reference), GP_ERANGE -> scala.MatchError
 * (getAtPosition past the end), GP_ENOMEM -> OutOfMemoryError, GP_ENOTPD -> gpcore.NotPositiveDefiniteException
 * (a RuntimeException subclass with a (String) constructor, declared in Native.scala, whose message carries the
 * 1-based failing pivot); the Scala shim rethrows it as breeze.linalg.NotConvergedException(Iterations) -- what
 * breeze.linalg.cholesky throws -- where scalac type-checks that constructor.  GP_EPEER (gp_dist_*: another rank failed, nothing was
 * exchanged) -> IllegalStateException; GP_EHIP / GP_ERCCL -> RuntimeException.
 *
 * tests/test_jni_glue_cpu.py keeps this file honest without a JDK: gcc -fsyntax-only -Wall -Werror against
 * bindings/jni/check/jni.h (a declaration-only subset of the JNI types and function table, NOT a JDK header), and the exported
 * Java_gpcore_Native_* names against the @native declarations of Native.scala, one to one.
 */
#include <jni.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "gpcore.h"

static void throw_for(JNIEnv *env, gp_ctx *ctx, gp_status st, int info) {
    const char *cls = "java/lang/RuntimeException";
    char msg[600];
    if (st == GP_EINVAL) cls = "java/lang/IllegalArgumentException";
    else if (st == GP_ENOTPD) cls = "gpcore/NotPositiveDefiniteException";
    else if (st == GP_ERANGE) cls = "scala/MatchError";
    else if (st == GP_ENOMEM) cls = "java/lang/OutOfMemoryError";
    else if (st == GP_EPEER) cls = "java/lang/IllegalStateException";   /* another rank of the gp_dist group failed; this JVM is intact */
    snprintf(msg, sizeof msg, "%s%s (gp_status %d, info %d)", st == GP_ENOTPD ? "matrix not positive definite: " : "",
             ctx ? gp_last_error(ctx) : "gpcore error", (int)st, info);
    jclass c = (*env)->FindClass(env, cls);
    if (!c) { (*env)->ExceptionClear(env); c = (*env)->FindClass(env, "java/lang/RuntimeException"); }
    if (c) (*env)->ThrowNew(env, c, msg);
}

#define CTX(h) ((gp_ctx *)(intptr_t)(h))
#define MODEL(h) ((gp_model *)(intptr_t)(h))
#define EP(h) ((gp_ep *)(intptr_t)(h))

/* elements of a column-major rows x cols view with leading dimension ld */
static jsize span(jint rows, jint cols, jint ld) { return (rows <= 0 || cols <= 0) ? 0 : (jsize)ld * (cols - 1) + rows; }

/* malloc'd copy of a[off .. off+count); NULL with a pending Java exception on failure (a NULL array is an error) */
static double *in_d(JNIEnv *env, jdoubleArray a, jsize off, jsize count) {
    if (!a) { throw_for(env, NULL, GP_EINVAL, 0); return NULL; }
    double *p = malloc(sizeof(double) * (size_t)(count > 0 ? count : 1));
    if (!p) { throw_for(env, NULL, GP_ENOMEM, 0); return NULL; }
    if (count > 0) (*env)->GetDoubleArrayRegion(env, a, off, count, p);
    if ((*env)->ExceptionCheck(env)) { free(p); return NULL; }
    return p;
}
static jint *in_i(JNIEnv *env, jintArray a, jsize count) {
    if (!a) { throw_for(env, NULL, GP_EINVAL, 0); return NULL; }
    jint *p = malloc(sizeof(jint) * (size_t)(count > 0 ? count : 1));
    if (!p) { throw_for(env, NULL, GP_ENOMEM, 0); return NULL; }
    if (count > 0) (*env)->GetIntArrayRegion(env, a, 0, count, p);
    if ((*env)->ExceptionCheck(env)) { free(p); return NULL; }
    return p;
}
/* zero-filled output staging buffer */
static double *out_d(JNIEnv *env, jsize count) {
    double *p = calloc((size_t)(count > 0 ? count : 1), sizeof(double));
    if (!p) throw_for(env, NULL, GP_ENOMEM, 0);
    return p;
}
static void put_d(JNIEnv *env, jdoubleArray a, jsize off, const double *p, jsize count) {
    if (a && p && count > 0 && !(*env)->ExceptionCheck(env)) (*env)->SetDoubleArrayRegion(env, a, off, count, p);
}
static void put_i(JNIEnv *env, jintArray a, const jint *p, jsize count) {
    if (a && p && count > 0 && !(*env)->ExceptionCheck(env)) (*env)->SetIntArrayRegion(env, a, 0, count, p);
}

JNIEXPORT jlong JNICALL Java_gpcore_Native_ctxCreate(JNIEnv *env, jclass k, jint device) {
    gp_ctx *ctx = NULL;
    gp_status st = gp_ctx_create(device, NULL, &ctx);
    if (st != GP_OK) { throw_for(env, NULL, st, 0); return 0; }
    return (jlong)(intptr_t)ctx;
}
JNIEXPORT void JNICALL Java_gpcore_Native_ctxDestroy(JNIEnv *env, jclass k, jlong h) { gp_ctx_destroy(CTX(h)); }
/* releases the context's cached device workspaces (call between phases of a long-lived JVM) */
JNIEXPORT void JNICALL Java_gpcore_Native_ctxTrim(JNIEnv *env, jclass k, jlong h) {
    gp_status st = gp_ctx_trim(CTX(h));
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
}

/* MatrixUtils.buildKernelMatrix(kernel, X) -> K (n x n, column-major) */
JNIEXPORT void JNICALL Java_gpcore_Native_gramRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                  jint ldx, jdoubleArray theta, jdoubleArray out) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *T = X ? in_d(env, theta, 0, d + 2) : NULL;
    double *K = T ? out_d(env, (jsize)n * n) : NULL;
    if (K) {
        gp_status st = gp_gram_rbf(CTX(h), X, n, d, ldx, T, K, n, GP_FULL);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, K, (jsize)n * n);
    }
    free(K); free(T); free(X);
}

/* MatrixUtils.buildMatrixWithFunc(X)(kernel.derAfterHyperParam(pos)) -> dK/dtheta_pos (n x n), pos 1-based */
JNIEXPORT void JNICALL Java_gpcore_Native_dgramRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                   jint ldx, jdoubleArray theta, jint pos, jdoubleArray out) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *T = X ? in_d(env, theta, 0, d + 2) : NULL;
    double *D = T ? out_d(env, (jsize)n * n) : NULL;
    if (D) {
        gp_status st = gp_dgram_rbf(CTX(h), X, n, d, ldx, T, pos, D, n);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, D, (jsize)n * n);   /* GP_ERANGE -> MatchError */
    }
    free(D); free(T); free(X);
}

/* MatrixUtils.buildKernelMatrix(kernel, X*, X) -> K* (m x n) */
JNIEXPORT void JNICALL Java_gpcore_Native_crossGramRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray xs, jint xsoff, jint m, jint ldxs,
                                                       jdoubleArray x, jint xoff, jint n, jint ldx, jint d, jdoubleArray theta,
                                                       jdoubleArray out) {
    double *XS = in_d(env, xs, xsoff, span(m, d, ldxs)), *X = XS ? in_d(env, x, xoff, span(n, d, ldx)) : NULL;
    double *T = X ? in_d(env, theta, 0, d + 2) : NULL, *K = T ? out_d(env, (jsize)m * n) : NULL;
    if (K) {
        gp_status st = gp_cross_gram_rbf(CTX(h), XS, m, ldxs, X, n, ldx, d, T, K, m);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, K, (jsize)m * n);
    }
    free(K); free(T); free(X); free(XS);
}

/* GpPredictor.preComputeComponents -> model handle (L, alpha, LML stay in HBM) */
JNIEXPORT jlong JNICALL Java_gpcore_Native_fitRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                  jint ldx, jdoubleArray y, jdoubleArray theta, jdouble sigmaNoiseOrNaN) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *Y = X ? in_d(env, y, 0, n) : NULL, *T = Y ? in_d(env, theta, 0, d + 2) : NULL;
    gp_model *m = NULL;
    if (T) {
        int info = 0;
        gp_status st = gp_fit_rbf(CTX(h), X, n, d, ldx, Y, T, sigmaNoiseOrNaN, &m, &info);
        if (st != GP_OK) { throw_for(env, CTX(h), st, info); m = NULL; }
    }
    free(T); free(Y); free(X);
    return (jlong)(intptr_t)m;
}
/* the same for a host-built Gram matrix (any KernelFunc) */
JNIEXPORT jlong JNICALL Java_gpcore_Native_fitFromGram(JNIEnv *env, jclass k, jlong h, jdoubleArray km, jint koff, jint n, jint ldk,
                                                       jdoubleArray y) {
    double *K = in_d(env, km, koff, span(n, n, ldk)), *Y = K ? in_d(env, y, 0, n) : NULL;
    gp_model *m = NULL;
    if (Y) {
        int info = 0;
        gp_status st = gp_fit_from_gram(CTX(h), K, n, ldk, Y, &m, &info);
        if (st != GP_OK) { throw_for(env, CTX(h), st, info); m = NULL; }
    }
    free(Y); free(K);
    return (jlong)(intptr_t)m;
}
/* what: 0 = L (count = n*n, ld = n), 1 = alpha (n), 2 = LML (1) */
JNIEXPORT void JNICALL Java_gpcore_Native_modelGet(JNIEnv *env, jclass k, jlong h, jlong m, jint what, jdoubleArray out, jint ld) {
    const jsize count = out ? (*env)->GetArrayLength(env, out) : 0;
    double *O = out_d(env, count);
    if (O) {
        gp_status st = gp_model_get(MODEL(m), what, O, ld);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, O, count);
    }
    free(O);
}
JNIEXPORT void JNICALL Java_gpcore_Native_modelDestroy(JNIEnv *env, jclass k, jlong m) { gp_model_destroy(MODEL(m)); }

/* GpPredictor.predict: mean[m], var[m] (may be null), cov[m*m] (may be null) */
JNIEXPORT void JNICALL Java_gpcore_Native_predict(JNIEnv *env, jclass k, jlong h, jlong m, jdoubleArray xs, jint xsoff, jint mm,
                                                  jint d, jint ldxs, jdoubleArray mean, jdoubleArray var, jdoubleArray cov) {
    double *XS = in_d(env, xs, xsoff, span(mm, d, ldxs));
    double *ME = XS ? out_d(env, mm) : NULL, *VA = (ME && var) ? out_d(env, mm) : NULL, *CO = (ME && cov) ? out_d(env, (jsize)mm * mm) : NULL;
    if (ME && (!var || VA) && (!cov || CO)) {
        gp_status st = gp_predict(MODEL(m), XS, mm, ldxs, ME, VA, CO, mm);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, mean, 0, ME, mm); put_d(env, var, 0, VA, mm); put_d(env, cov, 0, CO, (jsize)mm * mm); }
    }
    free(CO); free(VA); free(ME); free(XS);
}

/* GpPredictor.computePosterior(trainingData, testData, l, alphaVec) with GaussianRbfKernel: mean[m], cov[m*m] or null, v[n*m] or null */
JNIEXPORT void JNICALL Java_gpcore_Native_posteriorFromFactor(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                              jint ldx, jdoubleArray theta, jdoubleArray l, jint loff, jint ldl,
                                                              jdoubleArray alpha, jdoubleArray xs, jint xsoff, jint mm, jint ldxs,
                                                              jdoubleArray mean, jdoubleArray var, jdoubleArray cov, jdoubleArray v) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *T = X ? in_d(env, theta, 0, d + 2) : NULL;
    double *L = T ? in_d(env, l, loff, span(n, n, ldl)) : NULL, *A = L ? in_d(env, alpha, 0, n) : NULL;
    double *XS = A ? in_d(env, xs, xsoff, span(mm, d, ldxs)) : NULL;
    double *ME = XS ? out_d(env, mm) : NULL, *VA = (ME && var) ? out_d(env, mm) : NULL;
    double *CO = (ME && cov) ? out_d(env, (jsize)mm * mm) : NULL, *V = (ME && v) ? out_d(env, (jsize)n * mm) : NULL;
    if (ME && (!var || VA) && (!cov || CO) && (!v || V)) {
        gp_status st = gp_posterior_from_factor(CTX(h), X, n, d, ldx, T, L, ldl, A, XS, mm, ldxs, ME, VA, CO, mm, V, n);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, mean, 0, ME, mm); put_d(env, var, 0, VA, mm); put_d(env, cov, 0, CO, (jsize)mm * mm); put_d(env, v, 0, V, (jsize)n * mm); }
    }
    free(V); free(CO); free(VA); free(ME); free(XS); free(A); free(L); free(T); free(X);
}

/* computePosterior with any other KernelFunc: Ks (m x n) and Kss (m x m) evaluated by the Scala loops */
JNIEXPORT void JNICALL Java_gpcore_Native_posteriorFromGram(JNIEnv *env, jclass k, jlong h, jdoubleArray ks, jint mm, jint n,
                                                            jdoubleArray kss, jdoubleArray l, jint loff, jint ldl, jdoubleArray alpha,
                                                            jdoubleArray mean, jdoubleArray cov, jdoubleArray v) {
    double *KS = in_d(env, ks, 0, (jsize)mm * n), *KSS = KS ? in_d(env, kss, 0, (jsize)mm * mm) : NULL;
    double *L = KSS ? in_d(env, l, loff, span(n, n, ldl)) : NULL, *A = L ? in_d(env, alpha, 0, n) : NULL;
    double *ME = A ? out_d(env, mm) : NULL, *CO = (ME && cov) ? out_d(env, (jsize)mm * mm) : NULL, *V = (ME && v) ? out_d(env, (jsize)n * mm) : NULL;
    if (ME && (!cov || CO) && (!v || V)) {
        gp_status st = gp_posterior_from_gram(CTX(h), KS, mm, n, mm, KSS, mm, NULL, L, ldl, A, ME, NULL, CO, mm, V, n);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, mean, 0, ME, mm); put_d(env, cov, 0, CO, (jsize)mm * mm); put_d(env, v, 0, V, (jsize)n * mm); }
    }
    free(V); free(CO); free(ME); free(A); free(L); free(KSS); free(KS);
}

/* GpPredictor.predict on a model fitted with fitFromGram */
JNIEXPORT void JNICALL Java_gpcore_Native_predictFromGram(JNIEnv *env, jclass k, jlong h, jlong m, jdoubleArray ks, jint mm, jint n,
                                                          jdoubleArray kss, jdoubleArray mean, jdoubleArray cov) {
    double *KS = in_d(env, ks, 0, (jsize)mm * n), *KSS = KS ? in_d(env, kss, 0, (jsize)mm * mm) : NULL;
    double *ME = KSS ? out_d(env, mm) : NULL, *CO = ME ? out_d(env, (jsize)mm * mm) : NULL;
    if (CO) {
        gp_status st = gp_predict_from_gram(MODEL(m), KS, mm, mm, KSS, mm, NULL, ME, NULL, CO, mm);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, mean, 0, ME, mm); put_d(env, cov, 0, CO, (jsize)mm * mm); }
    }
    free(CO); free(ME); free(KSS); free(KS);
}

/* GpPredictor.logLikelihoodWithDerivatives at B settings: lml[B], grad[B*nparams], info[B] */
JNIEXPORT void JNICALL Java_gpcore_Native_lmlGradBatched(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                         jint ldx, jdoubleArray y, jdoubleArray thetas, jint B, jint nparams,
                                                         jdouble sigmaNoiseOrNaN, jdoubleArray lml, jdoubleArray grad, jintArray info) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *Y = X ? in_d(env, y, 0, n) : NULL;
    double *T = Y ? in_d(env, thetas, 0, (jsize)B * (d + 2)) : NULL;
    double *L = T ? out_d(env, B) : NULL, *G = L ? out_d(env, (jsize)B * (nparams > 0 ? nparams : 1)) : NULL;
    jint *I = G ? calloc((size_t)(B > 0 ? B : 1), sizeof(jint)) : NULL;
    if (G && !I) throw_for(env, NULL, GP_ENOMEM, 0);
    if (I) {
        gp_status st = gp_lml_grad_rbf_batched(CTX(h), X, n, d, ldx, Y, T, B, nparams, sigmaNoiseOrNaN, L, G, (int *)I);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, lml, 0, L, B); put_d(env, grad, 0, G, (jsize)B * nparams); put_i(env, info, I, B); }
    }
    free(I); free(G); free(L); free(T); free(Y); free(X);
}

/* GpPredictor.obtainOptimalHyperParams: thetaInOut holds theta0 on entry and the best-seen point on return; returns its LML */
JNIEXPORT jdouble JNICALL Java_gpcore_Native_optimizeRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                         jint ldx, jdoubleArray y, jdoubleArray thetaInOut, jint nparams,
                                                         jdouble sigmaNoiseOrNaN, jint maxIter, jint history) {
    double lml = 0.0;
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *Y = X ? in_d(env, y, 0, n) : NULL;
    double *T0 = Y ? in_d(env, thetaInOut, 0, d + 2) : NULL, *T1 = T0 ? out_d(env, d + 2) : NULL;
    if (T1) {
        gp_status st = gp_optimize_rbf(CTX(h), X, n, d, ldx, Y, T0, nparams, sigmaNoiseOrNaN, maxIter, history, T1, &lml, NULL, NULL);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, thetaInOut, 0, T1, d + 2);
    }
    free(T1); free(T0); free(Y); free(X);
    return lml;
}

/* MeshHyperParamsLogLikelihoodEvaluator over the leaves of the grid: EP LML per setting, by setting index */
JNIEXPORT void JNICALL Java_gpcore_Native_epLmlRbfBatched(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                          jint ldx, jintArray y, jdoubleArray thetas, jint B, jdouble stopEps,
                                                          jint maxSweeps, jboolean strict, jdoubleArray lml, jintArray sweeps,
                                                          jintArray info) {
    double *X = in_d(env, x, xoff, span(n, d, ldx));
    jint *Y = X ? in_i(env, y, n) : NULL;
    double *T = Y ? in_d(env, thetas, 0, (jsize)B * (d + 2)) : NULL, *L = T ? out_d(env, B) : NULL;
    jint *S = L ? calloc((size_t)(B > 0 ? B : 1), sizeof(jint)) : NULL, *I = S ? calloc((size_t)(B > 0 ? B : 1), sizeof(jint)) : NULL;
    if (L && !I) throw_for(env, NULL, GP_ENOMEM, 0);
    if (I) {
        gp_status st = gp_ep_lml_rbf_batched(CTX(h), X, n, d, ldx, (const int32_t *)Y, T, B, stopEps, maxSweeps, strict ? 1 : 0, L,
                                             (int *)S, (int *)I);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, lml, 0, L, B); put_i(env, sweeps, S, B); put_i(env, info, I, B); }
    }
    free(I); free(S); free(L); free(T); free(Y); free(X);
}

/* breeze.linalg.cholesky: a (n x n at off, ld lda) is replaced by L */
JNIEXPORT void JNICALL Java_gpcore_Native_potrfLower(JNIEnv *env, jclass k, jlong h, jdoubleArray a, jint off, jint n, jint lda) {
    const jsize cnt = span(n, n, lda);
    double *A = in_d(env, a, off, cnt);
    if (A) {
        int info = 0;
        gp_status st = gp_potrf_lower(CTX(h), A, n, lda, &info);
        if (st != GP_OK) throw_for(env, CTX(h), st, info); else put_d(env, a, off, A, cnt);
    }
    free(A);
}
/* MatrixUtils.forwardSolve (trans = 0) / backSolve(L.t, .) (trans = 1): b (n x nrhs) is replaced by the solution */
JNIEXPORT void JNICALL Java_gpcore_Native_trsmLower(JNIEnv *env, jclass k, jlong h, jint trans, jdoubleArray l, jint loff, jint n,
                                                    jint ldl, jdoubleArray b, jint boff, jint nrhs, jint ldb) {
    const jsize cnt = span(n, nrhs, ldb);
    double *L = in_d(env, l, loff, span(n, n, ldl)), *B = L ? in_d(env, b, boff, cnt) : NULL;
    if (B) {
        gp_status st = gp_trsm_lower(CTX(h), trans, L, n, ldl, B, nrhs, ldb);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, b, boff, B, cnt);
    }
    free(B); free(L);
}
/* MatrixUtils.invTriangular(L, isUpper = false) */
JNIEXPORT void JNICALL Java_gpcore_Native_invLower(JNIEnv *env, jclass k, jlong h, jdoubleArray l, jint loff, jint n, jint ldl,
                                                   jdoubleArray out) {
    double *L = in_d(env, l, loff, span(n, n, ldl)), *O = L ? out_d(env, (jsize)n * n) : NULL;
    if (O) {
        gp_status st = gp_inv_lower(CTX(h), L, n, ldl, O, n);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, O, (jsize)n * n);
    }
    free(O); free(L);
}

/* EpParameterEstimator / GpClassifier */
JNIEXPORT jlong JNICALL Java_gpcore_Native_epCreate(JNIEnv *env, jclass k, jlong h, jdoubleArray km, jint off, jint n, jint ldk,
                                                    jintArray targets) {
    double *K = in_d(env, km, off, span(n, n, ldk));
    jint *Y = K ? in_i(env, targets, n) : NULL;
    gp_ep *ep = NULL;
    if (Y) {
        gp_status st = gp_ep_create(CTX(h), K, n, ldk, (const int32_t *)Y, &ep);
        if (st != GP_OK) { throw_for(env, CTX(h), st, 0); ep = NULL; }
    }
    free(Y); free(K);
    return (jlong)(intptr_t)ep;
}
JNIEXPORT void JNICALL Java_gpcore_Native_epSweep(JNIEnv *env, jclass k, jlong h, jlong e, jint nsweeps, jint n, jdoubleArray tau,
                                                  jdoubleArray nu) {
    double *T = out_d(env, n), *N = T ? out_d(env, n) : NULL;
    if (N) {
        int info = 0;
        gp_status st = gp_ep_sweep(EP(e), nsweeps, T, N, &info);
        if (st != GP_OK) throw_for(env, CTX(h), st, info); else { put_d(env, tau, 0, T, n); put_d(env, nu, 0, N, n); }
    }
    free(N); free(T);
}
/* learnParams = Some(siteParams, _): load site parameters estimated earlier */
JNIEXPORT void JNICALL Java_gpcore_Native_epSetSiteParams(JNIEnv *env, jclass k, jlong h, jlong e, jint n, jdoubleArray tau,
                                                          jdoubleArray nu) {
    double *T = in_d(env, tau, 0, n), *N = T ? in_d(env, nu, 0, n) : NULL;
    if (N) {
        int info = 0;
        gp_status st = gp_ep_set_site_params(EP(e), T, N, &info);
        if (st != GP_OK) throw_for(env, CTX(h), st, info);
    }
    free(N); free(T);
}
JNIEXPORT jdouble JNICALL Java_gpcore_Native_epLml(JNIEnv *env, jclass k, jlong h, jlong e, jboolean strict) {
    double v = 0.0;
    gp_status st = gp_ep_lml(EP(e), strict ? 1 : 0, &v);
    if (st != GP_OK) throw_for(env, CTX(h), st, 0);
    return v;
}
/* MarginalLikelihoodEvaluator.logLikelihoodDerivativesAfterHyperParams: grad[d+2] */
JNIEXPORT void JNICALL Java_gpcore_Native_epLmlGradRbf(JNIEnv *env, jclass k, jlong h, jlong e, jdoubleArray x, jint xoff, jint n,
                                                       jint d, jint ldx, jdoubleArray theta, jboolean strict, jdoubleArray grad) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *T = X ? in_d(env, theta, 0, d + 2) : NULL, *G = T ? out_d(env, d + 2) : NULL;
    if (G) {
        gp_status st = gp_ep_lml_grad_rbf(EP(e), X, d, ldx, T, strict ? 1 : 0, G);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, grad, 0, G, d + 2);
    }
    free(G); free(T); free(X);
}
/* what: 0 = L, 1 = Sigma (n*n, ld), 2 = mu, 3 = cavity tau, 4 = cavity nu (n) */
JNIEXPORT void JNICALL Java_gpcore_Native_epGet(JNIEnv *env, jclass k, jlong h, jlong e, jint what, jdoubleArray out, jint ld) {
    const jsize count = out ? (*env)->GetArrayLength(env, out) : 0;
    double *O = out_d(env, count);
    if (O) {
        gp_status st = gp_ep_get(EP(e), what, O, ld);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, O, count);
    }
    free(O);
}
JNIEXPORT void JNICALL Java_gpcore_Native_epPredict(JNIEnv *env, jclass k, jlong h, jlong e, jdoubleArray ks, jint off, jint m, jint n,
                                                    jint ldks, jdoubleArray kssDiag, jdoubleArray prob) {
    double *KS = in_d(env, ks, off, span(m, n, ldks)), *KD = KS ? in_d(env, kssDiag, 0, m) : NULL, *P = KD ? out_d(env, m) : NULL;
    if (P) {
        gp_status st = gp_ep_predict(EP(e), KS, m, ldks, KD, P);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, prob, 0, P, m);
    }
    free(P); free(KD); free(KS);
}
JNIEXPORT void JNICALL Java_gpcore_Native_epDestroy(JNIEnv *env, jclass k, jlong e) { gp_ep_destroy(EP(e)); }

/* ---- EP hyper-parameter fitting (HyperParamsOptimization.scala:31-55) ---- */
JNIEXPORT jdouble JNICALL Java_gpcore_Native_epOptimizeRbf(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d, jint ldx,
                                                           jintArray y, jdoubleArray thetaInOut, jdouble stopEps, jint maxSweeps, jboolean strict,
                                                           jint maxIter, jint history) {
    double lml = 0.0;
    double *X = in_d(env, x, xoff, span(n, d, ldx));
    jint *Y = X ? in_i(env, y, n) : NULL;
    double *T0 = Y ? in_d(env, thetaInOut, 0, d + 2) : NULL, *T1 = T0 ? out_d(env, d + 2) : NULL;
    if (T1) {
        gp_status st = gp_ep_optimize_rbf(CTX(h), X, n, d, ldx, (const int32_t *)Y, T0, stopEps, maxSweeps, strict ? 1 : 0, maxIter, history, T1, &lml,
                                          NULL, NULL);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, thetaInOut, 0, T1, d + 2);
    }
    free(T1); free(T0); free(Y); free(X);
    return lml;
}

/* ---- batched small-n posteriors (GPOptimizer.scala:47-109, GPUnscentedKalmanFilter.scala:63-147) ---- */
#define SMALL(h) ((gp_small *)(intptr_t)(h))
JNIEXPORT jlong JNICALL Java_gpcore_Native_smallFit(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d, jint ldx,
                                                    jdoubleArray y, jint g, jdoubleArray thetas, jdouble sigmaNoiseOrNaN, jint capacity) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *Y = X ? in_d(env, y, 0, (jsize)n * g) : NULL;
    double *T = Y ? in_d(env, thetas, 0, (jsize)g * (d + 2)) : NULL;
    gp_small *s = NULL;
    if (T) {
        int info = 0;
        gp_status st = gp_small_fit(CTX(h), X, n, d, ldx, Y, n, g, T, sigmaNoiseOrNaN, capacity, &s, &info);
        if (st != GP_OK) { throw_for(env, CTX(h), st, info); s = NULL; }
    }
    free(T); free(Y); free(X);
    return (jlong)(intptr_t)s;
}
JNIEXPORT void JNICALL Java_gpcore_Native_smallDestroy(JNIEnv *env, jclass k, jlong s) { gp_small_destroy(SMALL(s)); }
/* mean[g*m], variance[g*m]: model-major */
JNIEXPORT void JNICALL Java_gpcore_Native_smallPosterior(JNIEnv *env, jclass k, jlong h, jlong s, jint g, jdoubleArray xs, jint xsoff, jint m, jint d,
                                                         jint ldxs, jdoubleArray mean, jdoubleArray var) {
    double *XS = in_d(env, xs, xsoff, span(m, d, ldxs)), *ME = XS ? out_d(env, (jsize)g * m) : NULL, *VA = ME ? out_d(env, (jsize)g * m) : NULL;
    if (VA) {
        gp_status st = gp_small_posterior(SMALL(s), XS, m, ldxs, ME, VA);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else { put_d(env, mean, 0, ME, (jsize)g * m); put_d(env, var, 0, VA, (jsize)g * m); }
    }
    free(VA); free(ME); free(XS);
}
JNIEXPORT void JNICALL Java_gpcore_Native_smallUcb(JNIEnv *env, jclass k, jlong h, jlong s, jint g, jdoubleArray xs, jint xsoff, jint m, jint d, jint ldxs,
                                                   jdouble kappa, jdoubleArray value, jdoubleArray grad) {
    double *XS = in_d(env, xs, xsoff, span(m, d, ldxs)), *V = XS ? out_d(env, m) : NULL, *G = V ? out_d(env, (jsize)m * d) : NULL;
    if (G) {
        gp_status st = gp_small_ucb(SMALL(s), g, XS, m, ldxs, kappa, V, G);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else { put_d(env, value, 0, V, m); put_d(env, grad, 0, G, (jsize)m * d); }
    }
    free(G); free(V); free(XS);
}
JNIEXPORT void JNICALL Java_gpcore_Native_smallAppend(JNIEnv *env, jclass k, jlong h, jlong s, jint d, jint g, jdoubleArray xNew, jdoubleArray yNew) {
    double *X = in_d(env, xNew, 0, d), *Y = X ? in_d(env, yNew, 0, g) : NULL;
    if (Y) {
        int info = 0;
        gp_status st = gp_small_append(SMALL(s), X, Y, &info);
        if (st != GP_OK) throw_for(env, CTX(h), st, info);
    }
    free(Y); free(X);
}
/* starts: c x d column-major; bestX[d] out; returns the best UCB value */
JNIEXPORT jdouble JNICALL Java_gpcore_Native_smallMaximizeUcb(JNIEnv *env, jclass k, jlong h, jlong s, jint g, jdoubleArray starts, jint c, jint d,
                                                              jdouble kappa, jint maxIter, jint history, jdoubleArray bestX) {
    double best = 0.0;
    double *S = in_d(env, starts, 0, (jsize)c * d), *B = S ? out_d(env, d) : NULL;
    if (B) {
        gp_status st = gp_small_maximize_ucb(SMALL(s), g, S, c, c, kappa, maxIter, history, B, &best, NULL);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, bestX, 0, B, d);
    }
    free(B); free(S);
    return best;
}

/* ---- Co2Kernel (Co2Prediction.scala:29-137) ---- */
JNIEXPORT jlong JNICALL Java_gpcore_Native_fitCo2(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint n, jdoubleArray y, jdoubleArray theta,
                                                  jdouble sigmaNoiseOrNaN) {
    double *X = in_d(env, x, 0, n), *Y = X ? in_d(env, y, 0, n) : NULL, *T = Y ? in_d(env, theta, 0, 11) : NULL;
    gp_model *m = NULL;
    if (T) {
        int info = 0;
        gp_status st = gp_fit_co2(CTX(h), X, n, Y, T, sigmaNoiseOrNaN, &m, &info);
        if (st != GP_OK) { throw_for(env, CTX(h), st, info); m = NULL; }
    }
    free(T); free(Y); free(X);
    return (jlong)(intptr_t)m;
}
JNIEXPORT void JNICALL Java_gpcore_Native_lmlGradCo2Batched(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint n, jdoubleArray y, jdoubleArray thetas,
                                                            jint B, jint nparams, jdouble sigmaNoiseOrNaN, jdoubleArray lml, jdoubleArray grad,
                                                            jintArray info) {
    double *X = in_d(env, x, 0, n), *Y = X ? in_d(env, y, 0, n) : NULL, *T = Y ? in_d(env, thetas, 0, (jsize)B * 11) : NULL;
    double *L = T ? out_d(env, B) : NULL, *G = L ? out_d(env, (jsize)B * (nparams > 0 ? nparams : 1)) : NULL;
    jint *I = G ? calloc((size_t)(B > 0 ? B : 1), sizeof(jint)) : NULL;
    if (G && !I) throw_for(env, NULL, GP_ENOMEM, 0);
    if (I) {
        gp_status st = gp_lml_grad_co2_batched(CTX(h), X, n, Y, T, B, nparams, sigmaNoiseOrNaN, L, G, (int *)I);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, lml, 0, L, B); put_d(env, grad, 0, G, (jsize)B * nparams); put_i(env, info, I, B); }
    }
    free(I); free(G); free(L); free(T); free(Y); free(X);
}
JNIEXPORT jdouble JNICALL Java_gpcore_Native_optimizeCo2(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint n, jdoubleArray y, jdoubleArray thetaInOut,
                                                         jint nparams, jdouble sigmaNoiseOrNaN, jint maxIter, jint history) {
    double lml = 0.0;
    double *X = in_d(env, x, 0, n), *Y = X ? in_d(env, y, 0, n) : NULL, *T0 = Y ? in_d(env, thetaInOut, 0, 11) : NULL, *T1 = T0 ? out_d(env, 11) : NULL;
    if (T1) {
        gp_status st = gp_optimize_co2(CTX(h), X, n, Y, T0, nparams, sigmaNoiseOrNaN, maxIter, history, T1, &lml, NULL, NULL);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, thetaInOut, 0, T1, 11);
    }
    free(T1); free(T0); free(Y); free(X);
    return lml;
}

/* ---- multi-GPU: one JVM per GPU, RCCL (gp_dist_*) ---- */
#define DIST(h) ((gp_dist *)(intptr_t)(h))
JNIEXPORT jbyteArray JNICALL Java_gpcore_Native_distUniqueId(JNIEnv *env, jclass k, jlong h) {
    unsigned char id[GP_DIST_ID_BYTES];
    gp_status st = gp_dist_unique_id(CTX(h), id);
    if (st != GP_OK) { throw_for(env, CTX(h), st, 0); return NULL; }
    jbyteArray out = (*env)->NewByteArray(env, GP_DIST_ID_BYTES);
    if (out) (*env)->SetByteArrayRegion(env, out, 0, GP_DIST_ID_BYTES, (const jbyte *)id);
    return out;
}
JNIEXPORT jlong JNICALL Java_gpcore_Native_distInit(JNIEnv *env, jclass k, jlong h, jbyteArray id, jint rank, jint world) {
    unsigned char buf[GP_DIST_ID_BYTES];
    if (!id || (*env)->GetArrayLength(env, id) != GP_DIST_ID_BYTES) { throw_for(env, NULL, GP_EINVAL, 0); return 0; }
    (*env)->GetByteArrayRegion(env, id, 0, GP_DIST_ID_BYTES, (jbyte *)buf);
    gp_dist *d = NULL;
    gp_status st = gp_dist_init(CTX(h), buf, rank, world, &d);
    if (st != GP_OK) { throw_for(env, CTX(h), st, 0); return 0; }
    return (jlong)(intptr_t)d;
}
JNIEXPORT void JNICALL Java_gpcore_Native_distDestroy(JNIEnv *env, jclass k, jlong d) { gp_dist_destroy(DIST(d)); }
JNIEXPORT void JNICALL Java_gpcore_Native_distLmlGradBatched(JNIEnv *env, jclass k, jlong h, jlong dist, jdoubleArray x, jint xoff, jint n, jint d, jint ldx,
                                                             jdoubleArray y, jdoubleArray thetas, jint B, jint nparams, jdouble sigmaNoiseOrNaN,
                                                             jdoubleArray lml, jdoubleArray grad, jintArray info) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *Y = X ? in_d(env, y, 0, n) : NULL, *T = Y ? in_d(env, thetas, 0, (jsize)B * (d + 2)) : NULL;
    double *L = T ? out_d(env, B) : NULL, *G = L ? out_d(env, (jsize)B * (nparams > 0 ? nparams : 1)) : NULL;
    jint *I = G ? calloc((size_t)(B > 0 ? B : 1), sizeof(jint)) : NULL;
    if (G && !I) throw_for(env, NULL, GP_ENOMEM, 0);
    if (I) {
        gp_status st = gp_dist_lml_grad_batched(DIST(dist), X, n, d, ldx, Y, T, B, nparams, sigmaNoiseOrNaN, L, G, (int *)I);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, lml, 0, L, B); put_d(env, grad, 0, G, (jsize)B * nparams); put_i(env, info, I, B); }
    }
    free(I); free(G); free(L); free(T); free(Y); free(X);
}
JNIEXPORT void JNICALL Java_gpcore_Native_distPredict(JNIEnv *env, jclass k, jlong h, jlong dist, jlong model, jdoubleArray xs, jint xsoff, jint m, jint d,
                                                      jint ldxs, jdoubleArray mean, jdoubleArray var) {
    double *XS = in_d(env, xs, xsoff, span(m, d, ldxs)), *ME = XS ? out_d(env, m) : NULL, *VA = ME ? out_d(env, m) : NULL;
    if (VA) {
        gp_status st = gp_dist_predict(DIST(dist), MODEL(model), XS, m, ldxs, ME, VA);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else { put_d(env, mean, 0, ME, m); put_d(env, var, 0, VA, m); }
    }
    free(VA); free(ME); free(XS);
}

/* ---- forwards added in round 3 (VERDICT r02 missing #5) ---- */

/* GpPredictor.logLikelihoodWithDerivatives (:60-80) for a KernelFunc without a device form: K and the dK_p come from the Scala
 * loops (MatrixUtils.buildKernelMatrix / buildMatrixWithFunc); dks = Array[Array[Double]], each n x n with leading dimension lddk */
JNIEXPORT jdouble JNICALL Java_gpcore_Native_lmlGradFromGram(JNIEnv *env, jclass k, jlong h, jdoubleArray kk, jint koff, jint n, jint ldk,
                                                             jdoubleArray y, jobjectArray dks, jint lddk, jdouble sigmaNoiseOrNaN,
                                                             jdoubleArray grad) {
    double lml = 0.0;
    const jsize P = dks ? (*env)->GetArrayLength(env, dks) : 0;
    double *K = in_d(env, kk, koff, span(n, n, ldk)), *Y = K ? in_d(env, y, 0, n) : NULL, *G = Y ? out_d(env, P > 0 ? P : 1) : NULL;
    double **D = G ? calloc((size_t)(P > 0 ? P : 1), sizeof(double *)) : NULL;
    if (G && !D) throw_for(env, NULL, GP_ENOMEM, 0);
    int ok = D != NULL;
    for (jsize p = 0; ok && p < P; ++p) {
        jdoubleArray a = (jdoubleArray)(*env)->GetObjectArrayElement(env, dks, p);
        D[p] = in_d(env, a, 0, span(n, n, lddk));          /* a NULL element -> IllegalArgumentException */
        if (a) (*env)->DeleteLocalRef(env, a);
        ok = D[p] != NULL;
    }
    if (ok) {
        int info = 0;
        gp_status st = gp_lml_grad_from_gram(CTX(h), K, n, ldk, Y, (const double *const *)D, (int)P, lddk, sigmaNoiseOrNaN, &lml, G, &info);
        if (st != GP_OK) throw_for(env, CTX(h), st, info); else put_d(env, grad, 0, G, P);
    }
    for (jsize p = 0; D && p < P; ++p) free(D[p]);
    free(D); free(G); free(Y); free(K);
    return lml;
}

/* EP LML and gradient over B settings (HyperParamsOptimization.scala:31-55 / MeshHyperParamsLogLikelihoodEvaluator.scala:26-40), grad B x (d + 2) */
JNIEXPORT void JNICALL Java_gpcore_Native_epLmlGradRbfBatched(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d,
                                                              jint ldx, jintArray y, jdoubleArray thetas, jint B, jdouble stopEps,
                                                              jint maxSweeps, jboolean strict, jdoubleArray lml, jdoubleArray grad,
                                                              jintArray sweeps, jintArray info) {
    const jsize P = d + 2;
    double *X = in_d(env, x, xoff, span(n, d, ldx));
    jint *Y = X ? in_i(env, y, n) : NULL;
    double *T = Y ? in_d(env, thetas, 0, (jsize)B * P) : NULL, *L = T ? out_d(env, B) : NULL, *G = L ? out_d(env, (jsize)B * P) : NULL;
    jint *S = G ? calloc((size_t)(B > 0 ? B : 1), sizeof(jint)) : NULL, *I = S ? calloc((size_t)(B > 0 ? B : 1), sizeof(jint)) : NULL;
    if (G && !I) throw_for(env, NULL, GP_ENOMEM, 0);
    if (I) {
        gp_status st = gp_ep_lml_grad_rbf_batched(CTX(h), X, n, d, ldx, (const int32_t *)Y, T, B, stopEps, maxSweeps, strict ? 1 : 0, L, G,
                                                  (int *)S, (int *)I);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0);
        else { put_d(env, lml, 0, L, B); put_d(env, grad, 0, G, (jsize)B * P); put_i(env, sweeps, S, B); put_i(env, info, I, B); }
    }
    free(I); free(S); free(G); free(L); free(T); free(Y); free(X);
}

/* G small models from the (L, alpha) pairs the caller holds (GPUnscentedKalmanFilter.scala:116-132 keeps one per state dimension):
 * ls = g blocks of n x n (ld n), alphas = g blocks of n */
JNIEXPORT jlong JNICALL Java_gpcore_Native_smallFromFactors(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint xoff, jint n, jint d, jint ldx,
                                                            jdoubleArray thetas, jint g, jdoubleArray ls, jdoubleArray alphas, jint capacity) {
    double *X = in_d(env, x, xoff, span(n, d, ldx)), *T = X ? in_d(env, thetas, 0, (jsize)g * (d + 2)) : NULL;
    double *L = T ? in_d(env, ls, 0, (jsize)g * n * n) : NULL, *A = L ? in_d(env, alphas, 0, (jsize)g * n) : NULL;
    gp_small *s = NULL;
    if (A) {
        gp_status st = gp_small_from_factors(CTX(h), X, n, d, ldx, T, g, L, n, A, capacity, &s);
        if (st != GP_OK) { throw_for(env, CTX(h), st, 0); s = NULL; }
    }
    free(A); free(L); free(T); free(X);
    return (jlong)(intptr_t)s;
}
/* (n, capacity, G) */
JNIEXPORT jintArray JNICALL Java_gpcore_Native_smallSize(JNIEnv *env, jclass k, jlong s) {
    int v[3] = {0, 0, 0};
    gp_status st = gp_small_size(SMALL(s), &v[0], &v[1], &v[2]);
    if (st != GP_OK) { throw_for(env, NULL, st, 0); return NULL; }
    jintArray out = (*env)->NewIntArray(env, 3);
    jint w[3] = {v[0], v[1], v[2]};
    if (out) (*env)->SetIntArrayRegion(env, out, 0, 3, w);
    return out;
}
/* what: 0 = L, 1 = L^-1 (n x n into out with leading dimension ld), 2 = alpha (n) of model g */
JNIEXPORT void JNICALL Java_gpcore_Native_smallGet(JNIEnv *env, jclass k, jlong h, jlong s, jint what, jint g, jdoubleArray out, jint ld) {
    int n = 0, cap = 0, G = 0;
    gp_status st = gp_small_size(SMALL(s), &n, &cap, &G);
    if (st != GP_OK) { throw_for(env, CTX(h), st, 0); return; }
    const jsize cnt = what == GP_SMALL_GET_ALPHA ? n : span(n, n, ld);
    double *O = out_d(env, cnt);
    if (O) {
        st = gp_small_get(SMALL(s), g, what, O, ld);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, O, cnt);
    }
    free(O);
}

/* Co2Kernel matrices (Co2Prediction.scala:39-137) for callers that hold their own factor: computePosterior with co2Kernel */
JNIEXPORT void JNICALL Java_gpcore_Native_gramCo2(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint n, jdoubleArray theta, jdoubleArray out) {
    double *X = in_d(env, x, 0, n), *T = X ? in_d(env, theta, 0, 11) : NULL, *K = T ? out_d(env, (jsize)n * n) : NULL;
    if (K) {
        gp_status st = gp_gram_co2(CTX(h), X, n, T, K, n, GP_FULL);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, K, (jsize)n * n);
    }
    free(K); free(T); free(X);
}
JNIEXPORT void JNICALL Java_gpcore_Native_dgramCo2(JNIEnv *env, jclass k, jlong h, jdoubleArray x, jint n, jdoubleArray theta, jint pos, jdoubleArray out) {
    double *X = in_d(env, x, 0, n), *T = X ? in_d(env, theta, 0, 11) : NULL, *D = T ? out_d(env, (jsize)n * n) : NULL;
    if (D) {
        gp_status st = gp_dgram_co2(CTX(h), X, n, T, pos, D, n);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, D, (jsize)n * n);      /* GP_ERANGE -> MatchError */
    }
    free(D); free(T); free(X);
}
JNIEXPORT void JNICALL Java_gpcore_Native_crossGramCo2(JNIEnv *env, jclass k, jlong h, jdoubleArray xs, jint m, jdoubleArray x, jint n, jdoubleArray theta,
                                                       jdoubleArray out) {
    double *XS = in_d(env, xs, 0, m), *X = XS ? in_d(env, x, 0, n) : NULL, *T = X ? in_d(env, theta, 0, 11) : NULL;
    double *K = T ? out_d(env, (jsize)m * n) : NULL;
    if (K) {
        gp_status st = gp_cross_gram_co2(CTX(h), XS, m, X, n, T, K, m);
        if (st != GP_OK) throw_for(env, CTX(h), st, 0); else put_d(env, out, 0, K, (jsize)m * n);
    }
    free(K); free(T); free(X); free(XS);
}
