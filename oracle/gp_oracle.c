/*
 * gp_oracle.c -- CPU restatement of the reference's GP hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the MI355X-native core in gp_algos_amd/csrc.  It is NOT
 * part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product path (libgpcore.so) never links or calls anything in here.
 *
 * It restates, loop order for loop order, the Scala/Breeze code of astroHaoPeng/gp_algos
 * (paths below are relative to /root/reference/src/main/scala/):
 *
 *   utils/KernelRequisites.scala:39-114   GaussianRbfParams / GaussianRbfKernel
 *   utils/MatrixUtils.scala:17-133        Gram builders, forward/back substitution, invTriangular
 *   gp/regression/GpPredictor.scala:24-149  fit, predict, LML, LML gradient
 *   gp/classification/EpParameterEstimator.scala:29-109  EP sweeps, EP LML, probit moments
 *   gp/classification/GpClassifier.scala:24-47           EP predictive probabilities
 *   gp/classification/MarginalLikelihoodEvaluator.scala:46-66  EP LML gradient
 *   utils/StatsUtils.scala:13-17          pnorm / dnorm
 *
 * Third-party arithmetic the reference delegates to Breeze 0.8.1 / netlib-java 1.1.2 (not in the
 * reference tree) is restated from the published reference-BLAS/LAPACK algorithms: dpotf2 (lower),
 * dgemv 'N'/'T', dgemm 'N','N' / 'T','N', ddot -- all plain ascending-index accumulation with one
 * rounding per multiply and per add (compile with -ffp-contract=off: the JVM never fuses).
 *
 * Pinning: the known-answer vectors of the reference's own tests
 * (src/test/scala/utils/MatrixUtilsTest.scala:24-114, KernelRequisitesTest.scala:20-47) are checked
 * in tests/test_oracle_kat.py.  Everything the reference's tests do not pin (Cholesky bits, erf,
 * EP) is "parity unpinned" against the JVM and is cross-checked against scipy/LAPACK instead.
 *
 * All matrices are COLUMN-MAJOR with explicit leading dimension (Breeze DenseMatrix layout:
 * element (i,j) at data[offset + i + j*majorStride]).
 * Hyper-parameter vector theta = [sf, l_1 .. l_d, sn]  (KernelRequisites.scala:48-52), P = d+2.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EL(M, ld, i, j) ((M)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

#define ORC_OK 0
#define ORC_EINVAL 1
#define ORC_ENOTPD 2
#define ORC_ENOMEM 3

/* ------------------------------------------------------------------------------------------ */
/* utils/StatsUtils.scala:13-17 -- Breeze Gaussian(0,1).pdf / .cdf                             */
/* pdf = exp(-x^2/2 - log(sqrt(2 pi)));  cdf = .5*(1+erf(x/sqrt 2)).  erf itself is Breeze's   */
/* (third party, unverifiable offline) -> libm erf here; parity on Phi-dependent outputs is    */
/* stated with that caveat.                                                                    */
/* ------------------------------------------------------------------------------------------ */
double orc_dnorm(double x) { return exp(-(x * x) / 2.0 - log(sqrt(2.0 * M_PI))); }
double orc_pnorm(double x) { return 0.5 * (1.0 + erf(x / sqrt(2.0))); }

/* ------------------------------------------------------------------------------------------ */
/* KernelRequisites.scala:109-113  inputWithLsProduct: ((x-y) :* 1/(l*l)) dot (x-y)            */
/* ------------------------------------------------------------------------------------------ */
static double ls_product(const double *X, size_t ldx, size_t i, const double *Y, size_t ldy, size_t j,
                         int d, const double *theta) {
    double acc = 0.0;
    for (int k = 0; k < d; ++k) {
        double diff = EL(X, ldx, i, k) - EL(Y, ldy, j, k);
        double ls = theta[1 + k];
        double inv = 1.0 / (ls * ls);
        acc = acc + (diff * inv) * diff;
    }
    return acc;
}

/* KernelRequisites.scala:66-72  GaussianRbfKernel.apply */
static double rbf_apply(const double *X, size_t ldx, size_t i, const double *Y, size_t ldy, size_t j, int d,
                        const double *theta, int same) {
    double sf = theta[0], sn = theta[d + 1];
    double v = sf * sf * exp(-0.5 * ls_product(X, ldx, i, Y, ldy, j, d, theta));
    return same ? v + sn * sn : v;
}

/* KernelRequisites.scala:76-86  derAfterHyperParam(p), p is 1-BASED */
static double rbf_der(const double *X, size_t ldx, size_t i, size_t j, int d, const double *theta, int p,
                      int same) {
    double sf = theta[0], sn = theta[d + 1];
    if (p == 1) return 2.0 * sf * exp(-0.5 * ls_product(X, ldx, i, X, ldx, j, d, theta));
    if (p < d + 2) {
        double diff = EL(X, ldx, i, p - 2) - EL(X, ldx, j, p - 2);
        return pow(sf, 2.0) * exp(-0.5 * ls_product(X, ldx, i, X, ldx, j, d, theta)) * pow(diff, 2.0) *
               pow(theta[p - 1], -3.0);
    }
    return same ? 2.0 * sn : 0.0;
}

/* single-pair entry points (scalar KATs) */
double orc_rbf_kernel(const double *x, const double *y, int d, const double *theta, int same) {
    return rbf_apply(x, 1, 0, y, 1, 0, d, theta, same);
}
/* GaussianRbfParams.getAtPosition (KernelRequisites.scala:40-46): 1-based; returns ORC_EINVAL past the
 * end where the Scala throws MatchError. */
int orc_hp_get_at_position(const double *theta, int d, int pos, double *out) {
    if (pos == 1) { *out = theta[0]; return ORC_OK; }
    if (pos > 1 && pos < d + 2) { *out = theta[pos - 1]; return ORC_OK; }
    if (pos == d + 2) { *out = theta[d + 1]; return ORC_OK; }
    return ORC_EINVAL;
}

/* MatrixUtils.scala:57-70  symmetric Gram: lower triangle evaluated, mirrored */
void orc_gram_sym(const double *X, int n, int d, int ldx, const double *theta, double *K, int ldk) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double v = rbf_apply(X, ldx, i, X, ldx, j, d, theta, i == j);
            EL(K, ldk, i, j) = v;
            EL(K, ldk, j, i) = v;
        }
}

/* MatrixUtils.scala:44-55,86-97  cross Gram, never adds noise */
void orc_gram_cross(const double *Xs, int m, int ldxs, const double *X, int n, int ldx, int d,
                    const double *theta, double *Ks, int ldks) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) EL(Ks, ldks, i, j) = rbf_apply(Xs, ldxs, i, X, ldx, j, d, theta, 0);
}

/* MatrixUtils.scala:72-84 with f = derAfterHyperParam(p) */
void orc_dgram_sym(const double *X, int n, int d, int ldx, const double *theta, int p, double *D, int ldd) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double v = rbf_der(X, ldx, i, j, d, theta, p, i == j);
            EL(D, ldd, i, j) = v;
            EL(D, ldd, j, i) = v;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* MatrixUtils.scala:123-133 solveTriangular: acc over the row left-to-right, then divide.     */
/* forward: rows 0..n-1, columns 0..r-1 ascending (:17-21).                                    */
/* back:    rows n-1..0, columns n-1 down to r+1 (:23-27).                                     */
/* `trans` != 0 reads the matrix transposed (the reference passes L.t, a Breeze view).         */
/* ------------------------------------------------------------------------------------------ */
#define TEL(M, ld, i, j, tr) ((tr) ? EL(M, ld, j, i) : EL(M, ld, i, j))

void orc_forward_solve_vec(const double *L, int n, int ldl, int trans, const double *b, double *x) {
    for (int r = 0; r < n; ++r) {
        double acc = 0.0;
        for (int c = 0; c < r; ++c) acc = acc + TEL(L, ldl, r, c, trans) * x[c];
        x[r] = (b[r] - acc) / TEL(L, ldl, r, r, trans);
    }
}
void orc_back_solve_vec(const double *R, int n, int ldr, int trans, const double *b, double *x) {
    for (int r = n - 1; r >= 0; --r) {
        double acc = 0.0;
        for (int c = n - 1; c > r; --c) acc = acc + TEL(R, ldr, r, c, trans) * x[c];
        x[r] = (b[r] - acc) / TEL(R, ldr, r, r, trans);
    }
}
/* MatrixUtils.scala:115-121 column-by-column */
void orc_forward_solve_mat(const double *L, int n, int ldl, int trans, const double *B, int nrhs, int ldb,
                           double *Xo, int ldxo) {
    for (int c = 0; c < nrhs; ++c) orc_forward_solve_vec(L, n, ldl, trans, B + (size_t)c * ldb, Xo + (size_t)c * ldxo);
}
void orc_back_solve_mat(const double *R, int n, int ldr, int trans, const double *B, int nrhs, int ldb,
                        double *Xo, int ldxo) {
    for (int c = 0; c < nrhs; ++c) orc_back_solve_vec(R, n, ldr, trans, B + (size_t)c * ldb, Xo + (size_t)c * ldxo);
}
/* MatrixUtils.scala:106-113 invTriangular: solve against eye(n), no sparsity exploited */
int orc_inv_triangular(const double *T, int n, int ldt, int is_upper, double *Ti, int ldti) {
    double *e = (double *)calloc((size_t)n, sizeof(double));
    if (!e) return ORC_ENOMEM;
    for (int c = 0; c < n; ++c) {
        e[c] = 1.0;
        if (is_upper) orc_back_solve_vec(T, n, ldt, 0, e, Ti + (size_t)c * ldti);
        else orc_forward_solve_vec(T, n, ldt, 0, e, Ti + (size_t)c * ldti);
        e[c] = 0.0;
    }
    free(e);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* breeze.linalg.cholesky (GpPredictor.scala:120, EpParameterEstimator.scala:58) = LAPACK       */
/* dpotrf('L') then zeroed strict upper triangle.  Restated as reference-LAPACK dpotf2 lower:   */
/*   ajj = a_jj - dot(row j, row j); sqrt; column j -= A(j+1:,0:j) * row j (dgemv, k ascending); */
/*   column j *= 1/ajj (dscal by the reciprocal).                                               */
/* info = 0 ok, else 1-based index of the failing pivot (Breeze throws NotConvergedException /  */
/* MatrixNotSymmetricException).                                                                */
/* ------------------------------------------------------------------------------------------ */
int orc_cholesky_lower(double *A, int n, int lda, int *info) {
    *info = 0;
    for (int j = 0; j < n; ++j) {
        double ajj = EL(A, lda, j, j);
        double dot = 0.0;
        for (int k = 0; k < j; ++k) dot = dot + EL(A, lda, j, k) * EL(A, lda, j, k);
        ajj = ajj - dot;
        if (!(ajj > 0.0)) { *info = j + 1; return ORC_ENOTPD; }
        ajj = sqrt(ajj);
        EL(A, lda, j, j) = ajj;
        for (int k = 0; k < j; ++k) {
            double t = -EL(A, lda, j, k);
            if (t != 0.0)
                for (int i = j + 1; i < n; ++i) EL(A, lda, i, j) = EL(A, lda, i, j) + t * EL(A, lda, i, k);
        }
        double r = 1.0 / ajj;
        for (int i = j + 1; i < n; ++i) EL(A, lda, i, j) = r * EL(A, lda, i, j);
    }
    for (int j = 1; j < n; ++j)
        for (int i = 0; i < j; ++i) EL(A, lda, i, j) = 0.0;
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* GpPredictor.scala:104-124 preComputeComponents                                               */
/* sigma_noise: NaN = None; otherwise added UN-SQUARED to the diagonal (:116).                  */
/* L (n x n, ldl) receives the factor, alpha (n) = L^T \ (L \ y).                               */
/* ------------------------------------------------------------------------------------------ */
int orc_fit(const double *X, int n, int d, int ldx, const double *y, const double *theta, double sigma_noise,
            double *L, int ldl, double *alpha, int *info) {
    orc_gram_sym(X, n, d, ldx, theta, L, ldl);
    if (!isnan(sigma_noise))
        for (int i = 0; i < n; ++i) EL(L, ldl, i, i) = EL(L, ldl, i, i) + 1.0 * sigma_noise;
    int rc = orc_cholesky_lower(L, n, ldl, info);
    if (rc) return rc;
    double *t = (double *)malloc((size_t)n * sizeof(double));
    if (!t) return ORC_ENOMEM;
    orc_forward_solve_vec(L, n, ldl, 0, y, t);
    orc_back_solve_vec(L, n, ldl, 1, t, alpha);
    free(t);
    return ORC_OK;
}

/* GpPredictor.scala:144-149 */
double orc_lml(const double *L, int n, int ldl, const double *alpha, const double *y) {
    double dot = 0.0;
    for (int i = 0; i < n; ++i) dot = dot + y[i] * alpha[i];
    double a1 = -0.5 * dot;
    double a2 = 0.0;
    for (int i = 0; i < n; ++i) a2 = a2 + log(EL(L, ldl, i, i));
    return a1 - a2 - 0.5 * n * log(2.0 * M_PI);
}

/* ------------------------------------------------------------------------------------------ */
/* GpPredictor.scala:24-43 predict / :45-58 computePosterior, given (L, alpha).                 */
/* mean[m]; cov (m x m, ldc) if non-NULL = Gram_sym(X*) - V^T V (full, as the reference);       */
/* var_diag[m] if non-NULL = its diagonal only (same arithmetic for those entries).             */
/* Vout (n x m, ldv) if non-NULL receives V = L \ K*^T.                                         */
/* ------------------------------------------------------------------------------------------ */
int orc_predict(const double *X, int n, int d, int ldx, const double *theta, const double *L, int ldl,
                const double *alpha, const double *Xs, int m, int ldxs, double *mean, double *var_diag,
                double *cov, int ldc, double *Vout, int ldv) {
    double *Ks = (double *)malloc((size_t)m * n * sizeof(double));
    double *V = (double *)malloc((size_t)n * m * sizeof(double));
    double *col = (double *)malloc((size_t)n * sizeof(double));
    if (!Ks || !V || !col) { free(Ks); free(V); free(col); return ORC_ENOMEM; }
    orc_gram_cross(Xs, m, ldxs, X, n, ldx, d, theta, Ks, m);
    /* fMean = K* alpha : dgemv 'N', column-wise axpy, j ascending */
    for (int i = 0; i < m; ++i) mean[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double t = alpha[j];
        for (int i = 0; i < m; ++i) mean[i] = mean[i] + t * EL(Ks, m, i, j);
    }
    /* V = forwardSolve(L, K*^T): column c of K*^T is row c of K* */
    for (int c = 0; c < m; ++c) {
        for (int r = 0; r < n; ++r) col[r] = EL(Ks, m, c, r);
        orc_forward_solve_vec(L, n, ldl, 0, col, V + (size_t)c * n);
    }
    if (Vout)
        for (int c = 0; c < m; ++c) memcpy(Vout + (size_t)c * ldv, V + (size_t)c * n, (size_t)n * sizeof(double));
    /* fVariance = buildKernelMatrix(X*) - V^T V   (dgemm 'T','N': dot over l ascending) */
    for (int i = 0; i < m; ++i) {
        int jlo = cov ? 0 : i, jhi = cov ? m : i + 1;
        for (int j = jlo; j < jhi; ++j) {
            double dot = 0.0;
            const double *vi = V + (size_t)i * n, *vj = V + (size_t)j * n;
            for (int l = 0; l < n; ++l) dot = dot + vi[l] * vj[l];
            int hi = i > j ? i : j, lo = i > j ? j : i;
            double kss = rbf_apply(Xs, ldxs, hi, Xs, ldxs, lo, d, theta, i == j);
            double v = kss - dot;
            if (cov) EL(cov, ldc, i, j) = v;
            if (i == j && var_diag) var_diag[i] = v;
        }
    }
    free(Ks); free(V); free(col);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* GpPredictor.scala:60-80 logLikelihoodWithDerivatives                                         */
/*   Linv = invTriangular(L); Kinv = Linv^T Linv (dgemm 'T','N'); W = alpha alpha^T - Kinv;     */
/*   g_p = 0.5 * trace(W * D_p), D_p = buildMatrixWithFunc(derAfterHyperParam(p+1)).            */
/* trace(W*D) only consumes the diagonal of the dgemm product, whose entries are                */
/* sum_l W(i,l)*D(l,i) (l ascending) -- computed directly, identical rounding, O(n^2) not O(n^3) */
/* ------------------------------------------------------------------------------------------ */
int orc_lml_grad(const double *X, int n, int d, int ldx, const double *y, const double *theta,
                 double sigma_noise, int nparams, double *lml, double *grad, int *info) {
    size_t nn = (size_t)n * n;
    double *L = (double *)malloc(nn * sizeof(double));
    double *Li = (double *)malloc(nn * sizeof(double));
    double *W = (double *)malloc(nn * sizeof(double));
    double *D = (double *)malloc(nn * sizeof(double));
    double *alpha = (double *)malloc((size_t)n * sizeof(double));
    int rc = ORC_ENOMEM;
    if (!L || !Li || !W || !D || !alpha) goto done;
    rc = orc_fit(X, n, d, ldx, y, theta, sigma_noise, L, n, alpha, info);
    if (rc) goto done;
    *lml = orc_lml(L, n, n, alpha, y);
    rc = orc_inv_triangular(L, n, n, 0, Li, n);
    if (rc) goto done;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            double dot = 0.0;
            for (int l = 0; l < n; ++l) dot = dot + EL(Li, n, l, i) * EL(Li, n, l, j);
            EL(W, n, i, j) = alpha[i] * alpha[j] - dot;
        }
    for (int p = 0; p < nparams; ++p) {
        orc_dgram_sym(X, n, d, ldx, theta, p + 1, D, n);
        double tr = 0.0;
        for (int i = 0; i < n; ++i) {
            double cii = 0.0;
            for (int l = 0; l < n; ++l) cii = cii + EL(W, n, i, l) * EL(D, n, l, i);
            tr = tr + cii;
        }
        grad[p] = 0.5 * tr;
    }
done:
    free(L); free(Li); free(W); free(D); free(alpha);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* KernelRequisites.scala:95-107  GaussianRbfKernel.gradient(afterFirstArg)(vec1, vec2):        */
/*   diff = vec1 - vec2;  a1 = apply(vec1, vec2, sameIndex = false);                            */
/*   afterFirstArg ? (diff :* 1/(l*l)) :* (-a1) : (diff :* 1/(l*l)) :* a1                      */
/* ------------------------------------------------------------------------------------------ */
void orc_kernel_gradient(const double *v1, const double *v2, int d, const double *theta, int after_first, double *out) {
    double a1 = rbf_apply(v1, 1, 0, v2, 1, 0, d, theta, 0);
    for (int k = 0; k < d; ++k) {
        double diff = v1[k] - v2[k];
        double ls = theta[1 + k];
        double inv = 1.0 / (ls * ls);
        out[k] = after_first ? (diff * inv) * (-a1) : (diff * inv) * a1;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* gp/optimization/GPOptimizer.scala:82-109  maximizeUCB's funcForOptimizer at ONE test point:   */
/*   (dist, v) = computePosterior(pointSet, x, ll, alphaVec)                    :91             */
/*   ucb      = mean(0) + kParam * sqrt(sigma(0,0))                             :94             */
/*   testTrainDer (d x n): column j = gradient(afterFirstArg = true)(x, X_j)    :95, :112-127   */
/*   trainTestDer (n x d): row j    = gradient(afterFirstArg = false)(X_j, x)   :96             */
/*   derAfterMean = testTrainDer * alphaVec                                     :97             */
/*   derAfterVarFirst = gradient(true)(x, x)                                    :98             */
/*   vAfterXDer = inversedL * trainTestDer   (inversedL = invTriangular(ll))    :85,:99         */
/*   derAfterVar = derAfterVarFirst - (vAfterXDer^T * v) :* 2                   :101            */
/*   ucbDer = derAfterMean + derAfterVar :* (kParam / (2 sqrt(sigma(0,0))))     :102-103        */
/* X is n x d column-major (pointSet), x the test point (d).                                    */
/* ------------------------------------------------------------------------------------------ */
int orc_inv_triangular(const double *T, int n, int ldt, int is_upper, double *out, int ldo);
int orc_ucb(const double *X, int n, int d, int ldx, const double *theta, const double *L, int ldl, const double *alpha,
            const double *x, double kappa, double *value, double *grad) {
    double *ks = (double *)malloc((size_t)n * sizeof(double)), *v = (double *)malloc((size_t)n * sizeof(double));
    double *Li = (double *)malloc((size_t)n * n * sizeof(double));
    double *tt = (double *)malloc((size_t)n * d * sizeof(double)), *vx = (double *)malloc((size_t)n * d * sizeof(double));
    double *row = (double *)malloc((size_t)d * sizeof(double)), *g = (double *)malloc((size_t)d * sizeof(double));
    if (!ks || !v || !Li || !tt || !vx || !row || !g) { free(ks); free(v); free(Li); free(tt); free(vx); free(row); free(g); return ORC_ENOMEM; }
    /* computePosterior, m = 1 */
    for (int j = 0; j < n; ++j) ks[j] = rbf_apply(x, 1, 0, X, ldx, j, d, theta, 0);
    double mean = 0.0;
    for (int j = 0; j < n; ++j) mean = mean + ks[j] * alpha[j];
    for (int r = 0; r < n; ++r) {                      /* forwardSolve(L, k*)  MatrixUtils.scala:123-133 */
        double acc = 0.0;
        for (int c = 0; c < r; ++c) acc = acc + EL(L, ldl, r, c) * v[c];
        v[r] = (ks[r] - acc) / EL(L, ldl, r, r);
    }
    double vv = 0.0;
    for (int r = 0; r < n; ++r) vv = vv + v[r] * v[r];
    double sigma = rbf_apply(x, 1, 0, x, 1, 0, d, theta, 1) - vv;     /* buildKernelMatrix(kernel, testData) has sn^2 on its diagonal */
    *value = mean + kappa * sqrt(sigma);
    int rc = orc_inv_triangular(L, n, ldl, 0, Li, n);
    if (rc) { free(ks); free(v); free(Li); free(tt); free(vx); free(row); free(g); return rc; }
    for (int k = 0; k < d; ++k) g[k] = 0.0;
    double *xj = (double *)malloc((size_t)d * sizeof(double));
    for (int j = 0; j < n; ++j) {
        for (int k = 0; k < d; ++k) xj[k] = EL(X, ldx, j, k);
        orc_kernel_gradient(xj, x, d, theta, 0, row);                    /* trainTestDer row j */
        for (int k = 0; k < d; ++k) EL(tt, n, j, k) = row[k];
    }
    /* derAfterMean = testTrainDer * alpha, column-wise axpy (dgemv 'N'), j ascending */
    for (int j = 0; j < n; ++j) {
        for (int k = 0; k < d; ++k) xj[k] = EL(X, ldx, j, k);
        orc_kernel_gradient(x, xj, d, theta, 1, row);                    /* testTrainDer column j */
        for (int k = 0; k < d; ++k) g[k] = g[k] + row[k] * alpha[j];
    }
    /* vAfterXDer = inversedL * trainTestDer  (dgemm 'N','N': for each column, l ascending axpy) */
    for (int k = 0; k < d; ++k) {
        for (int i = 0; i < n; ++i) EL(vx, n, i, k) = 0.0;
        for (int l = 0; l < n; ++l) {
            double b = EL(tt, n, l, k);
            for (int i = 0; i < n; ++i) EL(vx, n, i, k) = EL(vx, n, i, k) + EL(Li, n, i, l) * b;
        }
    }
    orc_kernel_gradient(x, x, d, theta, 1, row);                          /* derAfterVarFirst (a vector of -0.0) */
    double coeff = kappa / (2.0 * sqrt(sigma));
    for (int k = 0; k < d; ++k) {
        double dot = 0.0;
        for (int i = 0; i < n; ++i) dot = dot + EL(vx, n, i, k) * v[i];
        double der_var = row[k] - dot * 2.0;
        grad[k] = g[k] + der_var * coeff;
    }
    free(xj); free(ks); free(v); free(Li); free(tt); free(vx); free(row); free(g);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* gp/regression/Co2Prediction.scala:29-137  Co2Kernel: apply (:39-56) and derAfterHyperParam   */
/* (:69-137) on 1-D inputs, hp = hp1..hp11 (Co2HyperParams.getAtPosition is 1-based, :24).      */
/* ------------------------------------------------------------------------------------------ */
double orc_co2_kernel(double x1, double x2, int same, const double *hp) {
    double hp1 = hp[0], hp2 = hp[1], hp3 = hp[2], hp4 = hp[3], hp5 = hp[4], hp6 = hp[5], hp7 = hp[6], hp8 = hp[7], hp9 = hp[8],
           hp10 = hp[9], hp11 = hp[10];
    double xDiff = x1 - x2, xDiffSq = (x1 - x2) * (x1 - x2);
    double k1Val = hp1 * hp1 * exp(-xDiffSq / (2 * hp2 * hp2));
    double sinVal = sin(M_PI * xDiff);
    double k2Val = hp3 * hp3 * exp((-xDiffSq / (2 * hp4 * hp4)) - 2 * sinVal * sinVal / (hp5 * hp5));
    double k3Pow1 = 1 + xDiffSq / (2 * hp8 * hp7 * hp7);
    double k3Val = hp6 * hp6 * pow(k3Pow1, -hp8);
    double k4Val = hp9 * hp9 * exp(-xDiffSq / (2 * hp10 * hp10));
    double indNoise = same ? hp11 * hp11 : 0.0;
    return k1Val + k2Val + k3Val + k4Val + indNoise;
}
/* num is 1-BASED; returns ORC_EINVAL past 11 where the Scala match throws MatchError */
int orc_co2_der(double x1, double x2, int same, int num, const double *hp, double *out) {
    double hp1 = hp[0], hp2 = hp[1], hp3 = hp[2], hp4 = hp[3], hp5 = hp[4], hp6 = hp[5], hp7 = hp[6], hp8 = hp[7], hp9 = hp[8],
           hp10 = hp[9], hp11 = hp[10];
    double xDiff = x1 - x2, sqDiff = (x1 - x2) * (x1 - x2);
    if (num < 1 || num > 11) return ORC_EINVAL;
    if (num < 3) {
        *out = (num == 1) ? 2 * hp1 * exp(-sqDiff / (2 * hp2 * hp2)) : hp1 * hp1 * exp(-sqDiff / (2 * hp2 * hp2)) * sqDiff * pow(hp2, -3.0);
    } else if (num < 6) {
        double sinVal = sin(M_PI * xDiff);
        double k2Val = hp3 * hp3 * exp(-sqDiff / (2 * hp4 * hp4) - 2 * sinVal * sinVal / (hp5 * hp5));
        *out = (num == 3) ? 2 * k2Val / hp3 : (num == 4) ? k2Val * sqDiff * pow(hp4, -3.0) : k2Val * 4 * sinVal * sinVal * pow(hp5, -3.0);
    } else if (num < 9) {
        double k3Pow1 = 1 + sqDiff / (2 * hp8 * hp7 * hp7);
        if (num == 6) *out = 2 * hp6 * pow(k3Pow1, -hp8);
        else if (num == 7) *out = hp6 * hp6 * pow(k3Pow1, -hp8 - 1) * sqDiff * pow(hp7, -3.0);
        else {
            double firstTerm = exp(-hp8 * log(k3Pow1));
            double secondTerm = -log(k3Pow1) + (hp8 * sqDiff / (2 * hp7 * hp7 * hp8 * hp8 * k3Pow1));
            *out = hp6 * hp6 * firstTerm * secondTerm;
        }
    } else {
        double k4Val = hp9 * hp9 * exp(-sqDiff / (2 * hp10 * hp10));
        *out = (num == 9) ? 2 * k4Val / hp9 : (num == 10) ? k4Val * sqDiff * pow(hp10, -3.0) : (same ? 2 * hp11 : 0.0);
    }
    return ORC_OK;
}
/* MatrixUtils.buildKernelMatrix / buildMatrixWithFunc with the kernel above: pos = 0 the kernel, 1..11 a derivative; symmetric
 * (xs == NULL: lower evaluated with the i == j flag, mirrored) or cross (never the noise flag) */
int orc_co2_gram(const double *x, int n, const double *xs, int m, const double *hp, int pos, double *K, int ldk) {
    if (!xs) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                double v;
                if (pos) { int rc = orc_co2_der(x[i], x[j], i == j, pos, hp, &v); if (rc) return rc; }
                else v = orc_co2_kernel(x[i], x[j], i == j, hp);
                EL(K, ldk, i, j) = v;
                EL(K, ldk, j, i) = v;
            }
        return ORC_OK;
    }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) EL(K, ldk, i, j) = orc_co2_kernel(xs[i], x[j], 0, hp);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* EpParameterEstimator.scala:98-109 marginalMoments                                            */
/* ------------------------------------------------------------------------------------------ */
static void marginal_moments(double cav_mi, double cav_sigma, int target, double *mi_hat, double *sigma_hat) {
    double temp = sqrt(1.0 + cav_sigma);
    double z = (target * cav_mi) / temp;
    double dz = orc_dnorm(z), pz = orc_pnorm(z);
    *mi_hat = cav_mi + (target * cav_sigma * dz) / (pz * temp);
    *sigma_hat = cav_sigma - ((cav_sigma * cav_sigma * dz) * (z + dz / pz)) / ((1.0 + cav_sigma) * pz);
}
void orc_marginal_moments(double cav_mi, double cav_sigma, int target, double *mi_hat, double *sigma_hat) {
    marginal_moments(cav_mi, cav_sigma, target, mi_hat, sigma_hat);
}

/* y = A x, dgemv 'N' (column-wise axpy, beta = 0) */
static void gemv_n(const double *A, int n, int lda, const double *x, double *y) {
    for (int i = 0; i < n; ++i) y[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double t = x[j];
        for (int i = 0; i < n; ++i) y[i] = y[i] + t * EL(A, lda, i, j);
    }
}

/* EpParameterEstimator.scala:195-202 avgBetweenSiteParams: (sum/2)*n -- precedence as written */
double orc_avg_between_site_params(const double *old_tau, const double *old_nu, const double *cur_tau,
                                   const double *cur_nu, int n) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + (cur_nu[i] - old_nu[i]) + (cur_tau[i] - old_tau[i]);
    return s / 2 * n;
}

/* ------------------------------------------------------------------------------------------ */
/* EpParameterEstimator.scala:71-96 epMarginalLikelihood.                                        */
/* strict != 0: as COMPILED (the `+ 0.5*log(..) - log(L_ii)` line at :92 is a dropped statement) */
/* strict == 0: the intended formula (R transcription at :134-139).                              */
/* ------------------------------------------------------------------------------------------ */
double orc_ep_lml(const double *Sigma, int n, int lds, const double *L, int ldl, const double *tau,
                  const double *nu, const double *cav_tau, const double *cav_nu, const int *y, int strict) {
    double *t1 = (double *)malloc((size_t)n * sizeof(double));
    /* (nu^T * temp2) * nu, temp2 = Sigma - diag(1/(tau+cav_tau)) */
    double first = 0.0;
    for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int i = 0; i < n; ++i) {
            double e = EL(Sigma, lds, i, j);
            if (i == j) e = e - 1.0 / (tau[i] + cav_tau[i]);
            else e = e - 0.0;
            acc = acc + e * nu[i];
        }
        t1[j] = acc;
    }
    for (int j = 0; j < n; ++j) first = first + t1[j] * nu[j];
    double second = 0.0;
    for (int i = 0; i < n; ++i) {
        double cav_mi = cav_nu[i] / cav_tau[i];
        double temp3 = (cav_mi * cav_tau[i]) * (1.0 / (tau[i] + cav_tau[i]));
        double temp4 = (tau[i] * cav_mi) - (nu[i] * 2.0);
        second = second + temp3 * temp4;
    }
    double third = 0.0, fourth = 0.0;
    for (int i = 0; i < n; ++i) {
        double cav_mi = cav_nu[i] / cav_tau[i];
        third = third + log(orc_pnorm(y[i] * cav_mi / sqrt(1.0 + 1.0 / cav_tau[i])));
        if (!strict) fourth = fourth + 0.5 * log(1.0 + tau[i] / cav_tau[i]) - log(EL(L, ldl, i, i));
    }
    free(t1);
    return third + fourth + 0.5 * (first + second);
}

/* ------------------------------------------------------------------------------------------ */
/* EpParameterEstimator.scala:29-69 estimateSiteParams with a FIXED number of sweeps, or        */
/* (max_sweeps, eps >= 0) the AvgBasedStopCriterion (:187-193): stop before sweep j>0 when       */
/* |avgBetweenSiteParams(old,cur)| < eps.  Outputs: tau, nu (site params), cav_tau, cav_nu        */
/* (cavity values left by the LAST sweep's site loop), Sigma, mu, L (end-of-sweep), sweeps_done. */
/* ------------------------------------------------------------------------------------------ */
int orc_ep_estimate(const double *K, int n, int ldk, const int *y, int max_sweeps, double eps, double *tau,
                    double *nu, double *cav_tau, double *cav_nu, double *Sigma, double *mu, double *L,
                    int *sweeps_done, int *info) {
    size_t nn = (size_t)n * n;
    double *s = (double *)malloc((size_t)n * sizeof(double));
    double *st = (double *)malloc((size_t)n * sizeof(double));
    double *old_tau = (double *)malloc((size_t)n * sizeof(double));
    double *old_nu = (double *)malloc((size_t)n * sizeof(double));
    double *B = (double *)malloc(nn * sizeof(double));
    double *V = (double *)malloc(nn * sizeof(double));
    int rc = ORC_ENOMEM;
    if (!s || !st || !old_tau || !old_nu || !B || !V) goto done;
    for (int i = 0; i < n; ++i) tau[i] = nu[i] = mu[i] = cav_tau[i] = cav_nu[i] = old_tau[i] = old_nu[i] = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) EL(Sigma, n, i, j) = EL(K, ldk, i, j);
    *sweeps_done = 0;
    *info = 0;
    for (int sw = 0; sw < max_sweeps; ++sw) {
        if (sw > 0 && eps >= 0.0 && fabs(orc_avg_between_site_params(old_tau, old_nu, tau, nu, n)) < eps) break;
        memcpy(old_tau, tau, (size_t)n * sizeof(double));
        memcpy(old_nu, nu, (size_t)n * sizeof(double));
        for (int i = 0; i < n; ++i) {
            double sii = EL(Sigma, n, i, i);
            cav_tau[i] = 1.0 / sii - tau[i];
            cav_nu[i] = mu[i] / sii - nu[i];
            double mi_hat, sg_hat;
            marginal_moments(cav_nu[i] / cav_tau[i], 1.0 / cav_tau[i], y[i], &mi_hat, &sg_hat);
            double dtau = 1.0 / sg_hat - cav_tau[i] - tau[i];
            tau[i] = tau[i] + dtau;
            nu[i] = mi_hat / sg_hat - cav_nu[i];
            for (int r = 0; r < n; ++r) s[r] = EL(Sigma, n, r, i);
            double c = 1.0 / (1.0 / dtau + sii);
            for (int b = 0; b < n; ++b)
                for (int a = 0; a < n; ++a) EL(Sigma, n, a, b) = EL(Sigma, n, a, b) - (s[a] * s[b]) * c;
            gemv_n(Sigma, n, n, nu, mu);
        }
        for (int i = 0; i < n; ++i) st[i] = sqrt(tau[i]);
        /* L = chol(I + (st st^T) .* K) */
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i)
                EL(L, n, i, j) = (i == j ? 1.0 : 0.0) + (st[i] * st[j]) * EL(K, ldk, i, j);
        rc = orc_cholesky_lower(L, n, n, info);
        if (rc) goto done;
        /* V = L \ (cloneCols(st,n) .* K): row-scaling */
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) EL(B, n, i, j) = st[i] * EL(K, ldk, i, j);
        orc_forward_solve_mat(L, n, n, 0, B, n, n, V, n);
        /* Sigma = K - V^T V  (dgemm 'T','N') */
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                double dot = 0.0;
                for (int l = 0; l < n; ++l) dot = dot + EL(V, n, l, i) * EL(V, n, l, j);
                EL(Sigma, n, i, j) = EL(K, ldk, i, j) - dot;
            }
        gemv_n(Sigma, n, n, nu, mu);
        *sweeps_done = sw + 1;
    }
    rc = ORC_OK;
done:
    free(s); free(st); free(old_tau); free(old_nu); free(B); free(V);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* GpClassifier.scala:24-47 classify, from ready-made K (n x n), K* (m x n), diag(K**) (m).      */
/* Only the diagonal of K** - V^T V is consumed (:43-45).                                        */
/* ------------------------------------------------------------------------------------------ */
int orc_ep_classify(const double *K, int n, int ldk, const double *L, int ldl, const double *tau,
                    const double *nu, const double *Ks, int m, int ldks, const double *kss_diag,
                    double *prob, double *fmean_out, double *fvar_out) {
    double *st = (double *)malloc((size_t)n * sizeof(double));
    double *rhs = (double *)malloc((size_t)n * sizeof(double));
    double *t = (double *)malloc((size_t)n * sizeof(double));
    double *z = (double *)malloc((size_t)n * sizeof(double));
    double *w = (double *)malloc((size_t)n * sizeof(double));
    double *col = (double *)malloc((size_t)n * sizeof(double));
    double *v = (double *)malloc((size_t)n * sizeof(double));
    if (!st || !rhs || !t || !z || !w || !col || !v) { free(st); free(rhs); free(t); free(z); free(w); free(col); free(v); return ORC_ENOMEM; }
    for (int i = 0; i < n; ++i) st[i] = sqrt(tau[i]);
    /* rhs = (K(::,*) :* st) * nu : row i of K scaled by st_i, then dgemv */
    for (int i = 0; i < n; ++i) rhs[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double tj = nu[j];
        for (int i = 0; i < n; ++i) rhs[i] = rhs[i] + tj * (EL(K, ldk, i, j) * st[i]);
    }
    orc_forward_solve_vec(L, n, ldl, 0, rhs, t);
    orc_back_solve_vec(L, n, ldl, 1, t, z);
    for (int i = 0; i < n; ++i) { z[i] = st[i] * z[i]; w[i] = nu[i] - z[i]; }
    for (int i = 0; i < m; ++i) {
        /* fMean_i = sum_j K*(i,j) w_j, dgemv 'N' order == j ascending per output */
        double fm = 0.0;
        for (int j = 0; j < n; ++j) fm = fm + w[j] * EL(Ks, ldks, i, j);
        for (int r = 0; r < n; ++r) col[r] = EL(Ks, ldks, i, r) * st[r];
        orc_forward_solve_vec(L, n, ldl, 0, col, v);
        double dot = 0.0;
        for (int l = 0; l < n; ++l) dot = dot + v[l] * v[l];
        double fv = kss_diag[i] - dot;
        prob[i] = orc_pnorm(fm / sqrt(1.0 + fv));
        if (fmean_out) fmean_out[i] = fm;
        if (fvar_out) fvar_out[i] = fv;
    }
    free(st); free(rhs); free(t); free(z); free(w); free(col); free(v);
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* MarginalLikelihoodEvaluator.scala:46-66 logLikelihoodDerivativesAfterHyperParams.             */
/* strict != 0: as COMPILED: rMatrix = b b^T only (the `- backSolve(..)` at :59 is a dropped     */
/*   statement); b = nu - (S L) \ (L^T \ (S K nu)).                                              */
/* strict == 0: Rasmussen & Williams Alg. 5.2: b = nu - S L^T\(L\(S K nu)),                     */
/*   R = b b^T - S (L L^T)^-1 S.                                                                 */
/* g_p = 0.5 * trace(R * C_p), all P = d+2 parameters.                                           */
/* ------------------------------------------------------------------------------------------ */
int orc_ep_lml_grad(const double *X, int n, int d, int ldx, const double *theta, const double *K, int ldk,
                    const double *L, int ldl, const double *tau, const double *nu, int strict, double *grad) {
    int P = d + 2;
    size_t nn = (size_t)n * n;
    double *st = (double *)malloc((size_t)n * sizeof(double));
    double *rhs = (double *)malloc((size_t)n * sizeof(double));
    double *t = (double *)malloc((size_t)n * sizeof(double));
    double *b = (double *)malloc((size_t)n * sizeof(double));
    double *R = (double *)malloc(nn * sizeof(double));
    double *C = (double *)malloc(nn * sizeof(double));
    double *SL = (double *)malloc(nn * sizeof(double));
    double *T1 = NULL;
    int rc = ORC_ENOMEM;
    if (!st || !rhs || !t || !b || !R || !C || !SL) goto done;
    for (int i = 0; i < n; ++i) st[i] = sqrt(tau[i]);
    for (int i = 0; i < n; ++i) rhs[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double tj = nu[j];
        for (int i = 0; i < n; ++i) rhs[i] = rhs[i] + tj * (st[i] * EL(K, ldk, i, j));
    }
    if (strict) {
        orc_back_solve_vec(L, n, ldl, 1, rhs, t);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) EL(SL, n, i, j) = st[i] * EL(L, ldl, i, j);
        orc_forward_solve_vec(SL, n, n, 0, t, b);
        for (int i = 0; i < n; ++i) b[i] = nu[i] - b[i];
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) EL(R, n, i, j) = b[i] * b[j];
    } else {
        orc_forward_solve_vec(L, n, ldl, 0, rhs, t);
        orc_back_solve_vec(L, n, ldl, 1, t, b);
        for (int i = 0; i < n; ++i) b[i] = nu[i] - st[i] * b[i];
        /* T1 = L \ diag(st);  R = b b^T - T1^T T1 */
        T1 = (double *)calloc(nn, sizeof(double));
        if (!T1) goto done;
        for (int c = 0; c < n; ++c) {
            for (int r = 0; r < n; ++r) t[r] = (r == c) ? st[c] : 0.0;
            orc_forward_solve_vec(L, n, ldl, 0, t, T1 + (size_t)c * n);
        }
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                double dot = 0.0;
                for (int l = 0; l < n; ++l) dot = dot + EL(T1, n, l, i) * EL(T1, n, l, j);
                EL(R, n, i, j) = b[i] * b[j] - dot;
            }
    }
    for (int p = 0; p < P; ++p) {
        orc_dgram_sym(X, n, d, ldx, theta, p + 1, C, n);
        double tr = 0.0;
        for (int i = 0; i < n; ++i) {
            double cii = 0.0;
            for (int l = 0; l < n; ++l) cii = cii + EL(R, n, i, l) * EL(C, n, l, i);
            tr = tr + cii;
        }
        grad[p] = 0.5 * tr;
    }
    rc = ORC_OK;
done:
    free(st); free(rhs); free(t); free(b); free(R); free(C); free(SL); free(T1);
    return rc;
}
