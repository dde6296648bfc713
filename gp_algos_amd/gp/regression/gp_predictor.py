"""Mirror of gp/regression/GpPredictor.scala on top of libgpcore.so.

class GpPredictor(kernelFunc) keeps the reference's method names, argument meaning, return tuples and
error behaviour; the bodies call the C-ABI (Gram + blocked Cholesky + triangular solves on the GPU).
GaussianRbfKernel goes through the fused device path; any other KernelFunc falls back to a HOST-BUILT
Gram matrix (the reference's own per-pair loop) that is then factored on the GPU."""
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from ... import default_context
from ...core import RegressionModel
from ...optimization.optimization import BreezeLbfgsOptimizer
from ...utils import matrix_utils
from ...utils.kernel_requisites import GaussianRbfKernel
from .co2_prediction import Co2Kernel
from ...utils.stats_utils import GaussianDistribution


@dataclass
class PredictionInput:   # GpPredictor.scala:162-169
    trainingData: np.ndarray
    testData: np.ndarray
    sigmaNoise: Optional[float]
    targets: np.ndarray

    def toPredictionTrainingInput(self):
        return PredictionTrainingInput(self.trainingData, self.sigmaNoise, self.targets)


@dataclass
class PredictionTrainingInput:   # :171-172
    trainingData: np.ndarray
    sigmaNoise: Optional[float]
    targets: np.ndarray


class GpPredictor:
    def __init__(self, kernelFunc):
        self.kernelFunc = kernelFunc

    # -- internal: fitted device model for (X, y, hyperParams) ----------------------------------
    def _fit(self, trainingData, hyperParams, sigmaNoise, targets):
        X = np.asfortranarray(np.asarray(trainingData, dtype=np.float64))
        y = np.asarray(targets, dtype=np.float64).reshape(-1)
        if X.shape[0] != y.size:   # require(...) :108-109
            raise ValueError("requirement failed: Number of objects in training data matrix should be equal to targets vector length")
        kf = self.kernelFunc.changeHyperParams(hyperParams.toDenseVector())
        ctx = default_context()
        if isinstance(kf, GaussianRbfKernel):
            return kf, RegressionModel(ctx, X, y, kf.rbfParams.toDenseVector(), sigma_noise=sigmaNoise)
        if isinstance(kf, Co2Kernel):            # a composite kernel with a device path of its own (gp_fit_co2)
            if X.shape[1] != 1:
                raise ValueError("requirement failed: This kernel is applicable only for 1D objects")
            return kf, RegressionModel(ctx, X[:, 0], y, kf.hyperParams.toDenseVector(), sigma_noise=sigmaNoise, kernel="co2")
        K = matrix_utils.buildKernelMatrix(kf, X)
        if sigmaNoise is not None:
            K = K + np.eye(X.shape[0]) * sigmaNoise   # un-squared, :116
        return kf, RegressionModel(ctx, y=y, gram=K)

    def preComputeComponents(self, trainingData, *args):
        """preComputeComponents(trainingData, sigmaNoise, targets)                :89-94
           preComputeComponents(trainingData, hyperParams, sigmaNoise, targets)   :104-124
           -> afterLearningComponents = (L, alphaVec, Option[noise * I])          :157"""
        if len(args) == 2:
            hyperParams, (sigmaNoise, targets) = self.kernelFunc.hyperParams, args
        else:
            hyperParams, sigmaNoise, targets = args
        _, mdl = self._fit(trainingData, hyperParams, sigmaNoise, targets)
        try:
            L, alpha = mdl.L(), mdl.alpha()
        finally:
            mdl.close()
        n = L.shape[0]
        return L, alpha, (np.eye(n) * sigmaNoise if sigmaNoise is not None else None)

    def predict(self, input, hyperParams=None):   # :24-43
        hyperParams = self.kernelFunc.hyperParams if hyperParams is None else hyperParams
        kf, mdl = self._fit(input.trainingData, hyperParams, input.sigmaNoise, input.targets)
        try:
            Xs = np.asfortranarray(np.asarray(input.testData, dtype=np.float64))
            if isinstance(kf, (GaussianRbfKernel, Co2Kernel)):
                mean, _, cov = mdl.predict(Xs, full_cov=True)
            else:
                mean, cov = self._posterior_from_host_gram(kf, input.trainingData, Xs, mdl)
            if input.sigmaNoise is not None:
                # fVariance + noiseDiagMtx.get (:37-39): an n x n matrix added to an m x m one -- only defined for m == n
                if cov.shape[0] != mdl.n:
                    raise ValueError("requirement failed: Dimension mismatch! (noiseDiagMtx is n x n, fVariance is m x m)")
                cov = cov + np.eye(mdl.n) * input.sigmaNoise
            return GaussianDistribution(mean=mean, sigma=cov), mdl.lml()
        finally:
            mdl.close()

    def computePosterior(self, trainingData, testData, l, alphaVec, kernelFunc=None):   # :45-58
        """(GaussianDistribution(fMean, fVariance), vMatrix) from a factor the caller holds.  GaussianRbfKernel: everything
        on the device (gp_posterior_from_factor); any other KernelFunc: K* and K** by the reference's per-pair loops on the host,
        the O(n^2 m) solve and the m^2 n product on the device (gp_posterior_from_gram)."""
        kf = self.kernelFunc if kernelFunc is None else kernelFunc
        X = np.asfortranarray(np.asarray(trainingData, dtype=np.float64))
        Xs = np.asfortranarray(np.asarray(testData, dtype=np.float64))
        ctx = default_context()
        if isinstance(kf, GaussianRbfKernel):
            mean, _, cov, V = ctx.posterior_from_factor(X, kf.rbfParams.toDenseVector(), l, alphaVec, Xs, full_cov=True, want_v=True)
        else:
            Ks = matrix_utils.buildKernelMatrix(kf, Xs, X)
            Kss = matrix_utils.buildKernelMatrix(kf, Xs)
            mean, _, cov, V = ctx.posterior_from_gram(Ks, l, alphaVec, Kss=Kss, want_v=True)
        return GaussianDistribution(mean=mean, sigma=cov), V

    def _posterior_from_host_gram(self, kf, trainingData, Xs, mdl):
        Ks = matrix_utils.buildKernelMatrix(kf, Xs, trainingData)
        Kss = matrix_utils.buildKernelMatrix(kf, Xs)
        mean, _, cov = mdl.predict_from_gram(Ks, Kss=Kss)
        return mean, cov

    def logLikelihoodWithDerivatives(self, input, hyperParams, optimizedParamsNum):   # :60-80
        kf = self.kernelFunc.changeHyperParams(hyperParams.toDenseVector())
        X = np.asfortranarray(np.asarray(input.trainingData, dtype=np.float64))
        if isinstance(kf, Co2Kernel):
            lml, grad, info = default_context().lml_grad_co2_batched(X[:, 0], input.targets, kf.hyperParams.toDenseVector()[None, :],
                                                                     nparams=optimizedParamsNum, sigma_noise=input.sigmaNoise)
        elif isinstance(kf, GaussianRbfKernel):
            lml, grad, info = default_context().lml_grad_batched(X, input.targets, kf.rbfParams.toDenseVector()[None, :],
                                                                 nparams=optimizedParamsNum, sigma_noise=input.sigmaNoise)
        else:
            # any other KernelFunc: K and the derivative matrices by the reference's host loops (:62, :74), the
            # factorisation, K^-1 and the traces on the device (gp_lml_grad_from_gram)
            K = matrix_utils.buildKernelMatrix(kf, X)
            dKs = [matrix_utils.buildMatrixWithFunc(X)(kf.derAfterHyperParam(i)) for i in range(1, optimizedParamsNum + 1)]
            return default_context().lml_grad_from_gram(K, input.targets, dKs, sigma_noise=input.sigmaNoise)
        if info[0]:
            from ..._lib import NotPositiveDefinite
            raise NotPositiveDefinite(2, "matrix not positive definite at pivot %d" % info[0], int(info[0]))
        return float(lml[0]), grad[0].copy()

    def obtainOptimalHyperParams(self, trainingData, sigmaNoise, targets, optimizeNoise):   # :126-142
        full = self.kernelFunc.hyperParams.toDenseVector()
        if isinstance(self.kernelFunc, Co2Kernel):
            x = np.asarray(trainingData, dtype=np.float64).reshape(-1)
            best, _, _, _ = default_context().optimize_co2(x, targets, full, nparams=11 if optimizeNoise else 10, sigma_noise=sigmaNoise,
                                                           max_iter=20, history=4)
            return self.kernelFunc.hyperParams.fromDenseVector(best)
        if optimizeNoise and isinstance(self.kernelFunc, GaussianRbfKernel):
            # native L-BFGS (gp_optimize_rbf): same objective, memory (m = 4), iteration cap (20) and best-seen rule as
            # BreezeLbfgsOptimizer; training data stay on the GPU and each line search is one lockstep batch
            X = np.asfortranarray(np.asarray(trainingData, dtype=np.float64))
            best, _, _, _ = default_context().optimize_rbf(X, targets, full, nparams=len(full), sigma_noise=sigmaNoise,
                                                           max_iter=20, history=4)
            return self.kernelFunc.hyperParams.fromDenseVector(best)
        optimizer = BreezeLbfgsOptimizer(maxIter=20)
        init = full if optimizeNoise else full[:-1]

        def objective(params):
            hp = self.kernelFunc.hyperParams.fromDenseVector(params)   # require(dv.length == d+2) :55 -- the
            # reference fails here when optimizeNoise = false; kept as is.
            ti = PredictionTrainingInput(trainingData, sigmaNoise, targets)
            return self.logLikelihoodWithDerivatives(ti, hp, len(params))

        best = optimizer.maximize(objective, init)
        return self.kernelFunc.hyperParams.fromDenseVector(best)

    def predictWithParamsOptimization(self, input, optimizeNoise):   # :82-87
        hp = self.obtainOptimalHyperParams(input.trainingData, input.sigmaNoise, input.targets, optimizeNoise)
        dist, ll = self.predict(input, hyperParams=hp)
        return dist, ll, hp

    def preComputeComponentsWithHpOptimization(self, trainingData, sigmaNoise, targets):   # :96-101
        hp = self.obtainOptimalHyperParams(trainingData, sigmaNoise, targets, True)
        return self.preComputeComponents(trainingData, hp, sigmaNoise, targets), hp

    @staticmethod
    def logLikelihood(alphaVector, L, targets):   # :144-149 (host form, for callers that already hold L and alpha)
        n = L.shape[0]
        a1 = -0.5 * float(np.dot(targets, alphaVector))
        a2 = 0.0
        for i in range(n):
            a2 = a2 + math.log(L[i, i])
        return a1 - a2 - 0.5 * n * math.log(2 * math.pi)
