"""Mirror of gp/classification/MarginalLikelihoodEvaluator.scala and a corrected batched stand-in for
MeshHyperParamsLogLikelihoodEvaluator.scala (the reference's mesh map is mis-keyed, SURVEY.md A23: results here are
returned by setting index)."""
import itertools

import numpy as np

from ...utils import matrix_utils
from ...utils.kernel_requisites import GaussianRbfKernel
from ... import default_context
from .ep_parameter_estimator import AvgBasedStopCriterion, EpParameterEstimator, FixedSweepsStopCriterion


class MarginalLikelihoodEvaluator:
    """class MarginalLikelihoodEvaluator(stopCriterion, kernelFunc)  (:13).  strict=True reproduces the code as compiled
    (EP LML without the dropped term, rMatrix = b b^T); strict=False evaluates the intended formulas."""

    def __init__(self, stopCriterion, kernelFunc, strict=True):
        self.stopCriterion, self.kernelFunc, self.strict = stopCriterion, kernelFunc, strict

    def logLikelihoodWithKernelMatrixPassed(self, kernelMatrix, targets):   # :18-22
        (site, _) = EpParameterEstimator(kernelMatrix, targets, self.stopCriterion, strict=self.strict).estimateSiteParams()
        return site.marginalLogLikelihood

    def logLikelihoodWithoutGrad(self, trainInput, targets, hyperParams):   # :24-31
        kf = self.kernelFunc.changeHyperParams(np.asarray(hyperParams, dtype=np.float64))
        K = matrix_utils.buildKernelMatrix(kf, trainInput)
        return self.logLikelihoodWithKernelMatrixPassed(K, targets)

    def logLikelihood(self, trainInput, targets, hyperParams):   # :33-44 -> (lml, gradient)
        kf = self.kernelFunc.changeHyperParams(np.asarray(hyperParams, dtype=np.float64))
        if not isinstance(kf, GaussianRbfKernel):
            raise NotImplementedError("device EP gradient is implemented for GaussianRbfKernel")
        X = np.asfortranarray(np.asarray(trainInput, dtype=np.float64))
        K = matrix_utils.buildKernelMatrix(kf, X)
        (site, _), st = EpParameterEstimator(K, targets, self.stopCriterion, strict=self.strict).estimateSiteParams(keep_state=True)
        try:
            grad = st.lml_grad_rbf(X, kf.rbfParams.toDenseVector(), strict=self.strict)
        finally:
            st.close()
        return site.marginalLogLikelihood, grad


class MeshHyperParamsLogLikelihoodEvaluator:
    """Grid evaluation of the EP log marginal likelihood over Cartesian hyper-parameter ranges (:18-40), leaves only,
    keyed by the evaluated setting."""

    def __init__(self, likelihoodEvaluator):
        self.likelihoodEvaluator = likelihoodEvaluator

    def evaluate(self, hyperParamsRanges, trainData, targets):
        settings = [np.array(t, dtype=np.float64) for t in itertools.product(*[list(r) for r in hyperParamsRanges])]
        ev = self.likelihoodEvaluator
        stop = ev.stopCriterion
        if isinstance(ev.kernelFunc, GaussianRbfKernel) and isinstance(stop, (AvgBasedStopCriterion, FixedSweepsStopCriterion)):
            # one C-ABI call for the whole grid: Gram, EP sweeps and LML of every setting stay on the GPU
            eps, cap = (stop.eps, 1000) if isinstance(stop, AvgBasedStopCriterion) else (-1.0, stop.sweeps)
            X = np.asfortranarray(np.asarray(trainData, dtype=np.float64))
            values, _, info = default_context().ep_lml_rbf_batched(X, targets, np.stack(settings), stop_eps=eps, max_sweeps=cap,
                                                                  strict=ev.strict)
            if np.any(info != 0):
                from ..._lib import NotPositiveDefinite
                b = int(np.flatnonzero(info)[0])
                raise NotPositiveDefinite(2, "setting %d: I + S^1/2 K S^1/2 not positive definite at pivot %d" % (b, info[b]), int(info[b]))
            return settings, values
        values = [ev.logLikelihoodWithoutGrad(trainData, targets, th) for th in settings]
        return settings, np.array(values)
