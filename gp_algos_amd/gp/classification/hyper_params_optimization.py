"""Mirror of gp/classification/HyperParamsOptimization.scala:31-136: hyper-parameter fitting of the EP classifier by maximising
the EP log marginal likelihood.

GradientHyperParamsOptimizer(marginalLikelihoodEvaluator, gradOptimizer) keeps the reference's constructor and
optimizeHyperParams(ClassifierInput); with a GaussianRbfKernel, the BreezeLbfgsOptimizer the reference wires in and an
AvgBasedStopCriterion / fixed-sweep criterion the whole optimisation runs natively (gp_ep_optimize_rbf: device Gram, EP runs,
gradient, L-BFGS with concurrent line-search trials); otherwise the objective is evaluated through
MarginalLikelihoodEvaluator.logLikelihood per point, as in the Scala.  ApacheCommonsOptimizer (commons-math's Polak-Ribiere CG,
third party) maps to the same objective driven by scipy's CG with its limits (MaxIter 10, MaxEval 20, 5 iterations)."""
import numpy as np

from ... import default_context
from ...optimization.optimization import BreezeLbfgsOptimizer
from ...utils.kernel_requisites import GaussianRbfKernel
from .ep_parameter_estimator import AvgBasedStopCriterion, FixedSweepsStopCriterion


class HyperParameterOptimizer:
    def optimizeHyperParams(self, optimizationInput):
        raise NotImplementedError


class GradientHyperParamsOptimizer(HyperParameterOptimizer):   # :31-55
    def __init__(self, marginalLikelihoodEvaluator, gradOptimizer):
        self.marginalLikelihoodEvaluator, self.gradOptimizer = marginalLikelihoodEvaluator, gradOptimizer

    def optimizeHyperParams(self, optimizationInput):
        ev = self.marginalLikelihoodEvaluator
        X = np.asfortranarray(np.asarray(optimizationInput.trainData, dtype=np.float64))
        targets = optimizationInput.targets
        init = optimizationInput.initHyperParams
        theta0 = init.toDenseVector()
        stop = ev.stopCriterion
        native = (isinstance(ev.kernelFunc, GaussianRbfKernel) and isinstance(self.gradOptimizer, BreezeLbfgsOptimizer)
                  and isinstance(stop, (AvgBasedStopCriterion, FixedSweepsStopCriterion)))
        if native:
            eps, cap = (stop.eps, 1000) if isinstance(stop, AvgBasedStopCriterion) else (-1.0, stop.sweeps)
            best, _, _, _ = default_context().ep_optimize_rbf(X, targets, theta0, stop_eps=eps, max_sweeps=cap, strict=ev.strict,
                                                              max_iter=self.gradOptimizer.maxIter, history=4)
            return init.fromDenseVector(best)

        def funcWithGradient(hyperParams):                     # :38-46
            logLikelihood, derivatives = ev.logLikelihood(X, targets, np.asarray(hyperParams, dtype=np.float64))
            assert len(hyperParams) == len(derivatives)
            return logLikelihood, np.asarray(derivatives)

        return init.fromDenseVector(self.gradOptimizer.maximize(funcWithGradient, theta0))


class ApacheCommonsOptimizer(HyperParameterOptimizer):   # :57-136
    def __init__(self, marginalLikelihoodEvaluator):
        self.marginalLikelihoodEvaluator = marginalLikelihoodEvaluator

    def optimizeHyperParams(self, optimizationInput):
        from scipy.optimize import minimize
        ev = self.marginalLikelihoodEvaluator
        X = np.asfortranarray(np.asarray(optimizationInput.trainData, dtype=np.float64))
        cache = {}

        def value_and_grad(hp):                                # pointGradientMapping :66-110: one EP run per distinct point
            key = tuple(np.asarray(hp, dtype=np.float64))
            if key not in cache:
                cache[key] = ev.logLikelihood(X, optimizationInput.targets, np.asarray(hp, dtype=np.float64))
            ll, g = cache[key]
            return -ll, -np.asarray(g)

        init = optimizationInput.initHyperParams
        res = minimize(value_and_grad, init.toDenseVector(), jac=True, method="CG", options={"maxiter": 5})   # GoalType.MAXIMIZE, 5 iterations :112-121
        return init.fromDenseVector(res.x)
