"""Mirror of optimization/Optimization.scala:30-63.  Breeze's LBFGS (m = 4) is third-party code that is
not part of the reference tree; scipy's L-BFGS-B with the same memory and iteration cap stands in for it,
so trajectories differ -- parity is defined on objective values/gradients at given points."""
import numpy as np
from scipy.optimize import minimize as _sp_minimize


class GradientBasedOptimizer:
    def minimize(self, func, initPoint):
        raise NotImplementedError

    def maximize(self, func, initPoint):
        raise NotImplementedError


class BreezeLbfgsOptimizer(GradientBasedOptimizer):
    def __init__(self, maxIter=10):   # def this() = this(10)
        self.maxIter = int(maxIter)

    def minimize(self, func, initPoint):
        best = {"x": np.array(initPoint, dtype=np.float64), "v": float("inf")}

        def wrapped(x):
            value, grad = func(np.array(x, dtype=np.float64))
            if value < best["v"]:          # tracks the best point seen  (:44-46)
                best["x"], best["v"] = np.array(x, dtype=np.float64), float(value)
            return float(value), np.asarray(grad, dtype=np.float64)

        res = _sp_minimize(wrapped, np.array(initPoint, dtype=np.float64), jac=True, method="L-BFGS-B",
                           options={"maxiter": self.maxIter, "maxcor": 4})
        optimal_val = func(res.x)[0]
        return np.array(res.x) if optimal_val < best["v"] else best["x"]   # :52-55

    def maximize(self, func, initPoint):   # :58-61
        def minus(point):
            value, grad = func(point)
            return -value, -np.asarray(grad, dtype=np.float64)
        return self.minimize(minus, initPoint)
