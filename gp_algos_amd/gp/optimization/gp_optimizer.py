"""Mirror of gp/optimization/GPOptimizer.scala (GP-UCB Bayesian optimisation) on top of the batched small-n entry points.

class GPOptimizer(gpPredictor, noise, gradientOptimizer) keeps the reference's constructor, `maximize` / `minimize` /
`prepareGrid` / `evaluateGridPoints` and GPOInput; what changes is how an iteration runs on the device:
  * the model is fitted ONCE (gp_small_fit) and every chosen point is APPENDED by a rank-1 extension of L, L^-1 and alpha
    (gp_small_append) where the reference refits from scratch (GPOptimizer.scala:51,64-67);
  * the c L-BFGS runs of maximizeUCB (:55-63,82-109) advance in lockstep, every iteration's trial points in one launch
    (gp_small_maximize_ucb); the objective mean + k sqrt(var) and its gradient through GaussianRbfKernel.gradient are evaluated
    on the device.
The random draws (initial grid, L-BFGS starting points) come from numpy instead of scala.util.Random(System.nanoTime()) /
commons-math: the reference's trajectory is not reproducible by construction."""
from dataclasses import dataclass
from typing import Sequence

import numpy as np

from ... import default_context
from ...core import SmallModelBatch
from ...utils.kernel_requisites import GaussianRbfKernel
from ...utils.stats_utils import meanAndVarOfData


@dataclass
class GPOInput:   # GPOptimizer.scala:170-172
    ranges: Sequence[range]
    mParam: int
    cParam: int
    kParam: float
    optimizeHpOnInitGrid: bool = False


class GPOptimizer:
    def __init__(self, gpPredictor, noise=None, gradientOptimizer=None, seed=None):
        self.gpPredictor = gpPredictor
        self.noise = noise
        self.gradientOptimizer = gradientOptimizer      # kept for signature compatibility; the device runs the L-BFGS (m = 4)
        self.hyperParams = gpPredictor.kernelFunc.hyperParams
        self._rng = np.random.default_rng(seed)

    def minimize(self, objFunc, params):   # :29-33
        optimum, value = self.maximize(lambda point: -objFunc(point), params)
        return optimum, -value

    def maximize(self, func, params):   # :35-80
        ranges, m, c, k = params.ranges, params.mParam, params.cParam, params.kParam
        if not (c >= 1 and m >= 1):
            raise ValueError("requirement failed: Params m and c needs to be greater or equal 1")
        if not isinstance(self.gpPredictor.kernelFunc, GaussianRbfKernel):
            raise NotImplementedError("the device UCB gradient is implemented for GaussianRbfKernel")
        pointGrid = self.prepareGrid(ranges)
        evaluated = self.evaluateGridPoints(pointGrid, func)
        hp = self.hyperParams
        if params.optimizeHpOnInitGrid:
            hp = self.gpPredictor.obtainOptimalHyperParams(pointGrid, self.noise, evaluated, True)
        n0, d = pointGrid.shape
        if n0 + m > 2048:
            raise ValueError("3 d + m exceeds GP_SMALL_MAX_N")
        batch = SmallModelBatch(default_context(), pointGrid, hp.toDenseVector()[None, :], Y=evaluated[:, None],
                                sigma_noise=self.noise, capacity=n0 + m)
        points, values = [row.copy() for row in pointGrid], list(evaluated)
        try:
            for _ in range(m):
                P = np.asarray(points)
                mean, cov = meanAndVarOfData(P)                                   # :52-53
                starts = self._rng.multivariate_normal(mean, cov, size=c, method="svd").reshape(c, d)
                best_x, best_ucb, _ = batch.maximize_ucb(starts, k, max_iter=10, history=4)   # BreezeLbfgsOptimizer() default maxIter = 10
                if not np.isfinite(best_ucb):
                    best_x = self._rng.multivariate_normal(mean, cov, method="svd")          # :64
                try:
                    y = float(func(best_x))
                    batch.append(best_x, [y])                                     # :66-69 without the O(n^3) refit
                    points.append(np.array(best_x))
                    values.append(y)
                except Exception:                                                  # :70-72: a failed evaluation keeps the old sets
                    pass
        finally:
            batch.close()
        values = np.asarray(values)
        i = int(np.argmax(values))                                                 # first maximum, like the fold at :73-78
        return np.array(points[i]), float(values[i])

    # /*grid(i,::) - i'th d-dimensional point*/   :131-136
    def evaluateGridPoints(self, grid, func):
        return np.array([float(func(np.array(grid[i, :]))) for i in range(grid.shape[0])])

    def prepareGrid(self, ranges):   # :138-152: 3*dim uniform points inside the ranges
        dim = len(ranges)
        grid = np.zeros((3 * dim, dim), order="F")
        for i in range(grid.shape[0]):
            for j, r in enumerate(ranges):
                lo, hi = float(r[0]), float(r[-1])      # Scala's `-6 to 6`: Range.start / Range.end (inclusive); also (lo, hi) pairs
                if not lo < hi:
                    raise ValueError("requirement failed")
                grid[i, j] = lo + (hi - lo) * self._rng.random()
        return grid
