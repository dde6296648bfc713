"""Mirror of the hot-path part of utils/StatsUtils.scala (:13-25) and of meanAndVarOfData (:59-71), which GP-UCB uses to place
its L-BFGS starting points."""
import math
from dataclasses import dataclass

import numpy as np


def dnorm(x):
    return math.exp(-(x * x) / 2.0 - math.log(math.sqrt(2.0 * math.pi)))


def pnorm(x):
    return 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))


@dataclass
class GaussianDistribution:
    mean: np.ndarray
    sigma: np.ndarray

    @property
    def dim(self):
        return int(np.asarray(self.mean).size)


def meanAndVarOfData(data):
    """/*data(i,::) - ith sample*/  StatsUtils.scala:59-71: (mean, covariance with the 1/N normalisation)."""
    data = np.asarray(data, dtype=np.float64)
    n = data.shape[0]
    mean = np.zeros(data.shape[1])
    for i in range(n):
        mean = mean + data[i, :]
    mean = mean / float(n)
    cov = np.zeros((data.shape[1], data.shape[1]))
    for i in range(n):
        diff = data[i, :] - mean
        cov = cov + np.outer(diff, diff)
    return mean, cov / float(n)
