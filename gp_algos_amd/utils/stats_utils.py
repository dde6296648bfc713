"""Mirror of the hot-path part of utils/StatsUtils.scala (:13-25)."""
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
