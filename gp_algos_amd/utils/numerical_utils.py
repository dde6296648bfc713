"""Mirror of utils/NumericalUtils.scala: the ~= helpers the reference's tests use (:10-20)."""
import numpy as np


class Precision:
    def __init__(self, p):
        self.p = float(p)


def approx_equal(a, b, precision):
    return bool(np.all(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) < precision.p))
