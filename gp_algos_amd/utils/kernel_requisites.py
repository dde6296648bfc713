"""Mirror of utils/KernelRequisites.scala: kernel-function traits and the ARD Gaussian RBF kernel.

Scalar evaluations here are host-side conveniences with the reference's exact arithmetic order;
matrices are built on the GPU by utils.matrix_utils.buildKernelMatrix."""
import math

import numpy as np


class MatchError(IndexError):
    """scala.MatchError: getAtPosition past the last hyper-parameter (KernelRequisitesTest.scala:32-34)."""


class KernelFuncHyperParams:
    # Indexed from 1, not from 0 !!!   (KernelRequisites.scala:17)
    def getAtPosition(self, i):
        raise NotImplementedError

    def toDenseVector(self):
        raise NotImplementedError

    def fromDenseVector(self, dv):
        raise NotImplementedError


class KernelFunc:
    """trait KernelFunc extends AbstractKernelFunc[featureVector] (KernelRequisites.scala:23-36)."""

    def apply(self, obj1, obj2, sameIndex):
        raise NotImplementedError

    def __call__(self, obj1, obj2, a, b=None):
        # apply(obj1, obj2, index1, index2) = apply(obj1, obj2, index1 == index2)   (:24)
        return self.apply(obj1, obj2, a if b is None else a == b)

    @property
    def hyperParametersNum(self):
        raise NotImplementedError

    def derAfterHyperParam(self, paramNum):
        raise NotImplementedError

    def changeHyperParams(self, dv):
        raise NotImplementedError

    @property
    def hyperParams(self):
        raise NotImplementedError


class GaussianRbfParams(KernelFuncHyperParams):
    """case class GaussianRbfParams(signalVar, lengthScales, noiseVar)  (KernelRequisites.scala:39-60)."""

    def __init__(self, signalVar, lengthScales, noiseVar):
        self.signalVar = float(signalVar)
        self.lengthScales = np.array(lengthScales, dtype=np.float64).reshape(-1)
        self.noiseVar = float(noiseVar)

    def getAtPosition(self, i):
        d = self.lengthScales.size
        if i == 1:
            return self.signalVar
        if 1 < i < d + 2:
            return float(self.lengthScales[i - 2])
        if i == d + 2:
            return self.noiseVar
        raise MatchError(i)

    def toDenseVector(self):
        return np.array([self.getAtPosition(k + 1) for k in range(self.lengthScales.size + 2)])

    def fromDenseVector(self, dv):
        dv = np.asarray(dv, dtype=np.float64).reshape(-1)
        if dv.size != self.lengthScales.size + 2:   # require(...) :55
            raise ValueError("requirement failed: %d does not equal to %d" % (dv.size, self.lengthScales.size + 2))
        return GaussianRbfParams(dv[0], dv[1:-1], dv[-1])

    def copy(self, signalVar=None, lengthScales=None, noiseVar=None):
        return GaussianRbfParams(self.signalVar if signalVar is None else signalVar,
                                 self.lengthScales if lengthScales is None else lengthScales,
                                 self.noiseVar if noiseVar is None else noiseVar)

    def __eq__(self, o):
        return (isinstance(o, GaussianRbfParams) and self.signalVar == o.signalVar and self.noiseVar == o.noiseVar
                and np.array_equal(self.lengthScales, o.lengthScales))

    def __repr__(self):
        return "GaussianRbfParams(%r,%r,%r)" % (self.signalVar, self.lengthScales.tolist(), self.noiseVar)


class GaussianRbfKernel(KernelFunc):
    """k(x_p,x_q) = signalVar^2*exp(-0.5*(x_p-x_q)^T diag(lengthScales^-2) (x_p-x_q)) + noiseVar^2*(p == q)  (:61-114)."""

    def __init__(self, rbfParams):
        self.rbfParams = rbfParams

    def _ls_product(self, a, b):   # inputWithLsProduct :109-113
        acc = 0.0
        for k in range(self.rbfParams.lengthScales.size):
            diff = float(a[k]) - float(b[k])
            ls = float(self.rbfParams.lengthScales[k])
            acc = acc + (diff * (1.0 / (ls * ls))) * diff
        return acc

    def apply(self, obj1, obj2, sameIndex):   # :66-72
        sf, sn = self.rbfParams.signalVar, self.rbfParams.noiseVar
        v = sf * sf * math.exp(-0.5 * self._ls_product(obj1, obj2))
        return v + sn * sn if sameIndex else v

    @property
    def hyperParametersNum(self):
        return self.rbfParams.lengthScales.size + 2

    def derAfterHyperParam(self, paramNum):   # :76-86, paramNum is 1-based
        d = self.rbfParams.lengthScales.size
        sf, sn = self.rbfParams.signalVar, self.rbfParams.noiseVar

        def f(vec1, vec2, sameIndex):
            if paramNum == 1:
                return 2 * sf * math.exp(-0.5 * self._ls_product(vec1, vec2))
            if paramNum < d + 2:
                diff = float(vec1[paramNum - 2]) - float(vec2[paramNum - 2])
                return (math.pow(sf, 2) * math.exp(-0.5 * self._ls_product(vec1, vec2)) * math.pow(diff, 2)
                        * math.pow(float(self.rbfParams.lengthScales[paramNum - 2]), -3))
            if paramNum == d + 2:
                return 2 * sn if sameIndex else 0.0
            raise MatchError(paramNum)
        return f

    def changeHyperParams(self, dv):   # :88-91
        return GaussianRbfKernel(self.rbfParams.fromDenseVector(dv))

    @property
    def hyperParams(self):
        return self.rbfParams

    def gradient(self, afterFirstArg):   # :99-107
        def g(vec1, vec2):
            diff = np.asarray(vec1, dtype=np.float64) - np.asarray(vec2, dtype=np.float64)
            inv = 1.0 / (self.rbfParams.lengthScales * self.rbfParams.lengthScales)
            a1 = self.apply(vec1, vec2, False)
            return (diff * inv) * (-a1) if afterFirstArg else (diff * inv) * a1
        return g

    def gradientAt(self, afterFirstArg, points):
        return self.gradient(afterFirstArg)(points[0], points[1])


def testTrainKernelMatrix(test, train, kernelFun):   # :116-124
    from .matrix_utils import buildKernelMatrix
    return buildKernelMatrix(kernelFun, test, train)
