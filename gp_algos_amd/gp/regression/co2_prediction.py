"""Mirror of gp/regression/Co2Prediction.scala: the composite CO2 kernel (a user-defined KernelFunc with 11 hyper-parameters on
one-dimensional inputs), its hyper-parameter record, and the Mauna Loa file helpers.  The kernel's matrices, the fit, the
predictions, the log marginal likelihood with its 11 derivatives and the hyper-parameter fit all run on the device
(gp_*_co2); the scalar `apply` / `derAfterHyperParam` below are host conveniences with the reference's arithmetic order."""
import math
import re

import numpy as np

from ...utils.kernel_requisites import KernelFunc, KernelFuncHyperParams, MatchError


class Co2HyperParams(KernelFuncHyperParams):   # :18-27
    def __init__(self, dv):
        self.dv = np.array(dv, dtype=np.float64).reshape(-1)

    def fromDenseVector(self, dv):
        return Co2HyperParams(dv)

    def toDenseVector(self):
        return self.dv.copy()

    def getAtPosition(self, i):   # 1-based
        if i < 1 or i > self.dv.size:
            raise IndexError("index %d out of bounds" % (i - 1))
        return float(self.dv[i - 1])

    def __repr__(self):
        return "Co2HyperParams(%s)" % self.dv


class Co2Kernel(KernelFunc):   # :29-137
    def __init__(self, co2HyperParams):
        self.co2HyperParams = co2HyperParams

    def _hp(self):
        return tuple(self.co2HyperParams.getAtPosition(i) for i in range(1, 12))

    def apply(self, obj1, obj2, sameIndex):   # :39-56
        obj1, obj2 = np.atleast_1d(obj1), np.atleast_1d(obj2)
        if not (obj1.size == 1 and obj2.size == 1):
            raise ValueError("requirement failed: This kernel is applicable only for 1D objects")
        hp1, hp2, hp3, hp4, hp5, hp6, hp7, hp8, hp9, hp10, hp11 = self._hp()
        x1, x2 = float(obj1[0]), float(obj2[0])
        xDiff, xDiffSq = x1 - x2, (x1 - x2) * (x1 - x2)
        k1Val = hp1 * hp1 * math.exp(-xDiffSq / (2 * hp2 * hp2))
        sinVal = math.sin(math.pi * xDiff)
        k2Val = hp3 * hp3 * math.exp((-xDiffSq / (2 * hp4 * hp4)) - 2 * sinVal * sinVal / (hp5 * hp5))
        k3Pow1 = 1 + xDiffSq / (2 * hp8 * hp7 * hp7)
        k3Val = hp6 * hp6 * math.pow(k3Pow1, -hp8)
        k4Val = hp9 * hp9 * math.exp(-xDiffSq / (2 * hp10 * hp10))
        indNoise = hp11 * hp11 if sameIndex else 0.0
        return k1Val + k2Val + k3Val + k4Val + indNoise

    @property
    def hyperParams(self):
        return self.co2HyperParams

    def changeHyperParams(self, dv):
        return Co2Kernel(Co2HyperParams(dv))

    def gradient(self, afterFirstArg):   # ??? in the reference (:62-64)
        raise NotImplementedError("an implementation is missing")

    gradientAt = gradient

    @property
    def hyperParametersNum(self):
        return 11


def _co2_der(hp, x1, x2, sameIndex, num):
    """derAfterFirstKernel .. derAfterFourthKernel (:92-137) in the reference's operation order."""
    hp1, hp2, hp3, hp4, hp5, hp6, hp7, hp8, hp9, hp10, hp11 = hp
    xDiff, sqDiff = x1 - x2, (x1 - x2) * (x1 - x2)
    if num < 3:
        if num == 1:
            return 2 * hp1 * math.exp(-sqDiff / (2 * hp2 * hp2))
        return hp1 * hp1 * math.exp(-sqDiff / (2 * hp2 * hp2)) * sqDiff * math.pow(hp2, -3)
    if num < 6:
        sinVal = math.sin(math.pi * xDiff)
        k2Val = hp3 * hp3 * math.exp(-sqDiff / (2 * hp4 * hp4) - 2 * sinVal * sinVal / (hp5 * hp5))
        return 2 * k2Val / hp3 if num == 3 else (k2Val * sqDiff * math.pow(hp4, -3) if num == 4 else k2Val * 4 * sinVal * sinVal * math.pow(hp5, -3))
    if num < 9:
        k3Pow1 = 1 + sqDiff / (2 * hp8 * hp7 * hp7)
        if num == 6:
            return 2 * hp6 * math.pow(k3Pow1, -hp8)
        if num == 7:
            return hp6 * hp6 * math.pow(k3Pow1, -hp8 - 1) * sqDiff * math.pow(hp7, -3)
        firstTerm = math.exp(-hp8 * math.log(k3Pow1))
        secondTerm = -math.log(k3Pow1) + (hp8 * sqDiff / (2 * hp7 * hp7 * hp8 * hp8 * k3Pow1))
        return hp6 * hp6 * firstTerm * secondTerm
    k4Val = hp9 * hp9 * math.exp(-sqDiff / (2 * hp10 * hp10))
    if num == 9:
        return 2 * k4Val / hp9
    if num == 10:
        return k4Val * sqDiff * math.pow(hp10, -3)
    return 2 * hp11 if sameIndex else 0.0


def _derAfterHyperParam(self, paramNum):   # :66-83, paramNum 1-based; past 11 the Scala match throws MatchError
    if paramNum < 1 or paramNum > 11:
        raise MatchError(paramNum)
    hp = self._hp()

    def f(vec1, vec2, sameIndex):
        return _co2_der(hp, float(np.atleast_1d(vec1)[0]), float(np.atleast_1d(vec2)[0]), sameIndex, paramNum)
    f._gpcore_co2 = (self, paramNum)      # lets MatrixUtils.buildMatrixWithFunc build dK/dhp on the device
    return f


Co2Kernel.derAfterHyperParam = _derAfterHyperParam


# ---- the Mauna Loa file helpers ---------------------------------------------------------------------------------------------
def loadInput(fileName):   # :139-152: whitespace / tab separated numbers, one row per line
    with open(fileName) as fh:
        lines = [ln for ln in fh.read().splitlines()]
    assert len(lines) > 1
    split = re.compile(r"(?:\s+|\t+)")
    colNum = len(split.split(lines[0]))
    out = np.zeros((len(lines), colNum), order="F")
    for i, ln in enumerate(lines):
        out[i, :] = [float(t) for t in split.split(ln)]
    return out


def co2DataToYearWithValue(matrix, trainTestRatio):   # :155-183: (train, test), rows = (year + (month - 1)/12, ppm), ppm > 0 only
    if not 0 <= trainTestRatio <= 1:
        raise ValueError("requirement failed: Division's ratio should be between 0 and 1")
    rows = []
    for r in range(matrix.shape[0]):
        year = matrix[r, 0]
        for month in range(1, matrix.shape[1] - 1):
            ppm = matrix[r, month]
            if ppm > 0:
                rows.append((year + (1 / 12.0) * (month - 1), ppm))
    whole = np.asfortranarray(np.array(rows, dtype=np.float64))
    trainNum = int(whole.shape[0] * trainTestRatio)
    return whole[:trainNum, :], whole[trainNum:, :]


def predictionComparisonToString(testData, posterior, targets):   # :185-196
    out = []
    for i in range(posterior.dim):
        out.append("%d - %s, predicted: %s +- %s, true value: %s" % (i, np.asarray(testData)[i, :], posterior.mean[i],
                                                                      2 * math.sqrt(posterior.sigma[i, i]), targets[i]))
    return "\n".join(out) + "\n"
