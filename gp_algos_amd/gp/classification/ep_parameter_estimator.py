"""Mirror of gp/classification/EpParameterEstimator.scala on top of libgpcore.so."""
from dataclasses import dataclass
from typing import Optional

import numpy as np

from ... import default_context
from ..._lib import GP_EP_GET_L
from ...core import EpClassifierState


@dataclass
class SiteParams:   # :181-182
    tauSiteParams: np.ndarray
    niSiteParams: np.ndarray
    marginalLogLikelihood: Optional[float] = None


@dataclass
class CavityDistributionParams:   # :183
    tauParams: np.ndarray
    niParams: np.ndarray


@dataclass
class EpEstimationContext:   # :185
    oldParams: SiteParams
    currentParams: SiteParams


def avgBetweenSiteParams(oldParams, currentParams):   # :195-202 -- (sum / 2) * n, precedence as written
    s = 0.0
    for i in range(len(oldParams.niSiteParams)):
        s = s + (currentParams.niSiteParams[i] - oldParams.niSiteParams[i]) + \
            (currentParams.tauSiteParams[i] - oldParams.tauSiteParams[i])
    return s / 2 * len(currentParams.niSiteParams)


class AvgBasedStopCriterion:   # :187-193
    def __init__(self, eps):
        self.eps = float(eps)

    def __call__(self, context):
        return bool(abs(avgBetweenSiteParams(context.oldParams, context.currentParams)) < self.eps)


class FixedSweepsStopCriterion:
    """Stops after a fixed number of sweeps (50 sweeps)."""

    def __init__(self, sweeps):
        self.sweeps, self._seen = int(sweeps), 0

    def __call__(self, context):
        self._seen += 1
        return self._seen >= self.sweeps


class EpParameterEstimator:
    """class EpParameterEstimator(kernelMatrix, targets: DenseVector[Int], stopCriterion)  (:11-12)."""

    def __init__(self, kernelMatrix, targets, stopCriterion, strict=True, max_sweeps=1000):
        self.kernelMatrix = np.asfortranarray(np.asarray(kernelMatrix, dtype=np.float64))
        self.targets = np.asarray(targets, dtype=np.int32).reshape(-1)
        if self.kernelMatrix.shape[0] != self.targets.size:   # require(kernelMatrix.rows == targets.length) :20
            raise ValueError("requirement failed")
        self.stopCriterion = stopCriterion
        self.strict = strict          # True: EP LML as compiled (term at :92 dropped); False: intended formula
        self.max_sweeps = max_sweeps

    def estimateSiteParams(self, keep_state=False):
        """-> (SiteParams(tau, ni, Some(lml)), lowerTriangular).  Sweep 0 always runs; before every later sweep the
        stop criterion sees (old, current) exactly like the Stream.takeWhile at :40."""
        n = self.targets.size
        st = EpClassifierState(default_context(), self.kernelMatrix, self.targets)
        cur = SiteParams(np.zeros(n), np.zeros(n))
        old = SiteParams(np.zeros(n), np.zeros(n))
        j = 0
        while j == 0 or (j < self.max_sweeps and not self.stopCriterion(EpEstimationContext(oldParams=old, currentParams=cur))):
            old = SiteParams(cur.tauSiteParams.copy(), cur.niSiteParams.copy())
            tau, nu = st.sweep(1)
            cur = SiteParams(tau, nu)
            j += 1
        lml = st.lml(strict=self.strict)
        L = st.get(GP_EP_GET_L)
        out = SiteParams(cur.tauSiteParams, cur.niSiteParams, lml), L
        if keep_state:
            return out, st
        st.close()
        return out
