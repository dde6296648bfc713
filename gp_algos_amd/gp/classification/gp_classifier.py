"""Mirror of gp/classification/GpClassifier.scala on top of libgpcore.so."""
from dataclasses import dataclass
from typing import Any, Optional, Tuple

import numpy as np

from ... import default_context
from ...core import EpClassifierState
from .ep_parameter_estimator import EpParameterEstimator, SiteParams


@dataclass
class ClassifierInput:   # :63-65
    trainKernelMatrix: np.ndarray
    targets: np.ndarray
    initHyperParams: Any = None
    trainData: Optional[np.ndarray] = None


@dataclass
class AfterEstimationClassifierInput:   # :58-61
    targets: np.ndarray
    learnParams: Optional[Tuple[SiteParams, np.ndarray]]
    hyperParams: Any
    trainKernelMatrix: np.ndarray
    testTrainKernelMatrix: np.ndarray
    testKernelMatrix: np.ndarray


class GpClassifier:
    def __init__(self, stopCriterion):
        self.stopCriterion = stopCriterion

    def trainClassifier(self, classInput):   # :18-22 ; targets must contain values from set {-1,1}
        return EpParameterEstimator(classInput.trainKernelMatrix, classInput.targets, self.stopCriterion).estimateSiteParams()

    def classify(self, input):   # :24-47 -> probabilities of class 1
        K = np.asfortranarray(np.asarray(input.trainKernelMatrix, dtype=np.float64))
        Ks = np.asfortranarray(np.asarray(input.testTrainKernelMatrix, dtype=np.float64))
        kss = np.ascontiguousarray(np.diag(np.asarray(input.testKernelMatrix, dtype=np.float64)))
        if input.learnParams is None:
            (site, _), st = EpParameterEstimator(K, input.targets, self.stopCriterion).estimateSiteParams(keep_state=True)
        else:
            # site parameters are given: rebuild the device state (L, Sigma) from them with one refactorisation
            site, _ = input.learnParams
            st = EpClassifierState(default_context(), K, input.targets)
            st.load_site_params(site.tauSiteParams, site.niSiteParams)
        try:
            return st.predict(Ks, kss)
        finally:
            st.close()
