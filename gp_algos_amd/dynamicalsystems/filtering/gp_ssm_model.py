"""The Gaussian-process state-space model of dynamicalsystems/filtering/GPUnscentedKalmanFilter.scala:63-147 on the device.

The GP-UKF learns one GP per hidden-state dimension for the transition (on the state DIFFERENCES, :98-108) and one per
observation dimension (:110-114), all over the same training inputs, and then calls GpPredictor.computePosterior with ONE test
point for every sigma point, dimension and time step (:72-87, :138-147).  Here each family of GPs is ONE resident batch
(gp_small_fit) and a whole sigma-point set goes through ONE launch per family (gp_small_posterior): 2 D + 1 points x D (or O)
models.  The unscented filter recursion itself (UnscentedKalmanFilter.scala, state dimensions 1-4) is host code that calls
these functions; it is not part of the hot path."""
import numpy as np

from ... import default_context
from ...core import SmallModelBatch
from ...utils.kernel_requisites import GaussianRbfKernel


class GpSsmModel:
    def __init__(self, systemBatch, obsBatch):
        self.systemBatch, self.obsBatch = systemBatch, obsBatch

    @classmethod
    def learn(cls, gpPredictor, observations, trueHiddenStates, optimizeGpLearning=False):
        """learnNewSsmModelWithNoises (:63-96): observations is O x T, trueHiddenStates D x T (one column per time step)."""
        kf = gpPredictor.kernelFunc
        if not isinstance(kf, GaussianRbfKernel):
            raise NotImplementedError("the batched posterior is implemented for GaussianRbfKernel")
        H = np.asarray(trueHiddenStates, dtype=np.float64)
        O = np.asarray(observations, dtype=np.float64)
        D, T = H.shape
        # learnSystemFunction :98-108: inputs = hidden states 0..T-2 (as rows), targets = differences to the next state
        sysX = np.asfortranarray(H[:, :-1].T)
        sysY = np.asfortranarray((H[:, 1:] - H[:, :-1]).T)
        # learnObsFunction :110-114: inputs = all hidden states, targets = observations
        obsX = np.asfortranarray(H.T)
        obsY = np.asfortranarray(O.T)

        def thetas(X, Y):
            base = kf.hyperParams.toDenseVector()
            if not optimizeGpLearning:                       # learnInputOutput :116-129, None -> the predictor's own kernel
                return np.tile(base, (Y.shape[1], 1))
            return np.stack([gpPredictor.obtainOptimalHyperParams(X, None, Y[:, g], True).toDenseVector() for g in range(Y.shape[1])])

        ctx = default_context()
        return cls(SmallModelBatch(ctx, sysX, thetas(sysX, sysY), Y=sysY), SmallModelBatch(ctx, obsX, thetas(obsX, obsY), Y=obsY))

    def close(self):
        self.systemBatch.close()
        self.obsBatch.close()

    # transitionFuncImpl :72-80 for a whole sigma-point set: rows of `prevHiddenStates` are points; returns the same shape
    def transitionFuncImpl(self, prevHiddenStates):
        P = np.atleast_2d(np.asarray(prevHiddenStates, dtype=np.float64))
        mean, _ = self.systemBatch.posterior(P)              # (D, m): mean(0) of every dimension's GP at every point
        return P + mean.T

    # observationFuncImpl :81-87
    def observationFuncImpl(self, hiddenStates):
        P = np.atleast_2d(np.asarray(hiddenStates, dtype=np.float64))
        mean, _ = self.obsBatch.posterior(P)
        return mean.T

    # computeNoiseMatrix :138-147: diag of sigma(0,0) of every dimension's GP at ONE test point
    def qNoise(self, hiddenMean):
        _, var = self.systemBatch.posterior(np.asarray(hiddenMean, dtype=np.float64).reshape(1, -1))
        return np.diag(var[:, 0])

    def rNoise(self, hiddenMean):
        _, var = self.obsBatch.posterior(np.asarray(hiddenMean, dtype=np.float64).reshape(1, -1))
        return np.diag(var[:, 0])
