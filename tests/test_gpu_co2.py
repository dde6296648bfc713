"""The composite CO2 kernel (gp/regression/Co2Prediction.scala:29-137) on the device through the C-ABI
(gp_*_co2) and through the mirror of GpPredictor wired with `co2Kernel` (config/spring-context.xml:29-31,49-51), against the
oracle's restatement of Co2Kernel.apply / derAfterHyperParam and the reference pipeline built from it (Gram by pair loops,
unblocked Cholesky, substitution solves).  Fixture: the reference's own data file co2/maunaLoa.txt and the head of its stored
result dump co2/co2PredResults.txt (written after an L-BFGS run whose final hyper-parameters were only printed: loose sanity)."""
import os

import numpy as np
import pytest

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "co2")
HP0 = np.array([60., 70., 8., 50., 2., 0.34, 2.4, 0.88, 0.26, 0.2, 0.19])      # TestingUtils.co2HyperParamsVec (utils/TestingUtils.scala:17-20)


@pytest.fixture(scope="module")
def ctx():
    from gp_algos_amd.core import Context
    c = Context(0)
    yield c
    c.close()


def _mauna(ratio=0.7):
    from gp_algos_amd.gp.regression.co2_prediction import co2DataToYearWithValue, loadInput
    return co2DataToYearWithValue(loadInput(os.path.join(GOLD, "maunaLoa.txt")), ratio)


def test_co2_gram_matrices_vs_oracle(ctx):
    x = np.concatenate([np.linspace(1958.2, 1990.0, 141), [1990.0, 1971.25]])        # includes a duplicate and an unsorted point
    xs = np.linspace(1957.0, 2001.5, 37)
    K, Ko = ctx.gram_co2(x, HP0), orc.co2_gram(x, HP0)
    assert np.max(np.abs(K - Ko) / np.abs(Ko)) <= 1e-13 and np.array_equal(K, K.T)
    Ks, Kso = ctx.gram_co2(x, HP0, xs=xs), orc.co2_gram(x, HP0, xs=xs)
    assert Ks.shape == (37, 143) and np.max(np.abs(Ks - Kso) / np.abs(Kso)) <= 1e-13
    for pos in range(1, 12):
        D, Do = ctx.gram_co2(x, HP0, pos=pos), orc.co2_gram(x, HP0, pos=pos)
        assert np.max(np.abs(D - Do)) <= 1e-12 * max(np.max(np.abs(Do)), 1e-300), pos
    with pytest.raises(IndexError):
        ctx.gram_co2(x, HP0, pos=12)             # MatchError in the Scala match
    with pytest.raises(ValueError):
        ctx.gram_co2(x, HP0[:-1])


def test_co2_fit_predict_lml_vs_reference_pipeline(ctx):
    from gp_algos_amd.core import RegressionModel
    train, test = _mauna(0.35)                    # 212 training points
    x, y, xs = train[:, 0], train[:, 1], test[:40, 0]
    mdl = RegressionModel(ctx, x, y, HP0, kernel="co2")
    K = orc.co2_gram(x, HP0)
    Lo = orc.cholesky_lower(K)
    ao = orc.back_solve(Lo, orc.forward_solve(Lo, y), trans=True)
    assert np.linalg.norm(mdl.L() @ mdl.L().T - K) / np.linalg.norm(K) <= 1e-13
    assert np.max(np.abs(mdl.alpha() - ao)) <= 1e-7 * np.max(np.abs(ao))
    olml = orc.lml(Lo, ao, y)
    assert abs(mdl.lml() - olml) <= 1e-10 * abs(olml)
    Ks, Kss = orc.co2_gram(x, HP0, xs=xs), orc.co2_gram(xs, HP0)
    V = orc.forward_solve(Lo, np.asfortranarray(Ks.T))
    mean, var, cov = mdl.predict(np.asfortranarray(xs[:, None]), full_cov=True)
    assert np.max(np.abs(mean - Ks @ ao)) <= 1e-8 * np.max(np.abs(Ks @ ao))
    assert np.max(np.abs(cov - (Kss - V.T @ V))) <= 1e-8 * np.max(np.abs(Kss))
    assert np.max(np.abs(var - np.diag(cov))) <= 1e-9 * np.max(np.abs(Kss))
    mdl.close()


def test_co2_lml_gradient_vs_reference_formula_and_differences(ctx):
    train, _ = _mauna(0.25)                       # 151 points
    x, y = train[:, 0], train[:, 1]
    thetas = np.stack([HP0, HP0 * np.array([1.1, 0.9, 1.2, 1.0, 0.8, 1.3, 1.0, 1.1, 0.9, 1.2, 1.4])])
    lml, grad, info = ctx.lml_grad_co2_batched(x, y, thetas)
    assert np.all(info == 0) and grad.shape == (2, 11)
    for b in range(2):
        K = orc.co2_gram(x, thetas[b])
        Lo = orc.cholesky_lower(K)
        ao = orc.back_solve(Lo, orc.forward_solve(Lo, y), trans=True)
        Li = orc.inv_triangular(Lo, False)
        W = np.outer(ao, ao) - Li.T @ Li          # alphaSq - inversedK, GpPredictor.scala:66-69
        og = np.array([0.5 * np.trace(W @ orc.co2_gram(x, thetas[b], pos=p)) for p in range(1, 12)])
        assert abs(lml[b] - orc.lml(Lo, ao, y)) <= 1e-10 * abs(lml[b])
        assert np.max(np.abs(grad[b] - og)) <= 1e-7 * np.max(np.abs(og)), b
    l10, g10, _ = ctx.lml_grad_co2_batched(x, y, thetas[:1], nparams=10)      # optimizeNoise = false drops hp11
    assert g10.shape == (1, 10) and np.allclose(g10[0], grad[0, :10], rtol=1e-12)
    for k in (1, 4, 7, 10):
        h = 1e-5 * HP0[k]
        tp, tm = HP0.copy(), HP0.copy()
        tp[k] += h
        tm[k] -= h
        (lp, lm_), _, _ = ctx.lml_grad_co2_batched(x, y, np.stack([tp, tm]), nparams=0)
        assert abs((lp - lm_) / (2 * h) - grad[0, k]) <= 1e-4 * max(1.0, abs(grad[0, k]))
    bad = HP0.copy()
    bad[[0, 2, 5, 8, 10]] = 0.0                   # a zero kernel: not positive definite at the first pivot
    lb, gb, ib = ctx.lml_grad_co2_batched(x, y, np.stack([HP0, bad]))
    assert ib[0] == 0 and ib[1] == 1 and np.isnan(lb[1]) and np.isfinite(lb[0])


def test_co2_predictor_mirror_on_mauna_loa(ctx):
    """Co2Prediction.main / MasterThesisRelatedTasks.evaluateGpPredictionOnCo2Ds through the mirror: 70 % of Mauna Loa as training
    data, hyper-parameters fitted from co2HyperParamsVec, prediction over the WHOLE series; the reference's stored dump of exactly
    this experiment (co2/co2PredResults.txt: x, mean, stddev) is reproduced to plotting accuracy on its first rows."""
    import gp_algos_amd
    from gp_algos_amd.gp.regression.co2_prediction import Co2HyperParams, Co2Kernel
    from gp_algos_amd.gp.regression.gp_predictor import GpPredictor, PredictionInput
    from gp_algos_amd.utils.io_utilities import readVectorsFile
    gp_algos_amd.set_default_context(ctx)
    train, test = _mauna(0.7)
    whole = np.vstack([train, test])
    pred = GpPredictor(Co2Kernel(Co2HyperParams(HP0)))
    inp = PredictionInput(trainingData=train[:, :1], testData=whole[:, :1], sigmaNoise=None, targets=train[:, 1])
    dist0, ll0 = pred.predict(inp)
    dist, ll, hp = pred.predictWithParamsOptimization(inp, True)
    assert ll >= ll0 and hp.toDenseVector().shape == (11,)
    ref = readVectorsFile(os.path.join(GOLD, "co2PredResults_head.txt"))      # first 60 rows of the reference's dump
    assert np.allclose(ref[:, 0], whole[:60, 0], atol=1e-9)
    assert np.max(np.abs(dist.mean[:60] - ref[:, 1])) <= 0.5                  # ppm; the dump's own stddev there is ~0.23
    assert np.max(np.abs(dist0.mean[:60] - ref[:, 1])) <= 0.5
    sd = np.sqrt(np.diag(dist.sigma))
    assert np.all(sd[:424] < 1.0) and sd[-1] > sd[0]                           # uncertainty grows into the forecast
    # training-set fit and the 15-year forecast are close to the measurements
    assert np.sqrt(np.mean((dist.mean[:424] - train[:, 1]) ** 2)) <= 0.5
    assert np.sqrt(np.mean((dist.mean[424:] - test[:, 1]) ** 2)) <= 10.0      # the measured rise accelerates after 1993
    # logLikelihoodWithDerivatives and computePosterior take the device routes for this kernel too
    l1, g1 = pred.logLikelihoodWithDerivatives(inp.toPredictionTrainingInput(), Co2HyperParams(HP0), 11)
    assert abs(l1 - ll0) <= 1e-10 * abs(ll0) and g1.shape == (11,)
    Lm, alpha, _ = pred.preComputeComponents(train[:, :1], None, train[:, 1])
    post, V = pred.computePosterior(train[:, :1], whole[420:430, :1], Lm, alpha)
    assert np.max(np.abs(post.mean - dist0.mean[420:430])) <= 1e-7 * np.max(np.abs(post.mean)) and V.shape == (424, 10)
