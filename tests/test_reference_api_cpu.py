"""CPU tests of the host-side mirror of the reference API (no GPU compute): they read like
src/test/scala/utils/KernelRequisitesTest.scala and check the scalar kernel arithmetic, the fixtures
and the oracle-generated golden vectors against the oracle itself."""
import json
import os

import numpy as np
import pytest

from gp_algos_amd.utils.kernel_requisites import GaussianRbfKernel, GaussianRbfParams, MatchError
from gp_algos_amd.utils import stats_utils
from gp_algos_amd.gp.classification.ep_parameter_estimator import (AvgBasedStopCriterion, EpEstimationContext, SiteParams,
                                                                    avgBetweenSiteParams)
from oracle import gp_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ls = np.ones(5)


def test_hyperparams_to_dense_vector():   # KernelRequisitesTest.scala:20-23
    hp = GaussianRbfParams(signalVar=1.0, lengthScales=ls, noiseVar=0.0)
    assert np.array_equal(hp.toDenseVector(), [1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0])


def test_hyperparams_get_at_position():   # :25-35, 1-based, MatchError past the end
    hp = GaussianRbfParams(signalVar=1.0, lengthScales=[5.0, 2.0, 3.0], noiseVar=0.0)
    assert [hp.getAtPosition(k) for k in (1, 2, 3, 4, 5)] == [1.0, 5.0, 2.0, 3.0, 0.0]
    with pytest.raises(MatchError):
        hp.getAtPosition(6)


def test_kernel_updates_its_params():   # :40-47
    k0 = GaussianRbfKernel(GaussianRbfParams(1.0, ls, 0.0))
    assert k0.rbfParams == GaussianRbfParams(1.0, ls, 0.0)
    k1 = k0.changeHyperParams(np.array([2.0, 3.0, 3.0, 3.0, 3.0, 3.0, 0.0]))
    assert k1.rbfParams == GaussianRbfParams(2.0, ls * 3.0, 0.0)
    with pytest.raises(ValueError):     # require(dv.length == lengthScales.length + 2), KernelRequisites.scala:55
        k0.changeHyperParams(np.array([1.0, 2.0]))
    assert k0.hyperParametersNum == 7


def test_scalar_kernel_and_derivatives_match_oracle():
    rng = np.random.default_rng(1)
    x, y = rng.normal(size=4), rng.normal(size=4)
    theta = np.array([-1.7, 0.8, 1.3, 2.1, 0.6, 0.25])
    k = GaussianRbfKernel(GaussianRbfParams(theta[0], theta[1:-1], theta[-1]))
    assert k.apply(x, y, False) == orc.rbf_kernel(x, y, theta, False)
    assert k.apply(x, x, True) == orc.rbf_kernel(x, x, theta, True) == theta[0] ** 2 + theta[-1] ** 2
    X = np.asfortranarray(np.stack([x, y]))
    for p in range(1, 7):
        D = orc.dgram_sym(X, theta, p)
        assert k.derAfterHyperParam(p)(y, x, False) == D[1, 0]
        assert k.derAfterHyperParam(p)(x, x, True) == D[0, 0]
    with pytest.raises(MatchError):
        k.derAfterHyperParam(7)(x, y, False)


def test_pnorm_dnorm_match_oracle():
    for z in (-6.0, -1.3, 0.0, 0.4, 2.5, 7.0):
        assert stats_utils.pnorm(z) == orc.pnorm(z)
        assert abs(stats_utils.dnorm(z) - orc.dnorm(z)) <= 1e-17 + 1e-15 * orc.dnorm(z)


def test_avg_based_stop_criterion_precedence():   # EpParameterEstimator.scala:187-202: (sum / 2) * n
    old = SiteParams(np.array([1.0, 2.0, 3.0]), np.array([0.5, 0.5, 0.5]))
    cur = SiteParams(np.array([1.5, 2.0, 2.0]), np.array([0.75, 0.5, 0.25]))
    assert avgBetweenSiteParams(old, cur) == orc.avg_between_site_params(old.tauSiteParams, old.niSiteParams, cur.tauSiteParams, cur.niSiteParams)
    assert avgBetweenSiteParams(old, cur) == (-0.5) / 2 * 3
    assert AvgBasedStopCriterion(1.0)(EpEstimationContext(old, cur)) is True
    assert AvgBasedStopCriterion(0.5)(EpEstimationContext(old, cur)) is False


def test_boston_fixture_is_the_reference_data_file():
    data = np.loadtxt(os.path.join(GOLD, "boston.csv"))
    assert data.shape == (506, 14)                       # GpPredictorTest.scala:52-56 loads this file
    assert data[0, -1] == 24.0 and data[1, -1] == 21.6
    head = np.loadtxt(os.path.join(GOLD, "bostonPredResults_head.txt"))
    assert head.shape[1] == 4 and head[0, 3] == 24.0


def test_oracle_golden_vectors_reproduce():
    gold = json.load(open(os.path.join(GOLD, "oracle_vectors.json")))
    for c in gold["cases"]:
        X, y, Xs, th = np.asfortranarray(c["X"]), np.array(c["y"]), np.asfortranarray(c["Xs"]), np.array(c["theta"])
        L, alpha = orc.fit(X, y, th)
        np.testing.assert_allclose(alpha, c["alpha"], rtol=1e-12, atol=1e-14)
        mean, var, _, _ = orc.predict(X, th, L, alpha, Xs)
        np.testing.assert_allclose(mean, c["mean"], rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(var, c["var"], rtol=1e-11, atol=1e-14)
        lml, grad = orc.lml_grad(X, y, th)
        assert abs(lml - c["lml"]) <= 1e-12 * abs(c["lml"])
        np.testing.assert_allclose(grad, c["grad"], rtol=1e-10)
        ep = orc.ep_estimate(orc.gram_sym(X, np.array(c["theta_class"])), np.array(c["y_class"]), 3)
        np.testing.assert_allclose(ep["tau"], c["ep_tau"], rtol=1e-11)
        np.testing.assert_allclose(ep["nu"], c["ep_nu"], rtol=1e-11)


def test_oracle_lml_gradient_matches_finite_differences():
    g = json.load(open(os.path.join(GOLD, "oracle_vectors.json")))["cases"][1]
    X, y, th = np.asfortranarray(g["X"]), np.array(g["y"]), np.array(g["theta"])
    _, grad = orc.lml_grad(X, y, th)
    for k in range(th.size):
        tp, tm = th.copy(), th.copy()
        tp[k] += 1e-6
        tm[k] -= 1e-6
        La, aa = orc.fit(X, y, tp)
        Lb, ab = orc.fit(X, y, tm)
        fd = (orc.lml(La, aa, y) - orc.lml(Lb, ab, y)) / 2e-6
        assert abs(fd - grad[k]) <= 1e-5 * max(1.0, abs(grad[k]))


def test_oracle_ep_fixed_point_and_strict_vs_corrected():
    """Closed checks that do not need the JVM: at convergence the EP marginals match the tilted moments, and
    corrected - strict equals exactly the dropped term sum(0.5 log(1 + tau/cav_tau) - log L_ii)."""
    g = json.load(open(os.path.join(GOLD, "oracle_vectors.json")))["cases"][1]
    X, yc, thc = np.asfortranarray(g["X"]), np.array(g["y_class"]), np.array(g["theta_class"])
    K = orc.gram_sym(X, thc)
    ep = orc.ep_estimate(K, yc, 40)
    for i in range(len(yc)):
        mh, sh = orc.marginal_moments(ep["cav_nu"][i] / ep["cav_tau"][i], 1.0 / ep["cav_tau"][i], yc[i])
        assert abs(mh - ep["mu"][i]) <= 1e-6 * max(1.0, abs(mh))
        assert abs(sh - ep["Sigma"][i, i]) <= 1e-6
    dropped = sum(0.5 * np.log(1 + ep["tau"][i] / ep["cav_tau"][i]) - np.log(ep["L"][i, i]) for i in range(len(yc)))
    assert abs((orc.ep_lml(ep, yc, False) - orc.ep_lml(ep, yc, True)) - dropped) <= 1e-10 * max(1.0, abs(dropped))
    # Sigma from the literal recursion equals the closed form (K^-1 + diag(tau))^-1
    closed = np.linalg.inv(np.linalg.inv(K) + np.diag(ep["tau"]))
    assert np.max(np.abs(closed - ep["Sigma"])) <= 1e-7
