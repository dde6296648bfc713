"""Boundary entry points a caller that HOLDS its own factor needs:
  gp_posterior_from_factor  GpPredictor.computePosterior(X, X*, l, alphaVec)           gp/regression/GpPredictor.scala:45-58
  gp_posterior_from_gram    the overload with an explicit (non-RBF) kernelFunc          :50-58
  gp_predict_from_gram      GpPredictor.predict on a model fitted from a host-built Gram :24-43
Every call goes through the C-ABI and is compared with the CPU oracle (scalar substitution in the reference's order)."""
import numpy as np
import pytest

from gp_algos_amd import synth
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

TOL_MEAN = 1e-9
TOL_VAR = 1e-9


@pytest.fixture(scope="module")
def ctx():
    from gp_algos_amd.core import Context
    c = Context(0)
    yield c
    c.close()


def _problem(n, d, m, seed):
    return synth.regression(n, d, m, seed, seed + 1, seed + 2, synth.ard_theta(d, 1.3, 1.0, 0.12))


def _periodic_kernel(x, y, same):
    """A KernelFunc that is not GaussianRbfKernel (shape of Co2Kernel's periodic term, Co2Prediction.scala:29-137)."""
    r = x - y   # per-dimension periodic term times an RBF envelope: a product of positive definite kernels
    return 1.7 * np.exp(-2.0 * (np.sin(0.9 * r) ** 2).sum() / 1.3 ** 2 - 0.5 * (r ** 2).sum() / 2.5 ** 2) + (0.04 if same else 0.0)


def _host_gram(A, B=None):
    sym = B is None
    B = A if sym else B
    return np.asfortranarray([[_periodic_kernel(a, b, sym and i == j) for j, b in enumerate(B)] for i, a in enumerate(A)])


@pytest.mark.parametrize("n,d,m", [(1, 1, 1), (12, 2, 1), (200, 3, 1), (200, 3, 40), (300, 8, 129), (640, 4, 257)])
def test_posterior_from_factor_vs_oracle(ctx, n, d, m):
    """m = 1 is the GP-UKF / GP-UCB shape (one sigma point / one candidate per call, GPUnscentedKalmanFilter.scala:78-87)."""
    p = _problem(n, d, m, seed=7 * n + m)
    Lo, ao = orc.fit(p["X"], p["y"], p["theta"])
    mean, var, cov, V = ctx.posterior_from_factor(p["X"], p["theta"], Lo, ao, p["Xs"], full_cov=True, want_v=True)
    omean, ovar, ocov, oV = orc.predict(p["X"], p["theta"], Lo, ao, p["Xs"], full_cov=True, want_v=True)
    sf2 = p["theta"][0] ** 2
    assert np.max(np.abs(mean - omean)) <= TOL_MEAN * max(1.0, np.max(np.abs(omean)))
    assert np.max(np.abs(var - ovar)) <= TOL_VAR * sf2
    assert np.max(np.abs(cov - ocov)) <= TOL_VAR * sf2
    assert np.array_equal(cov, cov.T)
    assert np.max(np.abs(V - oV)) <= 1e-9 * max(1.0, np.max(np.abs(oV)))
    # outputs are optional one by one
    mean2, var2, cov2, V2 = ctx.posterior_from_factor(p["X"], p["theta"], Lo, ao, p["Xs"])
    assert cov2 is None and V2 is None and np.array_equal(mean2, mean) and np.max(np.abs(var2 - var)) <= 1e-13 * sf2


def test_posterior_from_factor_ignores_upper_triangle_and_views(ctx):
    """The reference passes breeze's cholesky output (zero upper triangle); a caller handing over a full buffer with garbage
    above the diagonal, or strided views, must get the same answer."""
    from gp_algos_amd import _lib as L
    import ctypes as C
    p = _problem(150, 2, 9, seed=41)
    Lo, ao = orc.fit(p["X"], p["y"], p["theta"])
    ref = ctx.posterior_from_factor(p["X"], p["theta"], Lo, ao, p["Xs"])
    Lbig = np.full((170, 150), 7.5, order="F")
    Lbig[:150] = Lo + np.triu(np.full((150, 150), 3.25), 1)
    Xbig = np.full((190, 2), -9.0, order="F")
    Xbig[:150] = p["X"]
    mean, var = np.zeros(9), np.zeros(9)
    Xs = np.asfortranarray(p["Xs"])
    ctx.check(ctx._lib.gp_posterior_from_factor(ctx.h, L.dptr(Xbig), 150, 2, 190, L.dptr(L.f64(p["theta"])), L.dptr(Lbig), 170, L.dptr(ao),
                                                L.dptr(Xs), 9, 9, L.dptr(mean), L.dptr(var), None, 9, None, 150))
    assert np.array_equal(mean, ref[0]) and np.array_equal(var, ref[1])


def test_posterior_from_gram_any_kernel(ctx):
    rng = np.random.default_rng(5)
    n, m = 180, 33
    X, Xs = rng.uniform(-2, 2, (n, 2)), rng.uniform(-2, 2, (m, 2))
    y = np.sin(X.sum(axis=1)) + 0.1 * rng.standard_normal(n)
    K, Ks, Kss = _host_gram(X), _host_gram(Xs, X), _host_gram(Xs)
    Lo = orc.cholesky_lower(K)
    ao = orc.back_solve(Lo, orc.forward_solve(Lo, y), trans=True)
    oV = orc.forward_solve(Lo, np.asfortranarray(Ks.T))
    omean, ocov = Ks @ ao, Kss - oV.T @ oV
    mean, var, cov, V = ctx.posterior_from_gram(Ks, Lo, ao, Kss=Kss, want_v=True)
    assert np.max(np.abs(mean - omean)) <= TOL_MEAN * max(1.0, np.max(np.abs(omean)))
    assert np.max(np.abs(cov - ocov)) <= 1e-9 * np.max(np.abs(Kss))
    assert np.max(np.abs(var - np.diag(ocov))) <= 1e-9 * np.max(np.abs(Kss))
    assert np.max(np.abs(V - oV)) <= 1e-9 * max(1.0, np.max(np.abs(oV)))
    # diagonal-only variant
    mean2, var2, cov2, _ = ctx.posterior_from_gram(Ks, Lo, ao, kss_diag=np.diag(Kss).copy())
    assert cov2 is None and np.array_equal(mean2, mean) and np.max(np.abs(var2 - var)) <= 1e-12
    # mean only
    mean3, var3, cov3, _ = ctx.posterior_from_gram(Ks, Lo, ao)
    assert var3 is None and cov3 is None and np.array_equal(mean3, mean)


def test_predict_from_gram_on_a_gram_fitted_model(ctx):
    from gp_algos_amd.core import RegressionModel
    rng = np.random.default_rng(6)
    n, m = 260, 70
    X, Xs = rng.uniform(-2, 2, (n, 3)), rng.uniform(-2, 2, (m, 3))
    y = np.cos(X[:, 0]) * X[:, 1] + 0.05 * rng.standard_normal(n)
    K, Ks, Kss = _host_gram(X), _host_gram(Xs, X), _host_gram(Xs)
    mdl = RegressionModel(ctx, y=y, gram=K)
    Lo = orc.cholesky_lower(K)
    ao = orc.back_solve(Lo, orc.forward_solve(Lo, y), trans=True)
    oV = orc.forward_solve(Lo, np.asfortranarray(Ks.T))
    mean, var, cov = mdl.predict_from_gram(Ks, Kss=Kss)
    assert np.max(np.abs(mean - Ks @ ao)) <= TOL_MEAN * max(1.0, np.max(np.abs(Ks @ ao)))
    assert np.max(np.abs(cov - (Kss - oV.T @ oV))) <= 1e-9 * np.max(np.abs(Kss))
    assert np.max(np.abs(var - np.diag(cov))) <= 1e-12
    with pytest.raises(ValueError):
        mdl.predict(Xs)                       # a Gram-fitted model has no training inputs: gp_predict refuses (d = 0)
    mdl.close()


def test_boundary_argument_errors(ctx):
    from gp_algos_amd import _lib as L
    p = _problem(30, 2, 4, seed=3)
    Lo, ao = orc.fit(p["X"], p["y"], p["theta"])
    Ks = orc.gram_cross(p["Xs"], p["X"], p["theta"])
    mean, cov = np.zeros(4), np.zeros((4, 4), order="F")
    # cov without Kss
    st = ctx._lib.gp_posterior_from_gram(ctx.h, L.dptr(Ks), 4, 30, 4, None, 4, None, L.dptr(Lo), 30, L.dptr(ao), L.dptr(mean), None,
                                         L.dptr(cov), 4, None, 30)
    assert st == L.GP_EINVAL
    # null factor
    st = ctx._lib.gp_posterior_from_factor(ctx.h, L.dptr(p["X"]), 30, 2, 30, L.dptr(L.f64(p["theta"])), None, 30, L.dptr(ao),
                                           L.dptr(L.f64(p["Xs"])), 4, 4, L.dptr(mean), None, None, 4, None, 30)
    assert st == L.GP_EINVAL
    # m = 0 is a no-op
    assert ctx._lib.gp_posterior_from_factor(ctx.h, L.dptr(p["X"]), 30, 2, 30, L.dptr(L.f64(p["theta"])), L.dptr(Lo), 30, L.dptr(ao),
                                             L.dptr(L.f64(p["Xs"])), 0, 1, L.dptr(mean), None, None, 1, None, 30) == L.GP_OK


def test_mirror_compute_posterior_both_kernel_routes(ctx):
    """GpPredictor.computePosterior through the Python mirror of the reference class: GaussianRbfKernel -> device Gram;
    a user KernelFunc -> host-built K*, K** (the reference's loops) and device solves."""
    import gp_algos_amd
    from gp_algos_amd.gp.regression.gp_predictor import GpPredictor
    from gp_algos_amd.utils.kernel_requisites import GaussianRbfKernel, GaussianRbfParams, KernelFunc
    gp_algos_amd.set_default_context(ctx)
    p = _problem(90, 2, 7, seed=19)
    th = p["theta"]
    kernel = GaussianRbfKernel(GaussianRbfParams(signalVar=th[0], lengthScales=th[1:-1], noiseVar=th[-1]))
    pred = GpPredictor(kernel)
    Lm, alpha, _ = pred.preComputeComponents(p["X"], None, p["y"])
    dist, V = pred.computePosterior(p["X"], p["Xs"], Lm, alpha)
    omean, _, ocov, oV = orc.predict(p["X"], th, Lm, alpha, p["Xs"], full_cov=True, want_v=True)
    assert np.max(np.abs(dist.mean - omean)) <= TOL_MEAN * max(1.0, np.max(np.abs(omean)))
    assert np.max(np.abs(dist.sigma - ocov)) <= TOL_VAR * th[0] ** 2
    assert V.shape == (90, 7) and np.max(np.abs(V - oV)) <= 1e-9 * max(1.0, np.max(np.abs(oV)))

    class Periodic(KernelFunc):
        hyperParams = None

        def apply(self, obj1, obj2, sameIndex):
            return _periodic_kernel(np.asarray(obj1), np.asarray(obj2), sameIndex)

        def changeHyperParams(self, dv):
            return self

    K = _host_gram(p["X"])
    Lo = orc.cholesky_lower(K)
    ao = orc.back_solve(Lo, orc.forward_solve(Lo, p["y"]), trans=True)
    dist2, V2 = GpPredictor(kernel).computePosterior(p["X"], p["Xs"], Lo, ao, Periodic())
    Ks, Kss = _host_gram(p["Xs"], p["X"]), _host_gram(p["Xs"])
    oV2 = orc.forward_solve(Lo, np.asfortranarray(Ks.T))
    assert np.max(np.abs(dist2.mean - Ks @ ao)) <= 1e-9 * max(1.0, np.max(np.abs(Ks @ ao)))
    assert np.max(np.abs(dist2.sigma - (Kss - oV2.T @ oV2))) <= 1e-9 * np.max(np.abs(Kss))
    assert np.max(np.abs(V2 - oV2)) <= 1e-9 * max(1.0, np.max(np.abs(oV2)))
