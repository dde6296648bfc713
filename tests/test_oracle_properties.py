"""Property tests (hypothesis) of the CPU oracle against independent numpy/scipy formulations on random small inputs:
the oracle is the yardstick for the HIP path, so its own algebra is checked from a second direction here."""
import numpy as np
import scipy.linalg as sla
from hypothesis import given, settings, strategies as st

from oracle import gp_oracle as orc

dims = st.tuples(st.integers(1, 24), st.integers(1, 4), st.integers(1, 9), st.integers(0, 2 ** 31 - 1))


def _prob(n, d, m, seed):
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.uniform(-2, 2, (n, d)))
    Xs = np.asfortranarray(rng.uniform(-2, 2, (m, d)))
    y = rng.normal(size=n)
    theta = np.concatenate(([rng.uniform(0.5, 2.0) * rng.choice([-1, 1])], rng.uniform(0.5, 2.5, d), [rng.uniform(0.05, 0.5)]))
    return X, Xs, y, theta


def _np_gram(A, B, theta, noise):
    Z1, Z2 = A / theta[1:-1], B / theta[1:-1]
    r2 = ((Z1[:, None, :] - Z2[None, :, :]) ** 2).sum(-1)
    K = theta[0] ** 2 * np.exp(-0.5 * r2)
    return K + (theta[-1] ** 2) * np.eye(len(A)) if noise else K


@settings(max_examples=40, deadline=None)
@given(dims)
def test_gram_fit_predict_lml_against_closed_forms(p):
    n, d, m, seed = p
    X, Xs, y, theta = _prob(n, d, m, seed)
    K = orc.gram_sym(X, theta)
    np.testing.assert_allclose(K, _np_gram(X, X, theta, True), rtol=1e-12, atol=1e-14)
    assert np.array_equal(K, K.T)
    L, alpha = orc.fit(X, y, theta)
    np.testing.assert_allclose(L @ L.T, K, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(alpha, sla.cho_solve((L, True), y), rtol=1e-8, atol=1e-10)
    mean, var, cov, V = orc.predict(X, theta, L, alpha, Xs, full_cov=True, want_v=True)
    Ks = _np_gram(Xs, X, theta, False)
    np.testing.assert_allclose(mean, Ks @ alpha, rtol=1e-9, atol=1e-10)
    Vref = sla.solve_triangular(L, Ks.T, lower=True)
    np.testing.assert_allclose(V, Vref, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(cov, _np_gram(Xs, Xs, theta, True) - Vref.T @ Vref, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(np.diag(cov), var, rtol=0, atol=1e-12)
    sign, logdet = np.linalg.slogdet(K)
    ref_lml = -0.5 * y @ alpha - 0.5 * logdet - 0.5 * n * np.log(2 * np.pi)
    assert abs(orc.lml(L, alpha, y) - ref_lml) <= 1e-9 * max(1.0, abs(ref_lml))


@settings(max_examples=25, deadline=None)
@given(dims)
def test_lml_gradient_against_trace_formula(p):
    n, d, _, seed = p
    X, _, y, theta = _prob(n, d, 1, seed)
    lml, grad = orc.lml_grad(X, y, theta)
    K = _np_gram(X, X, theta, True)
    Kinv = np.linalg.inv(K)
    a = Kinv @ y
    W = np.outer(a, a) - Kinv
    for pidx in range(1, d + 3):
        D = orc.dgram_sym(X, theta, pidx)
        assert abs(grad[pidx - 1] - 0.5 * np.sum(W * D)) <= 1e-7 * max(1.0, abs(grad[pidx - 1]))


@settings(max_examples=15, deadline=None)
@given(st.tuples(st.integers(2, 14), st.integers(0, 2 ** 31 - 1)))
def test_ep_sigma_matches_closed_form_after_each_run(p):
    n, seed = p
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.uniform(-2, 2, (n, 2)))
    y = np.where(X[:, 0] + 0.3 * rng.normal(size=n) > 0, 1, -1)
    K = orc.gram_sym(X, np.array([1.4, 1.0, 1.3, 0.1]))
    ep = orc.ep_estimate(K, y, 2)
    closed = np.linalg.inv(np.linalg.inv(K) + np.diag(ep["tau"]))
    np.testing.assert_allclose(ep["Sigma"], closed, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(ep["mu"], closed @ ep["nu"], rtol=1e-7, atol=1e-9)
    B = np.eye(n) + np.sqrt(np.outer(ep["tau"], ep["tau"])) * K
    np.testing.assert_allclose(ep["L"] @ ep["L"].T, B, rtol=1e-10, atol=1e-12)
