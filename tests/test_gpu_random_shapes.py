"""Generative shape coverage for the C-ABI on the GPU (hypothesis): sizes around the 16 / 64 / 128 tile and padding edges,
feature counts around the 8-wide register chunks, ragged test batches -- every draw against the oracle."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from gp_algos_amd import synth
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

edge = st.sampled_from([1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 191, 255, 256, 257, 300])
import os

SCALE = int(os.environ.get("GPCORE_TEST_EXAMPLES_SCALE", "1"))   # soak runs: more draws from the same strategies
COMMON = dict(deadline=None, max_examples=60 * SCALE, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow],
              derandomize=(SCALE == 1))


@pytest.fixture(scope="module")
def ctx():
    from gp_algos_amd.core import Context
    c = Context(0)
    yield c
    c.close()


def _prob(n, d, m, seed):
    theta = np.concatenate(([0.8 + 0.1 * (seed % 7)], 0.9 + (0.15 * np.arange(d)) % 1.3, [0.12 + 0.01 * (seed % 5)]))
    return synth.regression(n, d, m, 1000 + seed, 2000 + seed, 3000 + seed, theta)


@settings(**COMMON)
@given(n=edge, d=st.integers(1, 17), m=st.one_of(edge, st.integers(1, 40)), seed=st.integers(0, 10 ** 6))
def test_fit_predict_random_shapes(ctx, n, d, m, seed):
    from gp_algos_amd.core import RegressionModel
    p = _prob(n, d, m, seed)
    mdl = RegressionModel(ctx, p["X"], p["y"], p["theta"])
    mean, var, _ = mdl.predict(p["Xs"])
    lml = mdl.lml()
    mdl.close()
    L, a = orc.fit(p["X"], p["y"], p["theta"])
    omean, ovar, _, _ = orc.predict(p["X"], p["theta"], L, a, p["Xs"])
    assert np.max(np.abs(mean - omean)) <= 1e-9 * max(1.0, np.max(np.abs(omean)))
    assert np.max(np.abs(var - ovar)) <= 1e-9 * p["theta"][0] ** 2
    ol = orc.lml(L, a, p["y"])
    assert abs(lml - ol) <= 1e-11 * max(1.0, abs(ol))


@settings(**COMMON)
@given(n=st.sampled_from([1, 2, 17, 64, 127, 128, 129, 200]), d=st.integers(1, 10), B=st.integers(1, 7), seed=st.integers(0, 10 ** 6))
def test_lml_gradient_random_shapes(ctx, n, d, B, seed):
    p = _prob(n, d, 0, seed)
    rng = np.random.default_rng(seed)
    thetas = p["theta"][None, :] * rng.uniform(0.7, 1.5, size=(B, d + 2))
    lml, grad, info = ctx.lml_grad_batched(p["X"], p["y"], thetas)
    assert np.all(info == 0)
    for b in range(B):
        ol, og = orc.lml_grad(p["X"], p["y"], thetas[b])
        assert abs(lml[b] - ol) <= 1e-11 * max(1.0, abs(ol))
        assert np.max(np.abs(grad[b] - og)) <= 1e-8 * max(1e-3, np.max(np.abs(og)))


@settings(**{**COMMON, "max_examples": 25 * SCALE})
@given(n=st.sampled_from([2, 17, 100, 128, 129, 260]), sweeps=st.integers(1, 3), seed=st.integers(0, 10 ** 6))
def test_ep_random_shapes(ctx, n, sweeps, seed):
    from gp_algos_amd.core import EpClassifierState
    p = _prob(n, 3, 0, seed)
    theta = np.concatenate((p["theta"][:-1], [0.0]))
    f = p["X"].sum(axis=1) + 0.3 * synth.normal(seed + 9, np.arange(n))
    y = np.where(f >= 0.0, 1, -1).astype(np.int32)
    K = orc.gram_sym(p["X"], theta)
    st_ = EpClassifierState(ctx, K, y)
    tau, nu = st_.sweep(sweeps)
    lml = st_.lml(strict=False)
    st_.close()
    o = orc.ep_estimate(K, y, sweeps)
    assert np.max(np.abs(tau - o["tau"])) <= 1e-8 * max(1e-12, np.max(np.abs(o["tau"])))
    assert np.max(np.abs(nu - o["nu"])) <= 1e-8 * max(1e-12, np.max(np.abs(o["nu"])))
    ol = orc.ep_lml(o, y, False)
    assert abs(lml - ol) <= 1e-8 * max(1.0, abs(ol))
