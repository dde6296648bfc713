"""Full-size (BASELINE.json config) checks through size-independent properties -- the oracle is far too slow at
n = 8192 (minutes per factorisation, ~1 s per test point), so parity at these sizes is established by identities
that any correct implementation must satisfy, evaluated with an independent numpy path on the host:
  * K alpha = y          (round trip of Gram -> Cholesky -> two triangular solves), K rebuilt blockwise in numpy
  * L L^T x = K x        for random probe vectors x (factorisation residual without forming L L^T)
  * LML = -1/2 y.alpha - sum log diag(L) - n/2 log 2 pi  recomputed on the host from the downloaded L, alpha
  * posterior at TRAINING inputs: mean = y - sn^2 alpha,  var = sf^2+sn^2 - diag(K_f (K)^-1 K_f) via the same identity
  * a spot check of posterior points against the oracle (bounded: 3 test points)
  * EP (C4 size reduced to n = 2048 for memory/time): Sigma = (K^-1 + diag(tau))^-1 via Sigma (I + diag(tau) K)... identity"""
import numpy as np
import pytest

from gp_algos_amd import synth
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu


def _kmatvec(X, theta, V, noise=True, block=1024):
    """K @ V with K rebuilt blockwise in numpy (independent of the device Gram kernel)."""
    n, d = X.shape
    sf2, sn2 = theta[0] ** 2, theta[-1] ** 2
    Z = X / theta[1:-1]
    out = np.zeros_like(V)
    sq = (Z * Z).sum(axis=1)
    for lo in range(0, n, block):
        hi = min(n, lo + block)
        r2 = sq[lo:hi, None] + sq[None, :] - 2.0 * Z[lo:hi] @ Z.T
        Kb = sf2 * np.exp(-0.5 * np.maximum(r2, 0.0))
        if noise:
            Kb[np.arange(hi - lo), np.arange(lo, hi)] += sn2
        out[lo:hi] = Kb @ V
    return out


@pytest.fixture(scope="module")
def c2():
    from gp_algos_amd.core import Context, RegressionModel
    p = synth.config_c2(8192, 8, 256)
    ctx = Context(0)
    mdl = RegressionModel(ctx, p["X"], p["y"], p["theta"])
    yield ctx, mdl, p
    mdl.close()
    ctx.close()


def test_c2_fit_round_trips(c2):
    ctx, mdl, p = c2
    n = 8192
    alpha, L = mdl.alpha(), mdl.L()
    assert np.all(np.isfinite(alpha)) and np.all(np.diag(L) > 0)
    # K alpha = y
    Ka = _kmatvec(p["X"], p["theta"], alpha[:, None])[:, 0]
    assert np.max(np.abs(Ka - p["y"])) <= 1e-9 * np.max(np.abs(p["y"]))
    # L L^T x = K x for probe vectors
    rng = np.random.default_rng(0)
    V = rng.standard_normal((n, 4))
    LLt = L @ (L.T @ V)
    KV = _kmatvec(p["X"], p["theta"], V)
    assert np.linalg.norm(LLt - KV) / np.linalg.norm(KV) <= 1e-12
    # LML from its definition (GpPredictor.scala:144-149)
    lml = -0.5 * p["y"] @ alpha - np.log(np.diag(L)).sum() - 0.5 * n * np.log(2 * np.pi)
    assert abs(mdl.lml() - lml) <= 1e-11 * abs(lml)
    assert np.all(np.triu(L[:256, :256], 1) == 0.0) and L[0, n - 1] == 0.0


def test_c2_gram_matrices_rows_against_the_oracle(c2):
    """The Gram matrices at the BASELINE size (n = 8192, d = 8: 8256 tiles / 33 024 units of the unit kernel, jobs that cross strip
    boundaries): 96 rows spread over the matrix against the oracle's rows, elementwise 1e-13; exact diagonal; exact symmetry;
    the lower-only form writes nothing above the diagonal; a 16 384 x 8192 cross-Gram the same way."""
    ctx, mdl, p = c2
    n = 8192
    rows = np.unique(np.concatenate(([0, 1, 63, 64, 65, 4095, 4096, 8190, 8191], np.random.default_rng(3).integers(0, n, 87))))
    Ko = orc.gram_cross(np.asfortranarray(p["X"][rows]), p["X"], np.concatenate((p["theta"][:-1], [0.0])))
    diag = p["theta"][0] * p["theta"][0] + p["theta"][-1] * p["theta"][-1]
    K = ctx.gram_rbf(p["X"], p["theta"])
    assert np.all(np.diag(K) == diag) and np.array_equal(K, K.T)
    got = K[rows].copy()
    got[np.arange(rows.size), rows] = Ko[np.arange(rows.size), rows]       # the oracle's cross form carries no noise term
    assert np.max(np.abs(got - Ko) / Ko) <= 1e-13
    out = np.full((n, n), -7.0, order="F")
    Kl = ctx.gram_rbf(p["X"], p["theta"], full=False, out=out)
    assert np.array_equal(np.tril(Kl), np.tril(K))
    assert np.all(Kl[0, 1:] == -7.0) and np.all(Kl[4095, 4096:] == -7.0) and np.all(Kl[np.triu_indices(n, 1)][::4099] == -7.0)
    del K, Kl, out
    Xs = synth.config_c2(8192, 8, 16384)["Xs"]
    Ks = ctx.cross_gram_rbf(Xs, p["X"], p["theta"])
    rs = np.unique(np.concatenate(([0, 63, 64, 16383], np.random.default_rng(4).integers(0, 16384, 60))))
    Kso = orc.gram_cross(np.asfortranarray(Xs[rs]), p["X"], p["theta"])
    assert np.max(np.abs(Ks[rs] - Kso) / Kso) <= 1e-13


def test_c2_posterior_identities_and_spot_check(c2):
    ctx, mdl, p = c2
    sf2, sn2 = p["theta"][0] ** 2, p["theta"][-1] ** 2
    alpha = mdl.alpha()
    # at training inputs K* = K - sn^2 I, so mean = y - sn^2 alpha exactly
    idx = np.arange(0, 8192, 64)
    mean, var, _ = mdl.predict(p["X"][idx])
    assert np.max(np.abs(mean - (p["y"][idx] - sn2 * alpha[idx]))) <= 1e-8
    assert np.all(var > 0.0) and np.all(var <= sf2 + sn2 + 1e-9)
    # spot check against the oracle's scalar substitution (3 points ~ 3 s of CPU)
    mean, var, _ = mdl.predict(p["Xs"][:3])
    om, ov, _, _ = orc.predict(p["X"], p["theta"], mdl.L(), alpha, p["Xs"][:3])
    assert np.max(np.abs(mean - om)) <= 1e-9 * max(1.0, np.max(np.abs(om)))
    assert np.max(np.abs(var - ov)) <= 1e-9 * sf2
    # batch invariance: the same point gives the same answer in batches of different size (different padding, tiles)
    m2, v2, _ = mdl.predict(p["Xs"][:200])
    assert np.array_equal(m2[:3], mean) and np.max(np.abs(v2[:3] - var)) <= 1e-13


def test_c2_large_batch_path_agrees_with_small_batch_path_and_oracle(c2):
    """m = 32768 goes through the large-batch posterior (panel solves folded into the GEMMs through Lw, reductions in the GEMM
    epilogue); the same points in batches of 300 go through the right-looking row-panel solves.  Two different code paths, one
    answer -- and the oracle's scalar substitution on three of the points."""
    ctx, mdl, p = c2
    sf2 = p["theta"][0] ** 2
    big = synth.config_c2(8192, 8, 32768)["Xs"]
    mean, var, _ = mdl.predict(big)
    assert np.all(np.isfinite(mean)) and np.all(var > 0.0) and np.all(var <= sf2 + p["theta"][-1] ** 2 + 1e-9)
    idx = np.sort(np.random.default_rng(1).choice(32768, size=300, replace=False))
    m2, v2, _ = mdl.predict(np.asfortranarray(big[idx]))
    assert np.max(np.abs(mean[idx] - m2)) <= 1e-10 * max(1.0, np.max(np.abs(m2)))
    assert np.max(np.abs(var[idx] - v2)) <= 1e-10 * sf2
    om, ov, _, _ = orc.predict(p["X"], p["theta"], mdl.L(), mdl.alpha(), np.asfortranarray(big[idx[:3]]))
    assert np.max(np.abs(mean[idx[:3]] - om)) <= 1e-9 * max(1.0, np.max(np.abs(om)))
    assert np.max(np.abs(var[idx[:3]] - ov)) <= 1e-9 * sf2


def test_odd_sizes_through_the_padded_large_batch_path():
    """n = 5000 (padded to 5120), d = 5, m = 25100 (padded to 25216, an odd number of 128-row tiles): identities at the training inputs and the two posterior
    code paths against each other."""
    from gp_algos_amd.core import Context, RegressionModel
    p = synth.regression(5000, 5, 25000, 71, 72, 73, synth.ard_theta(5, 1.2, 1.0, 0.15))
    sf2, sn2 = p["theta"][0] ** 2, p["theta"][-1] ** 2
    with Context(0) as ctx:
        mdl = RegressionModel(ctx, p["X"], p["y"], p["theta"])
        alpha = mdl.alpha()
        Ka = _kmatvec(p["X"], p["theta"], alpha[:, None])[:, 0]
        assert np.max(np.abs(Ka - p["y"])) <= 1e-9 * np.max(np.abs(p["y"]))
        both = np.asfortranarray(np.vstack([p["X"][:1000], p["Xs"][:24100]]))      # 25100 rows -> 25216 padded = 98 tiles of 256 + one of 128
        mean, var, _ = mdl.predict(both)
        assert np.max(np.abs(mean[:1000] - (p["y"][:1000] - sn2 * alpha[:1000]))) <= 1e-8     # K* = K - sn^2 I at training inputs
        m2, v2, _ = mdl.predict(np.asfortranarray(both[900:1300]))                    # small batch: right-looking path
        assert np.max(np.abs(mean[900:1300] - m2)) <= 1e-10 * max(1.0, np.max(np.abs(m2)))
        assert np.max(np.abs(var[900:1300] - v2)) <= 1e-10 * sf2
        mdl.close()


def test_c3_sized_lml_gradient_against_finite_differences():
    from gp_algos_amd.core import Context
    p = synth.config_c3(4096, 8)
    th = p["thetas"][21]
    with Context(0) as ctx:
        lml, grad, info = ctx.lml_grad_batched(p["X"], p["y"], th[None, :])
        assert info[0] == 0
        for k in (0, 3, 9):
            tp, tm = th.copy(), th.copy()
            h = 1e-5 * abs(th[k])
            tp[k] += h
            tm[k] -= h
            (lp, lm_), _, _ = ctx.lml_grad_batched(p["X"], p["y"], np.stack([tp, tm]), nparams=0)
            assert abs((lp - lm_) / (2 * h) - grad[0, k]) <= 2e-5 * max(1.0, abs(grad[0, k]))


def test_c4_sized_ep_closed_form_identity():
    """After any number of sweeps the refactored Sigma must equal (K^-1 + diag(tau))^-1, i.e.
    Sigma (I + diag(tau) K) = K, and mu = Sigma nu; checked with probe vectors at n = 2048."""
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import Context, EpClassifierState
    p = synth.config_c4(2048, 8)
    with Context(0) as ctx:
        K = ctx.gram_rbf(p["X"], p["theta"])
        K[np.diag_indices_from(K)] += 1e-6     # the C4 kernel has sn = 0: keep K numerically PD for the host-side check
        ep = EpClassifierState(ctx, K, p["y"])
        tau, nu = ep.sweep(3)
        assert np.all(tau > 0)
        Sig, mu = ep.get(L.GP_EP_GET_SIGMA), ep.get(L.GP_EP_GET_MU)
        rng = np.random.default_rng(1)
        V = rng.standard_normal((2048, 3))
        lhs = Sig @ (V + tau[:, None] * (K @ V))
        assert np.linalg.norm(lhs - K @ V) / np.linalg.norm(K @ V) <= 1e-9
        assert np.max(np.abs(Sig @ nu - mu)) <= 1e-9 * np.max(np.abs(mu))
        assert np.array_equal(Sig, Sig.T)
        ep.close()
