"""EP at the edges of EpParameterEstimator.estimateSiteParams (gp/classification/EpParameterEstimator.scala:40-62) as WRITTEN --
VERDICT r03 missing #4: the device's chain kernels use a re-derived "moment form" of c and of the site update, so the two places
where the reference leans on IEEE arithmetic are pinned against the oracle's literal loop, ONE outcome each:

  * `:52-53`  Delta tau~ = 0 exactly  =>  `1/(1/0 + Sigma_ii)` = 1/inf = 0: the rank-1 update vanishes, the site keeps tau~ = 0 and
    still gets its nu~.  Built deterministically: a site whose prior variance is 2^-70 (block-diagonal K), so that the tilted variance
    equals the cavity variance to the last bit and 1/(1/x) round-trips.
  * `:56`     tau~ < 0 (no guard in the reference)  =>  sqrt gives NaN, `cholesky(I + S^1/2 K S^1/2)` fails.  Built deterministically:
    a site whose prior variance is 1e160, so sigma_-^4 overflows, the tilted variance is -inf, Delta tau~ = -1e-160 and the
    rank-1 coefficient 1/(1/Delta tau~ + Sigma_ii) = 1/0 = inf turns the whole covariance into NaN: every site from that one on is
    NaN and the FIRST bad pivot is that site's (1-based) index -- on both sides.
  * duplicate training points with sigma_n = 0 (singular K): EP never factors K itself, the run is regular.
  * an absurd signal variance (sf = 1e8): the literal loop stays finite; so must the device.
Sizes 70 (one site block) and 130 / 200 (two blocks: the link between blocks sees the same values)."""
import numpy as np
import pytest

from gp_algos_amd import synth
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu
TOL_EP = 1e-8      # EP site parameters after fixed sweeps


@pytest.fixture(scope="module")
def ctx():
    from gp_algos_amd.core import Context
    c = Context(0)
    yield c
    c.close()


def _block_diag_problem(n, value, pos, seed=5):
    """RBF Gram of n-1 points (sigma_n = 0) with ONE extra site of prior variance `value` at index `pos`, uncorrelated with the rest."""
    p = synth.regression(n - 1, 2, 0, seed, seed + 1, 0, synth.ard_theta(2, 1.3, 1.0, 0.0))
    Kr = orc.gram_sym(p["X"], p["theta"])
    K = np.zeros((n, n), order="F")
    idx = [i for i in range(n) if i != pos]
    K[np.ix_(idx, idx)] = Kr
    K[pos, pos] = value
    y = np.where(np.arange(n) % 3 == 0, -1, 1).astype(np.int32)
    return K, y


def _compare(ctx, K, y, sweeps, tol=TOL_EP):
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import EpClassifierState
    o = orc.ep_estimate(K, y, sweeps)
    ep = EpClassifierState(ctx, K, y)
    tau, nu = ep.sweep(sweeps)
    got = dict(tau=tau, nu=nu, mu=ep.get(L.GP_EP_GET_MU), Sigma=ep.get(L.GP_EP_GET_SIGMA), L=ep.get(L.GP_EP_GET_L))
    for key in ("tau", "nu", "mu", "Sigma", "L"):
        assert np.all(np.isfinite(got[key])), key
        assert np.max(np.abs(got[key] - o[key])) <= tol * np.max(np.abs(o[key])), (key, np.max(np.abs(got[key] - o[key])) / np.max(np.abs(o[key])))
    for strict in (True, False):
        ol = orc.ep_lml(o, y, strict=strict)
        assert abs(ep.lml(strict=strict) - ol) <= 1e-9 * max(1.0, abs(ol)), strict
    ep.close()
    return o, got


@pytest.mark.parametrize("n,pos", [(70, 69), (70, 37), (130, 129), (200, 5)])
def test_ep_site_with_zero_precision_update(ctx, n, pos):
    """Delta tau~ = 0 exactly (EpParameterEstimator.scala:52-53): c = 1/(1/0 + Sigma_ii) = 0."""
    K, y = _block_diag_problem(n, 2.0 ** -70, pos)
    o, got = _compare(ctx, K, y, 3)
    assert o["tau"][pos] == 0.0 and got["tau"][pos] == 0.0                     # the site never gains precision, to the bit
    assert abs(o["nu"][pos]) > 0.5 and abs(got["nu"][pos] - o["nu"][pos]) <= 1e-12 * abs(o["nu"][pos])   # but it does get its nu~ = y phi(0)/Phi(0)


@pytest.mark.parametrize("n,pos", [(70, 69), (70, 37), (130, 129), (130, 3), (200, 140)])
def test_ep_negative_site_precision_fails_at_the_same_pivot(ctx, n, pos):
    """tau~ < 0 (EpParameterEstimator.scala:56, no guard): NaN from the square root, the factorisation of I + S^1/2 K S^1/2 fails --
    with the reference's first bad pivot, not merely "somewhere"."""
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import EpClassifierState
    K, y = _block_diag_problem(n, 1e160, pos)
    with pytest.raises(orc.NotPositiveDefinite) as eo:
        orc.ep_estimate(K, y, 2)
    assert "pivot %d" % (pos + 1) in str(eo.value)
    ep = EpClassifierState(ctx, K, y)
    with pytest.raises(L.NotPositiveDefinite) as eg:
        ep.sweep(2)
    assert eg.value.info == pos + 1
    ep.close()
    # a healthy run on the same context afterwards
    K2, y2 = _block_diag_problem(n, 1.0, pos)
    _compare(ctx, K2, y2, 2)


@pytest.mark.parametrize("n", [70, 130])
def test_ep_site_with_huge_but_representable_prior_variance(ctx, n):
    """one decade below the overflow: sigma_-^4 = 1e306 is still a number, the site ends with tau~ = 1.75e-153 on both sides"""
    K, y = _block_diag_problem(n, 1e153, n // 2)
    o, got = _compare(ctx, K, y, 2)
    assert o["tau"][n // 2] > 0 and abs(got["tau"][n // 2] - o["tau"][n // 2]) <= 1e-8 * o["tau"][n // 2]


@pytest.mark.parametrize("n,dup", [(72, 8), (136, 20)])
def test_ep_duplicate_rows(ctx, n, dup):
    """duplicate training points, sigma_n = 0: K is singular, EP (which factors I + S^1/2 K S^1/2, never K) is not"""
    p = synth.regression(n - dup, 2, 0, 9, 10, 0, synth.ard_theta(2, 1.3, 1.0, 0.0))
    X = np.asfortranarray(np.vstack([p["X"], p["X"][:dup]]))
    K = orc.gram_sym(X, p["theta"])
    y = np.where(np.arange(n) % 2 == 0, -1, 1).astype(np.int32)
    y[n - dup:] = y[:dup]
    o, got = _compare(ctx, K, y, 4)
    assert np.max(np.abs(got["tau"][n - dup:] - got["tau"][:dup])) <= 1e-3 * np.max(got["tau"])      # twins end up (nearly) alike


def test_ep_absurd_signal_variance_stays_finite_like_the_literal_loop(ctx):
    """sf = 1e8 (the failing member of tests/test_gpu_ep_optimize.py, which only asks that a failure be reported consistently): the
    literal loop is finite at this size, so the device must be -- site parameters down to 1e-18 at the stated tolerance"""
    p = synth.regression(130, 2, 0, 51, 52, 0, synth.ard_theta(2, 1e8, 1.0, 0.0))
    K = orc.gram_sym(p["X"], p["theta"])
    y = np.where(p["y"] >= np.median(p["y"]), 1, -1).astype(np.int32)
    _compare(ctx, K, y, 3)
