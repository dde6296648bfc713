"""BASELINE.json configurations C3, C4, C5 at their FULL sizes through the C-ABI, and the lockstep batched LML + gradient
path past one outer panel (np > 512 with several settings per launch) against the oracle.

The oracle (scalar loops in the reference's order) finishes n <= ~1100 in seconds, so the lockstep path is compared with it
setting by setting there.  At n = 4096 / 32768 parity rests on size-independent identities evaluated by an independent
numpy/LAPACK path on the host (K rebuilt from X in numpy, never taken from the device):
  C3  LML = -1/2 y.alpha - sum log diag(chol K) - n/2 log 2 pi and g_p = 1/2 tr((alpha alpha^T - K^-1) dK/dtheta_p)
      (GpPredictor.scala:60-80,144-149) from scipy's Cholesky for a sample of the 64 settings, all P components
  C4  after 50 sweeps: Sigma (I + diag(tau) K) = K, mu = Sigma nu, Sigma symmetric (EpParameterEstimator.scala:56-61),
      and the sweep is at its fixed point (one more sweep moves tau, nu by < 1e-6 relative)
  C5  (K alpha)_i = y_i and mean_j = k*_j . alpha on sampled rows, 0 < var <= sf^2 + sn^2, the large-batch posterior path
      against the small-batch one and two points against the oracle's scalar substitution."""
import numpy as np
import pytest

from gp_algos_amd import synth
from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu

TOL_LML = 1e-11
TOL_GRAD = 1e-8


@pytest.fixture(scope="module")
def ctx():
    from gp_algos_amd.core import Context
    c = Context(0)
    yield c
    c.close()


# ---- lockstep batch past one outer panel ---------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,B,workers,tinv_outer", [
    (700, 3, 9, "1", None),      # one group of 9: batched K = 512 trailing update, two-level T = L^-T (count >= 4 -> 512)
    (700, 3, 9, "1", "128"),     # same group, plain right-looking T = L^-T
    (1100, 2, 6, "1", None),     # np = 1152: two full outer panels + a partial one, one group of 6
    (1100, 2, 6, "2", "512"),    # two workers x groups of 3 (count < 4), two-level form forced
    (1100, 2, 5, "2", None),     # ragged: groups of 3 and 2, default forms
])
def test_lockstep_batch_across_outer_panels_vs_oracle(ctx, monkeypatch, n, d, B, workers, tinv_outer):
    """chol_blocked's batched K = 512 trailing update (count > 1), inverse_transpose_lower's `right > 0` branch and the batched
    T T^T with more than 8 tile rows: every setting of the group against orc.lml_grad (GpPredictor.scala:60-80)."""
    monkeypatch.setenv("GPCORE_LML_WORKERS", workers)
    if tinv_outer is None:
        monkeypatch.delenv("GPCORE_TINV_OUTER", raising=False)
    else:
        monkeypatch.setenv("GPCORE_TINV_OUTER", tinv_outer)
    p = synth.regression(n, d, 0, 300 + n, 301 + n, 0, synth.ard_theta(d, 1.2, 1.0, 0.15))
    rng = np.random.default_rng(n + B)
    thetas = p["theta"][None, :] * rng.uniform(0.6, 1.7, size=(B, d + 2))
    lml, grad, info = ctx.lml_grad_batched(p["X"], p["y"], thetas)
    assert np.all(info == 0)
    for b in range(B):
        ol, og = orc.lml_grad(p["X"], p["y"], thetas[b])
        assert abs(lml[b] - ol) <= TOL_LML * abs(ol), b
        assert np.max(np.abs(grad[b] - og)) <= TOL_GRAD * np.max(np.abs(og)), b
    # the same setting alone (count = 1: single-problem forms of every step) agrees to rounding
    one, gone, _ = ctx.lml_grad_batched(p["X"], p["y"], thetas[B - 1:B])
    assert abs(one[0] - lml[B - 1]) <= 1e-12 * abs(one[0])
    assert np.max(np.abs(gone[0] - grad[B - 1])) <= 1e-10 * np.max(np.abs(gone[0]))


def test_lockstep_batch_with_a_non_pd_member_past_one_outer_panel(ctx, monkeypatch):
    """A non-PD setting inside a lockstep group at np = 768: its failing pivot is reported, its group mates still match."""
    monkeypatch.setenv("GPCORE_LML_WORKERS", "1")
    p = synth.regression(640, 2, 0, 77, 78, 0, synth.ard_theta(2, 1.1, 1.0, 0.2))
    X = np.asfortranarray(np.vstack([p["X"], p["X"][:30]]))      # 30 duplicated rows: singular without noise
    y = np.concatenate([p["y"], p["y"][:30]])
    thetas = p["theta"][None, :] * np.random.default_rng(9).uniform(0.7, 1.5, size=(5, 4))
    thetas[2, -1] = 0.0
    lml, grad, info = ctx.lml_grad_batched(X, y, thetas)
    assert 0 < info[2] <= 670 and np.isnan(lml[2]) and np.all(np.isnan(grad[2]))   # 1-based failing pivot (where exactly is rounding)
    for b in (0, 1, 3, 4):
        ol, og = orc.lml_grad(X, y, thetas[b])
        assert info[b] == 0 and abs(lml[b] - ol) <= TOL_LML * abs(ol)
        assert np.max(np.abs(grad[b] - og)) <= TOL_GRAD * np.max(np.abs(og))
    # the workers keep the zeros of their factor and inverse buffers from call to call (no 4 GB clear per call); a failed member
    # makes them clear again, and the calls after it -- same shape, then another shape, then the first again -- give the same bits
    good = thetas[[0, 1, 3, 4, 0]]
    l1, g1, i1 = ctx.lml_grad_batched(X, y, good)
    l2, g2, i2 = ctx.lml_grad_batched(X, y, good)
    assert np.all(i1 == 0) and np.array_equal(l1, l2) and np.array_equal(g1, g2)
    assert np.array_equal(l1[:4], lml[[0, 1, 3, 4]]) and np.array_equal(g1[:4], grad[[0, 1, 3, 4]]) and l1[4] == l1[0]
    ctx.lml_grad_batched(X[:300].copy(order="F"), y[:300], good[:2])
    K = ctx.gram_rbf(X, good[0])          # another user of the same workspace slot in between
    assert K.shape == (670, 670)
    l3, g3, _ = ctx.lml_grad_batched(X, y, good)
    assert np.array_equal(l1, l3) and np.array_equal(g1, g3)


# ---- C3 at full size -------------------------------------------------------------------------------------------------------
def _host_lml_grad(X, y, theta):
    """Closed forms of GpPredictor.scala:60-80,144-149 on LAPACK (independent of the device and of the oracle's loops)."""
    import scipy.linalg as sla
    n, d = X.shape
    sf, ell, sn = theta[0], theta[1:-1], theta[-1]
    Z = X / ell
    sq = (Z * Z).sum(axis=1)
    E = np.exp(-0.5 * np.maximum(sq[:, None] + sq[None, :] - 2.0 * Z @ Z.T, 0.0))
    np.fill_diagonal(E, 1.0)
    K = sf * sf * E
    K[np.diag_indices(n)] += sn * sn
    c = sla.cho_factor(K, lower=True, check_finite=False)
    alpha = sla.cho_solve(c, y, check_finite=False)
    lml = -0.5 * y @ alpha - np.log(np.diag(c[0])).sum() - 0.5 * n * np.log(2 * np.pi)
    W = np.outer(alpha, alpha) - sla.cho_solve(c, np.eye(n), check_finite=False)
    WE = W * E
    g = np.zeros(d + 2)
    g[0] = sf * WE.sum()
    for k in range(d):
        D2 = (X[:, k][:, None] - X[:, k][None, :]) ** 2
        g[1 + k] = 0.5 * sf * sf / ell[k] ** 3 * (WE * D2).sum()
    g[d + 1] = sn * np.trace(W)
    return lml, g


def test_c3_full_size_all_64_settings(ctx):
    """Config C3 as benchmarked: n = 4096, d = 8, all 64 settings in lockstep groups of 32 on two workers."""
    p = synth.config_c3(4096, 8)
    B = p["thetas"].shape[0]
    assert B == 64
    lml, grad, info = ctx.lml_grad_batched(p["X"], p["y"], p["thetas"])
    assert np.all(info == 0) and np.all(np.isfinite(lml)) and np.all(np.isfinite(grad))
    for b in (0, 13, 21, 38, 47, 63):                 # corners and interior of the 4 x 4 x 4 grid
        hl, hg = _host_lml_grad(p["X"], p["y"], p["thetas"][b])
        assert abs(lml[b] - hl) <= 1e-10 * abs(hl), b
        assert np.max(np.abs(grad[b] - hg)) <= 1e-7 * np.max(np.abs(hg)), b
    # group members do not influence each other: three settings alone (count = 1) agree to rounding
    for b in (5, 33, 62):
        one, gone, _ = ctx.lml_grad_batched(p["X"], p["y"], p["thetas"][b:b + 1])
        assert abs(one[0] - lml[b]) <= 1e-12 * abs(one[0])
        assert np.max(np.abs(gone[0] - grad[b])) <= 1e-9 * np.max(np.abs(gone[0]))
    # central differences on three settings (two LML-only evaluations each, batched)
    for b, k in ((21, 0), (21, 4), (40, 9)):
        th = p["thetas"][b]
        h = 1e-5 * abs(th[k])
        tp, tm = th.copy(), th.copy()
        tp[k] += h
        tm[k] -= h
        (lp, lm_), _, _ = ctx.lml_grad_batched(p["X"], p["y"], np.stack([tp, tm]), nparams=0)
        assert abs((lp - lm_) / (2 * h) - grad[b, k]) <= 2e-5 * max(1.0, abs(grad[b, k]))


# ---- C4 at full size -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("jitter", [0.0, 1e-6])
def test_c4_full_size_50_sweeps(ctx, jitter):
    """Config C4: n = 4096, 50 EP sweeps (EpParameterEstimator.scala:29-69).  jitter = 0 is the matrix bench.py --workload c4 runs
    (BASELINE C4: sigma_n = 0, so K is only positive SEMI-definite in floating point -- EP never factors K, only
    B = I + S^1/2 K S^1/2, and every identity below holds for a semi-definite K); 1e-6 is the well-conditioned copy kept from
    round 2."""
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import EpClassifierState
    n = 4096
    p = synth.config_c4(n, 8)
    K = ctx.gram_rbf(p["X"], p["theta"], full=True)      # exactly what bench.py hands to gp_ep_create
    K[np.diag_indices_from(K)] += jitter
    ep = EpClassifierState(ctx, K, p["y"])
    tau, nu = ep.sweep(50)
    assert np.all(np.isfinite(tau)) and np.all(np.isfinite(nu)) and np.all(tau > 0)
    Sig, mu = ep.get(L.GP_EP_GET_SIGMA), ep.get(L.GP_EP_GET_MU)
    V = np.random.default_rng(1).standard_normal((n, 3))
    KV = K @ V
    lhs = Sig @ (V + tau[:, None] * KV)
    assert np.linalg.norm(lhs - KV) / np.linalg.norm(KV) <= 1e-9
    assert np.max(np.abs(Sig @ nu - mu)) <= 1e-9 * np.max(np.abs(mu))
    assert np.array_equal(Sig, Sig.T)
    Lf = ep.get(L.GP_EP_GET_L)              # L L^T = I + S^1/2 K S^1/2  (:56-58)
    st = np.sqrt(tau)
    BV = V + st[:, None] * (K @ (st[:, None] * V))
    assert np.linalg.norm(Lf @ (Lf.T @ V) - BV) / np.linalg.norm(BV) <= 1e-12
    l_strict, l_corr = ep.lml(True), ep.lml(False)
    assert np.isfinite(l_strict) and np.isfinite(l_corr) and l_strict != l_corr
    # labels agree with the latent mean on the training set far better than chance
    assert np.mean(np.sign(mu) == p["y"]) > 0.85
    # 50 sweeps is a fixed point of the sweep map
    tau2, nu2 = ep.sweep(1)
    assert np.max(np.abs(tau2 - tau)) <= 1e-6 * np.max(np.abs(tau))
    assert np.max(np.abs(nu2 - nu)) <= 1e-6 * np.max(np.abs(nu))
    ep.close()


def test_ep_streamed_refactorisation_at_a_ragged_size(ctx, monkeypatch):
    """n = 3000 (np = 3072: six outer panels, the last site block partly padding): the refactorisation streamed under the site loop
    (default at this size) against the end-of-sweep form and against the identities of EpParameterEstimator.scala:56-61."""
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import EpClassifierState
    n = 3000
    p = synth.config_c4(n, 8)
    K = ctx.gram_rbf(p["X"], p["theta"])
    K[np.diag_indices_from(K)] += 1e-6
    got = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GPCORE_EP_PIPELINE", mode)
        ep = EpClassifierState(ctx, K, p["y"])
        tau, nu = ep.sweep(6)
        got[mode] = dict(tau=tau, nu=nu, Sig=ep.get(L.GP_EP_GET_SIGMA), mu=ep.get(L.GP_EP_GET_MU), L=ep.get(L.GP_EP_GET_L),
                         lml=ep.lml(False))
        ep.close()
    g = got["1"]
    assert np.all(np.isfinite(g["tau"])) and np.all(g["tau"] > 0)
    V = np.random.default_rng(2).standard_normal((n, 3))
    KV = K @ V
    assert np.linalg.norm(g["Sig"] @ (V + g["tau"][:, None] * KV) - KV) / np.linalg.norm(KV) <= 1e-9
    assert np.max(np.abs(g["Sig"] @ g["nu"] - g["mu"])) <= 1e-9 * np.max(np.abs(g["mu"]))
    st = np.sqrt(g["tau"])
    BV = V + st[:, None] * (K @ (st[:, None] * V))
    assert np.linalg.norm(g["L"] @ (g["L"].T @ V) - BV) / np.linalg.norm(BV) <= 1e-12
    assert np.all(np.triu(g["L"], 1) == 0.0) and np.array_equal(g["Sig"], g["Sig"].T)
    for key in ("tau", "nu", "mu", "Sig", "L"):
        assert np.max(np.abs(g[key] - got["0"][key])) <= 1e-9 * np.max(np.abs(got["0"][key])), key
    assert abs(g["lml"] - got["0"]["lml"]) <= 1e-9 * abs(got["0"]["lml"])


@pytest.mark.parametrize("n", [1100, 2500, 4000, 5200])
def test_cholesky_single_launch_and_lookahead_forms_agree_with_the_first_bad_pivot(ctx, monkeypatch, n):
    """The three ways a single matrix is factored -- ONE persistent launch over a task list (chol_mega_kernel, round 4: default where
    look-ahead is on and np >= 2048), the launch-per-step form with the far trailing updates on the CU-masked side stream, and the
    one-stream form -- apply the same products in the same order per element, so the factor is IDENTICAL, and so is the first bad
    pivot: inside the first panel, inside a later panel, in the last block.  3 / 5 / 8 / 11 outer panels, ragged last blocks."""
    from gp_algos_amd import _lib as L
    p = synth.regression(n, 3, 0, 5, 6, 0, synth.ard_theta(3, 1.3, 0.9, 0.3))
    K = orc.gram_sym(p["X"], p["theta"])
    lib = ctx._lib
    modes = (("1", 1), ("0", 1), ("0", 0))          # (GPCORE_CHOL_MEGA, look-ahead)
    try:
        got = []
        for mega, la in modes:
            monkeypatch.setenv("GPCORE_CHOL_MEGA", mega)
            ctx.check(lib.gp_ctx_set_lookahead(ctx.h, la))
            got.append([ctx.potrf_lower(K.copy(order="F")) for _ in range(2)])
        one = got[-1][0]
        for pair in got:
            for g in pair:
                assert np.array_equal(g, one)
        assert np.all(np.triu(one, 1) == 0.0)
        assert np.linalg.norm(one @ one.T - K) / np.linalg.norm(K) <= 1e-13
        if n <= 1100:
            Lo = orc.cholesky_lower(K)
            assert np.max(np.abs(one - Lo)) <= 1e-10 * np.max(np.abs(Lo))
        for j in (n - 7, 700, 130):
            Kbad = K.copy(order="F")
            Kbad[j, j] = -1.0
            for mega, la in modes:
                monkeypatch.setenv("GPCORE_CHOL_MEGA", mega)
                ctx.check(lib.gp_ctx_set_lookahead(ctx.h, la))
                with pytest.raises(L.NotPositiveDefinite) as ei:
                    ctx.potrf_lower(Kbad.copy(order="F"))
                assert ei.value.info == j + 1
        # and a healthy factorisation right after the failing ones (the flags of the task list are per launch)
        monkeypatch.setenv("GPCORE_CHOL_MEGA", "1")
        ctx.check(lib.gp_ctx_set_lookahead(ctx.h, 1))
        assert np.array_equal(ctx.potrf_lower(K.copy(order="F")), one)
    finally:
        ctx.check(lib.gp_ctx_set_lookahead(ctx.h, -1))


def test_cholesky_default_is_the_single_launch_where_it_wins(monkeypatch):
    """Which form runs is a size rule inside the library (5120 <= rows <= 14336: profiles/r04_final_fit_mega.log) plus one condition:
    a shape is planned when it comes the second time in a row (the plan costs 4-68 ms of host time, a one-off factorisation is better
    off without it).  The library's own launch counters say which form ran: the single persistent launch is ONE launch of the
    trailing-update class per refit, the launch-per-step form one per outer panel and more.  Same factor either way."""
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import Context, RegressionModel
    ctx = Context(0)                # its own context: what a context has planned before is part of the rule
    lib = ctx._lib
    launches, factors, first = {}, {}, {}
    for n in (6400, 4608):
        p = synth.regression(n, 4, 0, 61, 62, 0, np.array([1.2, 0.9, 1.3, 0.7, 1.1, 0.2]))
        for mega in (None, "0"):
            if mega is None:
                monkeypatch.delenv("GPCORE_CHOL_MEGA", raising=False)
            else:
                monkeypatch.setenv("GPCORE_CHOL_MEGA", mega)
            ctx.profile_read(L.GP_PROF_SYRK)                      # (reading resets the class's counters)
            ctx.profile(1 << L.GP_PROF_SYRK)
            m = RegressionModel(ctx, p["X"], p["y"], p["theta"])  # the shape's first factorisation: launch-per-step in every case
            ctx.sync()
            first[(n, mega)] = ctx.profile_read(L.GP_PROF_SYRK)[0]
            ctx.check(lib.gp_model_refit_dev(m.h, L.dptr(L.f64(p["theta"])), float("nan")))
            ctx.sync()
            launches[(n, mega)] = ctx.profile_read(L.GP_PROF_SYRK)[0]
            ctx.profile(0)
            factors[(n, mega)] = m.L()
            m.close()
    assert first[(6400, None)] == first[(6400, "0")] > 5                    # the first factorisation of a shape: never planned
    assert launches[(6400, None)] == 1 and launches[(6400, "0")] > 5        # by default one launch at 6400 rows ...
    assert launches[(4608, None)] == launches[(4608, "0")] > 5              # ... and the launch-per-step form below 5120
    for n in (6400, 4608):
        assert np.array_equal(factors[(n, None)], factors[(n, "0")])
    ctx.close()


def test_cholesky_single_launch_state_is_per_context_and_per_shape(ctx, monkeypatch):
    """The single launch keeps its plan, flags and claim counter in the context: two contexts refitting models of different sizes in
    turn, with nothing but stream order between the launches of each, and one context going back and forth between two shapes (the
    plan is rebuilt, the flags' epoch goes on) must each give the factor the launch-per-step form gives."""
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import Context, RegressionModel
    other = Context(0)
    try:
        probs = {n: synth.regression(n, 3, 0, 70 + n % 7, 71, 0, np.array([1.1, 0.8, 1.2, 0.9, 0.25])) for n in (5200, 6400)}
        monkeypatch.setenv("GPCORE_CHOL_MEGA", "0")
        want = {}
        for n, p in probs.items():
            m = RegressionModel(ctx, p["X"], p["y"], p["theta"])
            want[n] = (m.L(), m.alpha())
            m.close()
        monkeypatch.setenv("GPCORE_CHOL_MEGA", "1")
        models = {(c, n): RegressionModel(c, probs[n]["X"], probs[n]["y"], probs[n]["theta"]) for c in (ctx, other) for n in probs}
        for _ in range(3):                                   # interleaved: ctx 5200, other 6400, ctx 6400, other 5200, ...
            for (c, n), m in models.items():
                c.check(c._lib.gp_model_refit_dev(m.h, L.dptr(L.f64(probs[n]["theta"])), float("nan")))
        for (c, n), m in models.items():
            assert np.array_equal(m.L(), want[n][0]) and np.array_equal(m.alpha(), want[n][1]), n
            m.close()
    finally:
        other.close()


def test_cholesky_lookahead_on_the_side_stream_is_the_same_factorisation(ctx):
    """The far trailing updates on the CU-masked side stream (default for single factorisations with >= 6144 rows, i.e. the C2 fit)
    against the one-stream form: the same tiles by the same kernels, so L, alpha and the LML must be IDENTICAL -- also when
    several refits are queued back to back -- and L L^T = K."""
    from gp_algos_amd import _lib as L
    from gp_algos_amd.core import RegressionModel
    n, d = 6400, 4
    p = synth.regression(n, d, 0, 51, 52, 0, np.array([1.3, 1.0, 1.4, 0.8, 1.2, 0.15]))
    lib = ctx._lib
    got = {}
    for la in (1, 0):
        ctx.check(lib.gp_ctx_set_lookahead(ctx.h, la))            # la = 1: the single persistent launch (the y^T strip riding along as a 51st row block)
        m = RegressionModel(ctx, p["X"], p["y"], p["theta"])
        for _ in range(3):                                          # refits queued without a host sync in between
            ctx.check(lib.gp_model_refit_dev(m.h, L.dptr(L.f64(p["theta"])), float("nan")))
        got[la] = (m.L(), m.alpha(), m.lml())
        m.close()
    ctx.check(lib.gp_ctx_set_lookahead(ctx.h, -1))
    assert np.array_equal(got[1][0], got[0][0]) and np.array_equal(got[1][1], got[0][1]) and got[1][2] == got[0][2]
    Lf = got[1][0]
    K = orc.gram_sym(p["X"], p["theta"])
    V = np.random.default_rng(3).standard_normal((n, 3))
    assert np.linalg.norm(Lf @ (Lf.T @ V) - K @ V) / np.linalg.norm(K @ V) <= 1e-13
    assert np.all(np.triu(Lf, 1) == 0.0)


# ---- C5 at full size -------------------------------------------------------------------------------------------------------
def test_c5_full_size_fit_and_large_batch_variances(ctx):
    """Config C5 per GPU: n = 32768, d = 8 fit (8.6 GB factor, outer panel 1024) and one 131 072-point posterior batch."""
    from gp_algos_amd.core import RegressionModel
    n, d, m = 32768, 8, 131072
    p = synth.config_c5(n, d, m)
    sf2, sn2 = p["theta"][0] ** 2, p["theta"][-1] ** 2
    mdl = RegressionModel(ctx, p["X"], p["y"], p["theta"])
    alpha = mdl.alpha()
    assert np.all(np.isfinite(alpha))
    Z = p["X"] / p["theta"][1:-1]

    def krow(z):
        return sf2 * np.exp(-0.5 * ((Z - z) ** 2).sum(axis=1))

    rng = np.random.default_rng(0)
    idx = rng.choice(n, 64, replace=False)
    errK = max(abs(krow(Z[i]) @ alpha + sn2 * alpha[i] - p["y"][i]) for i in idx)
    assert errK <= 1e-8
    mean, var, _ = mdl.predict(p["Xs"])
    assert np.all(np.isfinite(mean)) and np.all(var > 0.0) and np.all(var <= sf2 + sn2 + 1e-9)
    Zs = p["Xs"] / p["theta"][1:-1]
    jdx = np.sort(rng.choice(m, 64, replace=False))
    errM = max(abs(krow(Zs[j]) @ alpha - mean[j]) for j in jdx)
    assert errM <= 1e-9 * max(1.0, np.max(np.abs(mean)))
    # the same points through the small-batch (right-looking) path
    m2, v2, _ = mdl.predict(np.asfortranarray(p["Xs"][jdx]))
    assert np.max(np.abs(mean[jdx] - m2)) <= 1e-10 * max(1.0, np.max(np.abs(m2)))
    assert np.max(np.abs(var[jdx] - v2)) <= 1e-10 * sf2
    # LML from its definition with the device's own L diagonal and alpha
    Lh = mdl.L()
    lml = -0.5 * p["y"] @ alpha - np.log(np.diag(Lh)).sum() - 0.5 * n * np.log(2 * np.pi)
    assert abs(mdl.lml() - lml) <= 1e-11 * abs(lml)
    # two points against the oracle's scalar forward substitution (n^2 flops each)
    om, ov, _, _ = orc.predict(p["X"], p["theta"], Lh, alpha, np.asfortranarray(p["Xs"][jdx[:2]]))
    assert np.max(np.abs(mean[jdx[:2]] - om)) <= 1e-9 * max(1.0, np.max(np.abs(om)))
    assert np.max(np.abs(var[jdx[:2]] - ov)) <= 1e-9 * sf2
    del Lh
    mdl.close()
    ctx.trim()
