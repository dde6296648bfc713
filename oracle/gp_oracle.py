"""ctypes front-end of the CPU parity oracle (oracle/gp_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (gp_algos_amd) never does.  Every function takes/returns numpy float64 arrays in
column-major (Fortran) order, matching Breeze's DenseMatrix layout.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgporacle.so")
_lib = None

_d = C.c_double
_i = C.c_int
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    """Compile oracle/gp_oracle.c with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "gp_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_dnorm.restype = _d
        _lib.orc_dnorm.argtypes = [_d]
        _lib.orc_pnorm.restype = _d
        _lib.orc_pnorm.argtypes = [_d]
        _lib.orc_rbf_kernel.restype = _d
        _lib.orc_lml.restype = _d
        _lib.orc_ep_lml.restype = _d
        _lib.orc_avg_between_site_params.restype = _d
        _lib.orc_co2_kernel.restype = _d
        _lib.orc_co2_kernel.argtypes = [_d, _d, _i, C.POINTER(C.c_double)]
    return _lib


def _f(a):
    """float64, Fortran-contiguous, 2-D or 1-D view suitable for passing by pointer."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp)


def _ld(a):
    return _i(a.shape[0] if a.ndim == 2 else a.size)


def dnorm(x):
    return lib().orc_dnorm(_d(x))


def pnorm(x):
    return lib().orc_pnorm(_d(x))


def rbf_kernel(x, y, theta, same):
    x, y, theta = _f(x), _f(y), _f(theta)
    return lib().orc_rbf_kernel(_p(x), _p(y), _i(x.size), _p(theta), _i(int(same)))


def hp_get_at_position(theta, pos):
    """1-based GaussianRbfParams.getAtPosition; raises IndexError where Scala throws MatchError."""
    theta = _f(theta)
    out = _d()
    rc = lib().orc_hp_get_at_position(_p(theta), _i(theta.size - 2), _i(pos), C.byref(out))
    if rc:
        raise IndexError("MatchError: position %d of %d" % (pos, theta.size))
    return out.value


def gram_sym(X, theta):
    X, theta = _f(X), _f(theta)
    n, d = X.shape
    K = np.zeros((n, n), order="F")
    lib().orc_gram_sym(_p(X), _i(n), _i(d), _ld(X), _p(theta), _p(K), _i(n))
    return K


def gram_cross(Xs, X, theta):
    Xs, X, theta = _f(Xs), _f(X), _f(theta)
    m, d = Xs.shape
    n = X.shape[0]
    Ks = np.zeros((m, n), order="F")
    lib().orc_gram_cross(_p(Xs), _i(m), _ld(Xs), _p(X), _i(n), _ld(X), _i(d), _p(theta), _p(Ks), _i(m))
    return Ks


def dgram_sym(X, theta, p):
    """d K / d theta_p, p 1-BASED as in the reference."""
    X, theta = _f(X), _f(theta)
    n, d = X.shape
    D = np.zeros((n, n), order="F")
    lib().orc_dgram_sym(_p(X), _i(n), _i(d), _ld(X), _p(theta), _i(p), _p(D), _i(n))
    return D


def forward_solve(L, b, trans=False):
    L, b = _f(L), _f(b)
    n = L.shape[0]
    x = np.zeros_like(b, order="F")
    if b.ndim == 1:
        lib().orc_forward_solve_vec(_p(L), _i(n), _ld(L), _i(int(trans)), _p(b), _p(x))
    else:
        lib().orc_forward_solve_mat(_p(L), _i(n), _ld(L), _i(int(trans)), _p(b), _i(b.shape[1]), _ld(b), _p(x), _ld(x))
    return x


def back_solve(R, b, trans=False):
    R, b = _f(R), _f(b)
    n = R.shape[0]
    x = np.zeros_like(b, order="F")
    if b.ndim == 1:
        lib().orc_back_solve_vec(_p(R), _i(n), _ld(R), _i(int(trans)), _p(b), _p(x))
    else:
        lib().orc_back_solve_mat(_p(R), _i(n), _ld(R), _i(int(trans)), _p(b), _i(b.shape[1]), _ld(b), _p(x), _ld(x))
    return x


def inv_triangular(T, is_upper):
    T = _f(T)
    n = T.shape[0]
    Ti = np.zeros((n, n), order="F")
    rc = lib().orc_inv_triangular(_p(T), _i(n), _ld(T), _i(int(is_upper)), _p(Ti), _i(n))
    assert rc == 0
    return Ti


class NotPositiveDefinite(Exception):
    def __init__(self, info):
        super().__init__("matrix not positive definite at pivot %d" % info)
        self.info = info


def cholesky_lower(A):
    L = np.array(A, dtype=np.float64, order="F", copy=True)
    n = L.shape[0]
    info = _i()
    rc = lib().orc_cholesky_lower(_p(L), _i(n), _i(n), C.byref(info))
    if rc:
        raise NotPositiveDefinite(info.value)
    return L


def fit(X, y, theta, sigma_noise=None):
    X, y, theta = _f(X), _f(y), _f(theta)
    n, d = X.shape
    L = np.zeros((n, n), order="F")
    alpha = np.zeros(n)
    info = _i()
    sn = float("nan") if sigma_noise is None else float(sigma_noise)
    rc = lib().orc_fit(_p(X), _i(n), _i(d), _ld(X), _p(y), _p(theta), _d(sn), _p(L), _i(n), _p(alpha), C.byref(info))
    if rc == 2:
        raise NotPositiveDefinite(info.value)
    assert rc == 0
    return L, alpha


def lml(L, alpha, y):
    L, alpha, y = _f(L), _f(alpha), _f(y)
    return lib().orc_lml(_p(L), _i(L.shape[0]), _ld(L), _p(alpha), _p(y))


def predict(X, theta, L, alpha, Xs, full_cov=False, want_v=False):
    """Returns (mean, var_diag, cov or None, V or None)."""
    X, theta, L, alpha, Xs = _f(X), _f(theta), _f(L), _f(alpha), _f(Xs)
    n, d = X.shape
    m = Xs.shape[0]
    mean = np.zeros(m)
    var = np.zeros(m)
    cov = np.zeros((m, m), order="F") if full_cov else None
    V = np.zeros((n, m), order="F") if want_v else None
    rc = lib().orc_predict(_p(X), _i(n), _i(d), _ld(X), _p(theta), _p(L), _ld(L), _p(alpha), _p(Xs), _i(m), _ld(Xs),
                           _p(mean), _p(var), _p(cov) if full_cov else None, _i(m),
                           _p(V) if want_v else None, _i(n))
    assert rc == 0
    return mean, var, cov, V


def lml_grad(X, y, theta, nparams=None, sigma_noise=None):
    X, y, theta = _f(X), _f(y), _f(theta)
    n, d = X.shape
    P = d + 2 if nparams is None else int(nparams)
    out = _d()
    grad = np.zeros(P)
    info = _i()
    sn = float("nan") if sigma_noise is None else float(sigma_noise)
    rc = lib().orc_lml_grad(_p(X), _i(n), _i(d), _ld(X), _p(y), _p(theta), _d(sn), _i(P), C.byref(out), _p(grad), C.byref(info))
    if rc == 2:
        raise NotPositiveDefinite(info.value)
    assert rc == 0
    return out.value, grad


def kernel_gradient(v1, v2, theta, after_first_arg):
    """GaussianRbfKernel.gradient(afterFirstArg)(vec1, vec2), KernelRequisites.scala:95-107."""
    v1, v2, theta = _f(v1), _f(v2), _f(theta)
    out = np.zeros(v1.size)
    lib().orc_kernel_gradient(_p(v1), _p(v2), _i(v1.size), _p(theta), _i(1 if after_first_arg else 0), _p(out))
    return out


def ucb(X, theta, L, alpha, x, kappa):
    """GPOptimizer.maximizeUCB's objective and gradient at one test point (GPOptimizer.scala:82-109): (value, grad[d])."""
    X, theta, L, alpha, x = _f(X), _f(theta), _f(L), _f(alpha), _f(x)
    n, d = X.shape
    val = _d()
    grad = np.zeros(d)
    rc = lib().orc_ucb(_p(X), _i(n), _i(d), _ld(X), _p(theta), _p(L), _ld(L), _p(alpha), _p(x), _d(float(kappa)), C.byref(val), _p(grad))
    assert rc == 0
    return val.value, grad


def co2_kernel(x1, x2, same, hp):
    """Co2Kernel.apply, gp/regression/Co2Prediction.scala:39-56."""
    hp = _f(hp)
    return lib().orc_co2_kernel(_d(float(x1)), _d(float(x2)), _i(1 if same else 0), _p(hp))


def co2_gram(x, hp, xs=None, pos=0):
    """Gram (xs None: symmetric with the noise flag, mirrored) / cross-Gram / derivative Gram (pos = 1..11) of the Co2Kernel."""
    x, hp = _f(np.asarray(x, dtype=np.float64).reshape(-1)), _f(hp)
    n = x.size
    if xs is None:
        K = np.zeros((n, n), order="F")
        rc = lib().orc_co2_gram(_p(x), _i(n), None, _i(0), _p(hp), _i(pos), _p(K), _i(n))
    else:
        xs = _f(np.asarray(xs, dtype=np.float64).reshape(-1))
        K = np.zeros((xs.size, n), order="F")
        rc = lib().orc_co2_gram(_p(x), _i(n), _p(xs), _i(xs.size), _p(hp), _i(pos), _p(K), _i(xs.size))
    if rc:
        raise IndexError("hyper-parameter position %d outside 1..11 (MatchError)" % pos)
    return K


def marginal_moments(cav_mi, cav_sigma, target):
    a, b = _d(), _d()
    lib().orc_marginal_moments(_d(cav_mi), _d(cav_sigma), _i(int(target)), C.byref(a), C.byref(b))
    return a.value, b.value


def avg_between_site_params(old_tau, old_nu, cur_tau, cur_nu):
    a, b, c, d = _f(old_tau), _f(old_nu), _f(cur_tau), _f(cur_nu)
    return lib().orc_avg_between_site_params(_p(a), _p(b), _p(c), _p(d), _i(a.size))


def ep_estimate(K, y, max_sweeps, eps=-1.0):
    """Literal EP (EpParameterEstimator.estimateSiteParams).  eps<0: run exactly max_sweeps sweeps.
    Returns dict(tau, nu, cav_tau, cav_nu, Sigma, mu, L, sweeps)."""
    K = _f(K)
    n = K.shape[0]
    yi = np.ascontiguousarray(y, dtype=np.int32)
    o = {k: np.zeros(n) for k in ("tau", "nu", "cav_tau", "cav_nu", "mu")}
    o["Sigma"] = np.zeros((n, n), order="F")
    o["L"] = np.zeros((n, n), order="F")
    sw, info = _i(), _i()
    rc = lib().orc_ep_estimate(_p(K), _i(n), _ld(K), yi.ctypes.data_as(_ip), _i(max_sweeps), _d(eps), _p(o["tau"]),
                               _p(o["nu"]), _p(o["cav_tau"]), _p(o["cav_nu"]), _p(o["Sigma"]), _p(o["mu"]),
                               _p(o["L"]), C.byref(sw), C.byref(info))
    if rc == 2:
        raise NotPositiveDefinite(info.value)
    assert rc == 0
    o["sweeps"] = sw.value
    return o


def ep_lml(ep, y, strict=True):
    n = ep["tau"].size
    yi = np.ascontiguousarray(y, dtype=np.int32)
    return lib().orc_ep_lml(_p(ep["Sigma"]), _i(n), _i(n), _p(ep["L"]), _i(n), _p(ep["tau"]), _p(ep["nu"]),
                            _p(ep["cav_tau"]), _p(ep["cav_nu"]), yi.ctypes.data_as(_ip), _i(int(strict)))


def ep_classify(K, L, tau, nu, Ks, kss_diag):
    K, L, tau, nu, Ks, kd = _f(K), _f(L), _f(tau), _f(nu), _f(Ks), _f(kss_diag)
    n, m = K.shape[0], Ks.shape[0]
    prob, fm, fv = np.zeros(m), np.zeros(m), np.zeros(m)
    rc = lib().orc_ep_classify(_p(K), _i(n), _ld(K), _p(L), _ld(L), _p(tau), _p(nu), _p(Ks), _i(m), _ld(Ks), _p(kd),
                               _p(prob), _p(fm), _p(fv))
    assert rc == 0
    return prob, fm, fv


def ep_lml_grad(X, theta, K, L, tau, nu, strict=True):
    X, theta, K, L, tau, nu = _f(X), _f(theta), _f(K), _f(L), _f(tau), _f(nu)
    n, d = X.shape
    g = np.zeros(d + 2)
    rc = lib().orc_ep_lml_grad(_p(X), _i(n), _i(d), _ld(X), _p(theta), _p(K), _ld(K), _p(L), _ld(L), _p(tau), _p(nu),
                               _i(int(strict)), _p(g))
    assert rc == 0
    return g
