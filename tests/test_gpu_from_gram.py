"""gp_lml_grad_from_gram: GpPredictor.logLikelihoodWithDerivatives (gp/regression/GpPredictor.scala:60-80) for ANY KernelFunc --
kernel matrix and derivative matrices evaluated on the HOST with the reference's loops, factorisation / K^-1 / traces on the
device ("any other kernel -> host-built Gram through gp_*_from_gram").  The arbitrary kernel here is the reference's
own Co2Kernel (gp/regression/Co2Prediction.scala:29-137) evaluated on the host by the oracle, so the result can be compared with the
device-native gp_lml_grad_co2_batched as well as with the reference formula.  Tolerance 1e-7 relative on the gradient, 1e-10 on
the LML.  Parity unpinned against the JVM (no reference test holds an LML gradient): oracle + formula only."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import gp_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "co2")
HP0 = np.array([60., 70., 8., 50., 2., 0.34, 2.4, 0.88, 0.26, 0.2, 0.19])      # utils/TestingUtils.scala:17-20


@pytest.fixture(scope="module")
def ctx():
    from gp_algos_amd.core import Context
    c = Context(0)
    yield c
    c.close()


def _mauna(ratio):
    from gp_algos_amd.gp.regression.co2_prediction import co2DataToYearWithValue, loadInput
    return co2DataToYearWithValue(loadInput(os.path.join(GOLD, "maunaLoa.txt")), ratio)


def _reference_formula(K, y, dKs):
    Lo = orc.cholesky_lower(K)
    ao = orc.back_solve(Lo, orc.forward_solve(Lo, y), trans=True)
    Li = orc.inv_triangular(Lo, False)
    W = np.outer(ao, ao) - Li.T @ Li                                # alphaSq - inversedK (:66-69)
    return orc.lml(Lo, ao, y), np.array([0.5 * np.trace(W @ D) for D in dKs])     # :76


@pytest.mark.parametrize("ratio,nparams", [(0.25, 11), (0.5, 10)])      # n = 151 (np = 256) and n = 303 (np = 384)
def test_host_built_co2_kernel_matches_device_co2_path_and_reference_formula(ctx, ratio, nparams):
    train, _ = _mauna(ratio)
    x, y = train[:, 0], train[:, 1]
    theta = HP0 * np.array([1.1, 0.9, 1.2, 1.0, 0.8, 1.3, 1.0, 1.1, 0.9, 1.2, 1.4])
    K = orc.co2_gram(x, theta)                                      # host: buildKernelMatrix(kernelFunc, trainingData) :62
    dKs = [orc.co2_gram(x, theta, pos=p) for p in range(1, nparams + 1)]           # host: buildMatrixWithFunc(...)(derAfterHyperParam(p)) :74
    lml, grad = ctx.lml_grad_from_gram(K, y, dKs)
    olml, ograd = _reference_formula(K, y, dKs)
    assert abs(lml - olml) <= 1e-10 * abs(olml)
    assert np.max(np.abs(grad - ograd)) <= 1e-7 * np.max(np.abs(ograd))
    dl, dg, info = ctx.lml_grad_co2_batched(x, y, theta[None, :], nparams=nparams)          # the device-native route of the same kernel
    assert info[0] == 0 and abs(lml - dl[0]) <= 1e-10 * abs(lml)
    assert np.max(np.abs(grad - dg[0])) <= 1e-7 * np.max(np.abs(grad))


def test_sigma_noise_is_added_unsquared_and_lml_only_call(ctx):
    train, _ = _mauna(0.2)
    x, y = train[:, 0], train[:, 1]
    K = orc.co2_gram(x, HP0)
    dKs = [orc.co2_gram(x, HP0, pos=p) for p in (1, 2, 11)]
    sn = 0.37
    lml, grad = ctx.lml_grad_from_gram(K, y, dKs, sigma_noise=sn)
    olml, ograd = _reference_formula(K + sn * np.eye(x.size), y, dKs)          # GpPredictor.scala:116: + sigmaNoise * I, not squared
    assert abs(lml - olml) <= 1e-10 * abs(olml) and np.max(np.abs(grad - ograd)) <= 1e-7 * np.max(np.abs(ograd))
    l0, g0 = ctx.lml_grad_from_gram(K, y, [], sigma_noise=sn)                 # optimizedParamsNum = 0: LML alone
    assert l0 == lml and g0.shape == (0,)


def test_leading_dimensions_and_errors_through_the_raw_abi(ctx):
    from gp_algos_amd import _lib as L
    rng = np.random.default_rng(5)
    n, ldk, ldd = 70, 75, 81
    A = rng.standard_normal((n, n))
    K = A @ A.T + n * np.eye(n)
    D1 = rng.standard_normal((n, n)); D1 = D1 + D1.T
    D2 = rng.standard_normal((n, n)); D2 = D2 + D2.T
    y = rng.standard_normal(n)
    Kb = np.full((ldk, n), np.nan, order="F"); Kb[:n] = K
    Db = [np.full((ldd, n), np.nan, order="F") for _ in range(2)]
    Db[0][:n], Db[1][:n] = D1, D2
    ptrs = (C.POINTER(C.c_double) * 2)(L.dptr(Db[0]), L.dptr(Db[1]))
    lml, grad, info = C.c_double(), np.zeros(2), C.c_int(-1)
    st = ctx._lib.gp_lml_grad_from_gram(ctx.h, L.dptr(Kb), n, ldk, L.dptr(y), ptrs, 2, ldd, float("nan"), C.byref(lml), L.dptr(grad), C.byref(info))
    assert st == L.GP_OK and info.value == 0
    olml, ograd = _reference_formula(K, y, [D1, D2])
    assert abs(lml.value - olml) <= 1e-10 * abs(olml) and np.max(np.abs(grad - ograd)) <= 1e-8 * np.max(np.abs(ograd))
    # not positive definite: GP_ENOTPD with the failing pivot, like breeze.linalg.cholesky throwing (:120)
    Kbad = K.copy(); Kbad[40, 40] = -1.0
    with pytest.raises(L.NotPositiveDefinite) as ei:
        ctx.lml_grad_from_gram(Kbad, y, [D1])
    assert ei.value.info == 41
    # argument errors
    assert ctx._lib.gp_lml_grad_from_gram(ctx.h, L.dptr(Kb), n, n - 1, L.dptr(y), ptrs, 2, ldd, float("nan"), C.byref(lml), L.dptr(grad), None) == L.GP_EINVAL
    assert ctx._lib.gp_lml_grad_from_gram(ctx.h, L.dptr(Kb), n, ldk, L.dptr(y), None, 2, ldd, float("nan"), C.byref(lml), L.dptr(grad), None) == L.GP_EINVAL
    nullp = (C.POINTER(C.c_double) * 2)(L.dptr(Db[0]), None)
    assert ctx._lib.gp_lml_grad_from_gram(ctx.h, L.dptr(Kb), n, ldk, L.dptr(y), nullp, 2, ldd, float("nan"), C.byref(lml), L.dptr(grad), None) == L.GP_EINVAL
    with pytest.raises(ValueError):
        ctx.lml_grad_from_gram(K, y[:-1], [D1])


class _OpaqueKernel:
    """A user KernelFunc the shim knows nothing about (it is NOT an instance of GaussianRbfKernel / Co2Kernel): a product of two
    squared-exponentials with hyper-parameters (a, l1, l2, noise) and hand-written derivatives, the way a user of the reference
    would write one (trait KernelFunc, utils/KernelRequisites.scala:28-36)."""

    class HP:
        def __init__(self, dv):
            self.dv = np.array(dv, dtype=np.float64)

        def toDenseVector(self):
            return self.dv.copy()

        def fromDenseVector(self, dv):
            return _OpaqueKernel.HP(dv)

        def getAtPosition(self, i):
            return float(self.dv[i - 1])

    def __init__(self, hp):
        self.hp = hp

    hyperParametersNum = 4

    @property
    def hyperParams(self):
        return self.hp

    def changeHyperParams(self, dv):
        return _OpaqueKernel(_OpaqueKernel.HP(dv))

    def _e(self, a, b):
        p = self.hp.dv
        return np.exp(-0.5 * (a[0] - b[0]) ** 2 / p[1] ** 2 - 0.5 * (a[1] - b[1]) ** 2 / p[2] ** 2)

    def apply(self, a, b, same):
        p = self.hp.dv
        return p[0] * p[0] * self._e(a, b) + (p[3] * p[3] if same else 0.0)

    def derAfterHyperParam(self, k):
        p = self.hp.dv

        def f(a, b, same):
            e = self._e(a, b)
            if k == 1:
                return 2 * p[0] * e
            if k in (2, 3):
                return p[0] * p[0] * e * (a[k - 2] - b[k - 2]) ** 2 / p[k - 1] ** 3
            return 2 * p[3] if same else 0.0
        return f


def test_mirror_routes_an_unknown_kernel_through_from_gram(ctx):
    """GpPredictor(kernelFunc) with a KernelFunc that has no device form: logLikelihoodWithDerivatives and
    obtainOptimalHyperParams (gp/regression/GpPredictor.scala:60-80,126-142) run, the LML matches the ARD-RBF device path (the opaque
    kernel IS an ARD-RBF in disguise) and the optimiser improves it."""
    import gp_algos_amd
    from gp_algos_amd import synth
    from gp_algos_amd.gp.regression.gp_predictor import GpPredictor, PredictionTrainingInput
    gp_algos_amd.set_default_context(ctx)
    p = synth.regression(90, 2, 0, 71, 72, 0, synth.ard_theta(2, 1.3, 0.8, 0.2))
    th = np.array([1.3, 0.8, 1.1, 0.2])
    pred = GpPredictor(_OpaqueKernel(_OpaqueKernel.HP(th)))
    ti = PredictionTrainingInput(p["X"], None, p["y"])
    lml, grad = pred.logLikelihoodWithDerivatives(ti, _OpaqueKernel.HP(th), 4)
    ol, og = orc.lml_grad(p["X"], p["y"], th)
    assert abs(lml - ol) <= 1e-10 * abs(ol) and np.max(np.abs(grad - og)) <= 1e-7 * np.max(np.abs(og))
    l3, g3 = pred.logLikelihoodWithDerivatives(ti, _OpaqueKernel.HP(th), 3)
    assert l3 == lml and g3.shape == (3,) and np.array_equal(g3, grad[:3])
    best = pred.obtainOptimalHyperParams(p["X"], None, p["y"], True)
    lbest, _ = pred.logLikelihoodWithDerivatives(ti, best, 4)
    assert lbest > lml + 0.1
