"""Mirror of object MatrixUtils (utils/MatrixUtils.scala:17-133) on top of libgpcore.so.

GaussianRbfKernel Gram matrices are built by the HIP kernels; any other KernelFunc (e.g. the
reference's Co2Kernel) is evaluated on the host pair by pair, exactly like the Scala loops, and the
resulting matrix is handed to the GPU through the *_from_gram entry points by the callers."""
import numpy as np

from .. import default_context
from .kernel_requisites import GaussianRbfKernel


def _mat(a):
    a = np.asarray(a, dtype=np.float64)
    return np.asfortranarray(a.reshape(-1, 1) if a.ndim == 1 else a)


def buildKernelMatrix(kernelFun, input1, input2=None):
    """buildKernelMatrix(kernelFun, data)           :57-70  symmetric, noise on the diagonal
       buildKernelMatrix(kernelFun, input1, input2) :44-55  cross, never adds noise"""
    X1 = _mat(input1)
    if isinstance(kernelFun, GaussianRbfKernel):
        theta = kernelFun.rbfParams.toDenseVector()
        ctx = default_context()
        if input2 is None:
            return ctx.gram_rbf(X1, theta, full=True)
        return ctx.cross_gram_rbf(X1, _mat(input2), theta)
    from ..gp.regression.co2_prediction import Co2Kernel
    if isinstance(kernelFun, Co2Kernel):          # a user KernelFunc of the reference that also has a device path (gp_*_co2)
        if X1.shape[1] != 1:
            raise ValueError("requirement failed: This kernel is applicable only for 1D objects")
        theta = kernelFun.hyperParams.toDenseVector()
        if input2 is None:
            return default_context().gram_co2(X1[:, 0], theta)
        return default_context().gram_co2(_mat(input2)[:, 0], theta, xs=X1[:, 0])
    if input2 is None:
        return buildMatrixWithFunc(X1)(lambda a, b, same: kernelFun.apply(a, b, same))
    return buildMatrixWithFunc(lambda a, b: kernelFun.apply(a, b, False), X1, _mat(input2))


def buildMatrixWithFunc(*args):
    """buildMatrixWithFunc(data)(f(vec1, vec2, sameIndex))   :72-84  (curried, symmetric)
       buildMatrixWithFunc(func(vec1, vec2), input1, input2)  :86-97"""
    if len(args) == 1:
        data = _mat(args[0])

        def build(f):
            tag = getattr(f, "_gpcore_rbf", None)
            if tag is not None:      # f = GaussianRbfKernel.derAfterHyperParam(p): gp_dgram_rbf builds dK/dtheta_p on the device
                kernel, pos = tag
                return default_context().dgram_rbf(data, kernel.rbfParams.toDenseVector(), pos)
            tag = getattr(f, "_gpcore_co2", None)
            if tag is not None:      # f = Co2Kernel.derAfterHyperParam(p): gp_dgram_co2
                kernel, pos = tag
                return default_context().gram_co2(data[:, 0], kernel.hyperParams.toDenseVector(), pos=pos)
            n = data.shape[0]
            out = np.zeros((n, n), order="F")
            for i in range(n):
                for j in range(i + 1):
                    v = f(data[i, :], data[j, :], i == j)
                    out[i, j] = v
                    out[j, i] = v
            return out
        return build
    func, a, b = args
    a, b = _mat(a), _mat(b)
    out = np.zeros((a.shape[0], b.shape[0]), order="F")
    for i in range(a.shape[0]):
        for j in range(b.shape[0]):
            out[i, j] = func(a[i, :], b[j, :])
    return out


def _require_square(M):
    M = np.asarray(M, dtype=np.float64)
    if M.ndim != 2 or M.shape[0] != M.shape[1]:
        raise ValueError("requirement failed: L.rows == L.cols")   # :125
    return M


def forwardSolve(L, b):
    """forwardSolve(L, b: vector | matrix)  :17-21, :29-31"""
    return default_context().trsm_lower(_require_square(L), b, trans=False)


def backSolve(R, b):
    """backSolve(R, b) with R upper triangular (the reference passes L.t)  :23-27, :33-35"""
    R = _require_square(R)
    return default_context().trsm_lower(np.asfortranarray(R.T), b, trans=True)


def invTriangular(matrix, isUpper):   # :106-113
    M = _require_square(matrix)
    if isUpper:
        return np.asfortranarray(default_context().inv_lower(np.asfortranarray(M.T)).T)
    return default_context().inv_lower(M)


def cloneCols(vec, colNum):   # :37-42
    return np.asfortranarray(np.repeat(np.asarray(vec, dtype=np.float64).reshape(-1, 1), colNum, axis=1))


def rowScale(dv, m):
    """implicit `dv :* m`: row i of m scaled by dv(i)  (ElementWiseMultDenseVector :143-152)."""
    return np.asarray(m, dtype=np.float64) * np.asarray(dv, dtype=np.float64).reshape(-1, 1)


def intDivVector(i, vec):
    """implicit `i / vec`  (IntDividingVector :135-141)."""
    return i / np.asarray(vec, dtype=np.float64)
