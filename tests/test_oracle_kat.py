"""Pins the CPU oracle against every known-answer vector the reference's own tests hold for the hot
path:  src/test/scala/utils/MatrixUtilsTest.scala:24-114 and
src/test/scala/utils/KernelRequisitesTest.scala:20-47.  Literals below are the test inputs of those
files; expected values are what the Scala asserts (Breeze `\\`, `inv`, exact 1.0 diagonal)."""
import numpy as np
import pytest
import scipy.linalg as sla

from oracle import gp_oracle as orc

LOWER = np.array([[0.3, 0.0, 0.0], [0.2, 0.3, 0.0], [0.1, 0.99, 0.11]])
UPPER = np.array([[0.4, 0.1, 0.9], [0.0, 0.2, 0.89], [0.0, 0.0, 0.5]])
EPS = 1e-3  # MatrixUtilsTest.scala:22
X3 = np.array([[2.4, 1.3, 1.9], [2.1, 0.99, 3.1], [1.89, 2.01, 4.0]])
THETA3 = np.array([1.0, 1.0, 1.0, 1.0, 0.0])


def test_forward_solve_vector_kat():  # MatrixUtilsTest.scala:29-36 (asserted `==` exact AND < eps)
    x = orc.forward_solve(LOWER, np.array([3.0, 2.0, 1.0]))
    ref = sla.solve_triangular(LOWER, [3.0, 2.0, 1.0], lower=True)
    assert np.max(np.abs(x - ref)) < EPS
    np.testing.assert_allclose(x, [10.0, 0.0, 0.0], atol=1e-12)
    assert x[0] == 10.0


def test_back_solve_vector_kat():  # :38-44
    x = orc.back_solve(UPPER, np.array([7.0, 3.0, 4.0]))
    np.testing.assert_allclose(x, [4.65, -20.6, 8.0], atol=1e-12)


def test_forward_solve_matrix_kat():  # :46-54
    rhs = np.array([[0.4, 0.9], [0.8, 0.3], [0.7, 0.4]])
    x = orc.forward_solve(LOWER, rhs)
    assert x.shape == (3, 2)
    ref = sla.solve_triangular(LOWER, rhs, lower=True)
    assert np.max(np.abs(x - ref)) < EPS
    np.testing.assert_allclose(x, [[4 / 3, 3.0], [16 / 9, -1.0], [-10.848484848484848, 9.909090909090908]], rtol=1e-12)


def test_back_solve_matrix_kat():  # :56-63
    rhs = np.array([[0.4, 0.9], [0.8, 0.3], [0.7, 0.4]])
    x = orc.back_solve(UPPER, rhs)
    np.testing.assert_allclose(x, [[-1.5925, 0.965], [-2.23, -2.06], [1.4, 0.8]], rtol=1e-12)


def test_gram_kat():  # :90-102 : diagonal == 1.0 EXACTLY, Cholesky succeeds
    K = orc.gram_sym(X3, THETA3)
    assert K.shape == (3, 3)
    for i in range(3):
        assert K[i, i] == 1.0
    np.testing.assert_allclose(K[1, 0], 0.4435033161650128, rtol=1e-14)
    np.testing.assert_allclose(K[2, 0], 0.07523791396600814, rtol=1e-14)
    np.testing.assert_allclose(K[2, 1], 0.387806024974919, rtol=1e-14)
    assert np.array_equal(K, K.T)
    L = orc.cholesky_lower(K)
    np.testing.assert_allclose(L, [[1, 0, 0], [0.4435033161650128, 0.8962727311207436, 0],
                                   [0.07523791396600814, 0.3954574855651918, 0.9153975275324375]], rtol=1e-13)
    assert np.all(np.triu(L, 1) == 0.0)


def test_inv_triangular_kat():  # :104-114 : invTriangular(L)^T invTriangular(L) ~= inv(K) to 1e-3
    K = orc.gram_sym(X3, THETA3)
    L = orc.cholesky_lower(K)
    Li = orc.inv_triangular(L, is_upper=False)
    Kinv = Li.T @ Li
    assert np.max(np.abs(Kinv - np.linalg.inv(K))) < EPS
    np.testing.assert_allclose(Kinv[0], [1.262170376455613, -0.61551966270158, 0.14373916749199114], rtol=1e-12)
    np.testing.assert_allclose(Kinv[1, 1:], [1.4771845232644265, -0.5265506426949201], rtol=1e-12)
    np.testing.assert_allclose(Kinv[2, 2], 1.1933848765741977, rtol=1e-12)


def test_hyper_param_positions_kat():  # KernelRequisitesTest.scala:20-35 : 1-based, MatchError past end
    theta = np.array([1.0, 5.0, 2.0, 3.0, 0.0])
    assert [orc.hp_get_at_position(theta, p) for p in range(1, 6)] == [1.0, 5.0, 2.0, 3.0, 0.0]
    with pytest.raises(IndexError):
        orc.hp_get_at_position(theta, 6)
    with pytest.raises(IndexError):
        orc.hp_get_at_position(theta, 0)


def test_cholesky_not_pd():
    A = np.array([[1.0, 2.0], [2.0, 1.0]])
    with pytest.raises(orc.NotPositiveDefinite) as e:
        orc.cholesky_lower(A)
    assert e.value.info == 2


def test_cholesky_vs_lapack():
    rng = np.random.default_rng(0)
    X = rng.uniform(-2, 2, (96, 3))
    theta = np.array([1.3, 0.7, 1.1, 2.0, 0.2])
    K = orc.gram_sym(X, theta)
    L = orc.cholesky_lower(K)
    Lref = sla.cholesky(K, lower=True)
    assert np.linalg.norm(L - Lref) / np.linalg.norm(Lref) < 1e-13
    assert np.linalg.norm(L @ L.T - K) / np.linalg.norm(K) < 1e-15
