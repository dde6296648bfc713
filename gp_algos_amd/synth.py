"""Deterministic synthetic inputs for the BASELINE.md configurations C1..C5.

u(seed, i) = (splitmix64(seed * 0x9E3779B97F4A7C15 + i) >> 11) * 2^-53 ; normals by Box-Muller on
(u(seed, 2i), u(seed, 2i+1)).  All matrices are column-major (Fortran order), fp64.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = 0x9E3779B97F4A7C15


def _splitmix64(x):
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(_GOLD)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def u(seed, idx):
    """Uniform [0,1) stream `seed` at integer indices idx (array-like)."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64((seed * _GOLD) & 0xFFFFFFFFFFFFFFFF) + idx
    return (_splitmix64(base) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def normal(seed, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    u1 = u(seed, 2 * idx)
    u2 = u(seed, 2 * idx + 1)
    u1 = np.where(u1 <= 0.0, 2.0 ** -53, u1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def _xmat(seed, n, d, lo, span):
    i = np.arange(n, dtype=np.uint64)[:, None]
    k = np.arange(d, dtype=np.uint64)[None, :]
    return np.asfortranarray(lo + span * u(seed, i * np.uint64(d) + k))


def _ytrend(X):
    d = X.shape[1]
    k = np.arange(1, d + 1, dtype=np.float64)[None, :]
    return np.sin(k * X / 2.0).sum(axis=1) / np.sqrt(d)


def ard_theta(d, sf, scale, sn):
    return np.concatenate(([sf], scale * (1.0 + 0.25 * np.arange(d)), [sn]))


def config_c1():
    n, m = 256, 100
    x = -5.0 + 10.0 * u(1, np.arange(n))
    y = np.sin(x) + 0.1 * normal(2, np.arange(n))
    xs = -6.0 + 12.0 * np.arange(m) / 99.0
    return dict(X=np.asfortranarray(x[:, None]), y=y, Xs=np.asfortranarray(xs[:, None]),
                theta=np.array([1.0, 1.0, 0.1]))


def regression(n, d, m, seed_x, seed_y, seed_xs, theta):
    """C2/C3/C5-shaped regression problem."""
    X = _xmat(seed_x, n, d, -2.0, 4.0)
    y = _ytrend(X) + 0.1 * normal(seed_y, np.arange(n))
    Xs = _xmat(seed_xs, m, d, -2.0, 4.0) if m else None
    return dict(X=X, y=y, Xs=Xs, theta=np.asarray(theta, dtype=np.float64))


def config_c2(n=8192, d=8, m=65536):
    return regression(n, d, m, 11, 12, 13, ard_theta(d, 1.5, 1.0, 0.1))


def config_c3(n=4096, d=8):
    """X, y plus the 4x4x4 hyper-parameter grid, index b = (i_sf*4 + i_s)*4 + i_sn."""
    p = regression(n, d, 0, 21, 22, 0, ard_theta(d, 1.0, 1.0, 0.1))
    thetas = []
    for sf in (0.5, 1.0, 1.5, 2.0):
        for s in (0.5, 1.0, 2.0, 4.0):
            for sn in (0.05, 0.1, 0.2, 0.4):
                thetas.append(ard_theta(d, sf, s, sn))
    p["thetas"] = np.ascontiguousarray(np.stack(thetas))  # B x P, row b = setting b
    return p


def config_c4(n=4096, d=8):
    X = _xmat(31, n, d, -2.0, 4.0)
    f = X.sum(axis=1) / np.sqrt(d) + 0.3 * normal(32, np.arange(n))
    y = np.where(f >= 0.0, 1, -1).astype(np.int32)
    theta = np.concatenate(([2.0], 2.0 * np.ones(d), [0.0]))
    return dict(X=X, y=y, theta=theta)


def config_c5(n=32768, d=8, m=1000000):
    return regression(n, d, m, 41, 42, 43, ard_theta(d, 1.5, 1.0, 0.2))
