"""Thin object layer over the C-ABI: Context (device + stream) and RegressionModel (L, alpha in HBM)."""
import ctypes as C

import numpy as np

from . import _lib as L


class Context:
    def __init__(self, device=0, stream=None):
        self._lib = L.load()
        h = C.c_void_p()
        st = self._lib.gp_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if st != L.GP_OK:
            raise L.GpCoreError(st, "gp_ctx_create(device=%d) failed: no usable gfx950 GPU (no CPU fallback)" % device)
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self._lib.gp_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, st, info=0):
        if st == L.GP_OK:
            return
        msg = self._lib.gp_last_error(self.h).decode()
        if st == L.GP_ENOTPD:
            raise L.NotPositiveDefinite(st, msg, info)
        if st == L.GP_EINVAL:
            raise ValueError(msg)
        if st == L.GP_ERANGE:
            raise IndexError(msg)
        if st == L.GP_EPEER:
            raise L.PeerFailure(st, msg, info)
        raise L.GpCoreError(st, msg, info)

    def trim(self):
        """Release cached device workspaces (gp_ctx_trim)."""
        self.check(self._lib.gp_ctx_trim(self.h))

    def sync(self):
        self.check(self._lib.gp_ctx_sync(self.h))

    # -- profiling --------------------------------------------------------------------------
    def profile(self, which):
        self.check(self._lib.gp_ctx_profile(self.h, which))

    def profile_read(self, which):
        n, ms, work = C.c_int64(), C.c_double(), C.c_double()
        self.check(self._lib.gp_ctx_profile_read(self.h, which, C.byref(n), C.byref(ms), C.byref(work)))
        return n.value, ms.value, work.value

    def probe_mfma_f64(self):
        v = C.c_double()
        self.check(self._lib.gp_probe_mfma_f64(self.h, C.byref(v)))
        return v.value

    def probe_mfma_f64_ex(self, waves_per_simd=1):
        v, clk, cyc = C.c_double(), C.c_double(), C.c_double()
        self.check(self._lib.gp_probe_mfma_f64_ex(self.h, waves_per_simd, C.byref(v), C.byref(clk), C.byref(cyc)))
        return dict(tflops=v.value, clock_mhz=clk.value, cycles_per_mfma=cyc.value)

    # -- raw device memory --------------------------------------------------------------------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self.check(self._lib.gp_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def dev_free(self, p):
        self.check(self._lib.gp_dev_free(self.h, p))

    def upload(self, host):
        host = np.ascontiguousarray(host) if host.ndim == 1 else np.asfortranarray(host)
        p = self.dev_alloc(host.nbytes)
        self.check(self._lib.gp_dev_upload(self.h, p, host.ctypes.data_as(C.c_void_p), host.nbytes))
        return p

    def download(self, p, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype, order="F")
        self.check(self._lib.gp_dev_download(self.h, out.ctypes.data_as(C.c_void_p), p, out.nbytes))
        return out

    # -- Gram / dense LA on host arrays -------------------------------------------------------
    def gram_rbf(self, X, theta, full=True, out=None):
        X, theta = L.f64(X), L.f64(theta)
        n, d = X.shape
        if theta.size != d + 2:
            raise ValueError("%d does not equal to %d" % (theta.size, d + 2))
        K = np.zeros((n, n), order="F") if out is None else out
        self.check(self._lib.gp_gram_rbf(self.h, L.dptr(X), n, d, max(n, 1), L.dptr(theta), L.dptr(K), max(n, 1),
                                         L.GP_FULL if full else L.GP_LOWER))
        return K

    def dgram_rbf(self, X, theta, pos):
        """dK/dtheta_pos (pos 1-based), full symmetric."""
        X = L.f64(X)
        n, d = X.shape
        theta = np.ascontiguousarray(theta, dtype=np.float64)
        if theta.size != d + 2:
            raise ValueError("requirement failed: hyper-parameter vector must have d + 2 entries")
        D = np.zeros((n, n), order="F")
        self.check(self._lib.gp_dgram_rbf(self.h, L.dptr(X), n, d, max(n, 1), L.dptr(theta), int(pos), L.dptr(D), max(n, 1)))
        return D

    def cross_gram_rbf(self, Xs, X, theta):
        Xs, X, theta = L.f64(Xs), L.f64(X), L.f64(theta)
        m, d = Xs.shape
        n = X.shape[0]
        if X.shape[1] != d or theta.size != d + 2:
            raise ValueError("dimension mismatch")
        Ks = np.zeros((m, n), order="F")
        self.check(self._lib.gp_cross_gram_rbf(self.h, L.dptr(Xs), m, max(m, 1), L.dptr(X), n, max(n, 1), d, L.dptr(theta),
                                               L.dptr(Ks), max(m, 1)))
        return Ks

    def potrf_lower(self, A):
        A = np.array(A, dtype=np.float64, order="F", copy=True)
        n = A.shape[0]
        if A.shape[1] != n:
            raise ValueError("matrix must be square")
        info = C.c_int()
        st = self._lib.gp_potrf_lower(self.h, L.dptr(A), n, max(n, 1), C.byref(info))
        self.check(st, info.value)
        return A

    def trsm_lower(self, Lm, B, trans=False):
        Lm = L.f64(Lm)
        n = Lm.shape[0]
        vec = np.ndim(B) == 1
        Bm = np.array(np.reshape(B, (n, -1), order="F"), dtype=np.float64, order="F", copy=True)
        self.check(self._lib.gp_trsm_lower(self.h, int(bool(trans)), L.dptr(Lm), n, max(n, 1), L.dptr(Bm), Bm.shape[1], max(n, 1)))
        return Bm[:, 0].copy() if vec else Bm

    def inv_lower(self, Lm):
        Lm = L.f64(Lm)
        n = Lm.shape[0]
        out = np.zeros((n, n), order="F")
        self.check(self._lib.gp_inv_lower(self.h, L.dptr(Lm), n, max(n, 1), L.dptr(out), max(n, 1)))
        return out

    # -- Co2Kernel (gp/regression/Co2Prediction.scala:29-137): 1-D inputs, 11 hyper-parameters -------------------------------
    def gram_co2(self, x, theta, xs=None, pos=0):
        """Gram (xs None; full symmetric) / cross-Gram (xs given: len(xs) x n) / derivative Gram (pos = 1..11) of the Co2Kernel."""
        x, theta = L.f64(np.asarray(x, dtype=np.float64).reshape(-1)), L.f64(theta)
        if theta.size != 11:
            raise ValueError("Co2Kernel takes 11 hyper-parameters")
        n = x.size
        if xs is not None:
            xs = L.f64(np.asarray(xs, dtype=np.float64).reshape(-1))
            K = np.zeros((xs.size, n), order="F")
            self.check(self._lib.gp_cross_gram_co2(self.h, L.dptr(xs), xs.size, L.dptr(x), n, L.dptr(theta), L.dptr(K), max(xs.size, 1)))
            return K
        K = np.zeros((n, n), order="F")
        if pos:
            self.check(self._lib.gp_dgram_co2(self.h, L.dptr(x), n, L.dptr(theta), int(pos), L.dptr(K), max(n, 1)))
        else:
            self.check(self._lib.gp_gram_co2(self.h, L.dptr(x), n, L.dptr(theta), L.dptr(K), max(n, 1), L.GP_FULL))
        return K

    def lml_grad_co2_batched(self, x, y, thetas, nparams=None, sigma_noise=None):
        x, y = L.f64(np.asarray(x, dtype=np.float64).reshape(-1)), L.f64(y)
        thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        B, P = thetas.shape
        if P != 11 or y.size != x.size:
            raise ValueError("dimension mismatch")
        nparams = P if nparams is None else int(nparams)
        lml, grad, info = np.zeros(B), np.zeros((B, max(nparams, 1))), np.zeros(B, dtype=np.int32)
        sn = float("nan") if sigma_noise is None else float(sigma_noise)
        self.check(self._lib.gp_lml_grad_co2_batched(self.h, L.dptr(x), x.size, L.dptr(y), L.dptr(thetas), B, nparams, sn, L.dptr(lml), L.dptr(grad),
                                                     info.ctypes.data_as(C.POINTER(C.c_int))))
        return lml, grad[:, :nparams], info

    def lml_grad_from_gram(self, K, y, dKs, sigma_noise=None):
        """gp_lml_grad_from_gram: (lml, grad[len(dKs)]) for ANY kernel from host-built K and derivative matrices dKs (symmetric)."""
        K, y = L.f64(K), L.f64(y)
        n = K.shape[0]
        if K.shape != (n, n) or y.size != n:
            raise ValueError("dimension mismatch")
        dKs = [L.f64(D) for D in dKs]
        if any(D.shape != (n, n) for D in dKs):
            raise ValueError("every derivative matrix must be n x n")
        P = len(dKs)
        ptrs = (C.POINTER(C.c_double) * max(P, 1))(*[L.dptr(D) for D in dKs])
        lml, grad, info = C.c_double(), np.zeros(max(P, 1)), C.c_int()
        sn = float("nan") if sigma_noise is None else float(sigma_noise)
        self.check(self._lib.gp_lml_grad_from_gram(self.h, L.dptr(K), n, n, L.dptr(y), ptrs, P, n, sn, C.byref(lml), L.dptr(grad), C.byref(info)),
                   info.value)
        return lml.value, grad[:P].copy()

    def optimize_co2(self, x, y, theta0, nparams=11, sigma_noise=None, max_iter=20, history=4):
        x, y = L.f64(np.asarray(x, dtype=np.float64).reshape(-1)), L.f64(y)
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        if theta0.size != 11 or y.size != x.size:
            raise ValueError("dimension mismatch")
        out, lml = np.zeros(11), np.zeros(1)
        its, evs = C.c_int(0), C.c_int(0)
        sn = float("nan") if sigma_noise is None else float(sigma_noise)
        self.check(self._lib.gp_optimize_co2(self.h, L.dptr(x), x.size, L.dptr(y), L.dptr(theta0), int(nparams), sn, int(max_iter), int(history),
                                             L.dptr(out), L.dptr(lml), C.byref(its), C.byref(evs)))
        return out, float(lml[0]), its.value, evs.value

    def posterior_from_factor(self, X, theta, Lm, alpha, Xs, full_cov=False, want_v=False):
        """gp_posterior_from_factor (GpPredictor.computePosterior with the ARD-RBF kernel): (mean, var, cov|None, V|None)."""
        X, theta, Lm, alpha, Xs = L.f64(X), L.f64(theta), L.f64(Lm), L.f64(alpha), L.f64(Xs)
        n, d = X.shape
        m = Xs.shape[0]
        if Xs.shape[1] != d or theta.size != d + 2 or Lm.shape != (n, n) or alpha.size != n:
            raise ValueError("dimension mismatch")
        mean, var = np.zeros(m), np.zeros(m)
        cov = np.zeros((m, m), order="F") if full_cov else None
        V = np.zeros((n, m), order="F") if want_v else None
        self.check(self._lib.gp_posterior_from_factor(self.h, L.dptr(X), n, d, max(n, 1), L.dptr(theta), L.dptr(Lm), max(n, 1), L.dptr(alpha),
                                                      L.dptr(Xs), m, max(m, 1), L.dptr(mean), L.dptr(var),
                                                      L.dptr(cov) if full_cov else None, max(m, 1), L.dptr(V) if want_v else None, max(n, 1)))
        return mean, var, cov, V

    def posterior_from_gram(self, Ks, Lm, alpha, Kss=None, kss_diag=None, want_v=False):
        """gp_posterior_from_gram (computePosterior with a host-evaluated KernelFunc): (mean, var|None, cov|None, V|None)."""
        Ks, Lm, alpha = L.f64(Ks), L.f64(Lm), L.f64(alpha)
        m, n = Ks.shape
        if Lm.shape != (n, n) or alpha.size != n:
            raise ValueError("dimension mismatch")
        Kss = None if Kss is None else L.f64(Kss)
        kd = None if kss_diag is None else L.f64(kss_diag)
        if (Kss is not None and Kss.shape != (m, m)) or (kd is not None and kd.size != m):
            raise ValueError("dimension mismatch")
        mean = np.zeros(m)
        var = np.zeros(m) if (Kss is not None or kd is not None) else None
        cov = np.zeros((m, m), order="F") if Kss is not None else None
        V = np.zeros((n, m), order="F") if want_v else None
        self.check(self._lib.gp_posterior_from_gram(self.h, L.dptr(Ks), m, n, max(m, 1), L.dptr(Kss) if Kss is not None else None, max(m, 1),
                                                    L.dptr(kd) if kd is not None else None, L.dptr(Lm), max(n, 1), L.dptr(alpha), L.dptr(mean),
                                                    L.dptr(var) if var is not None else None, L.dptr(cov) if cov is not None else None,
                                                    max(m, 1), L.dptr(V) if want_v else None, max(n, 1)))
        return mean, var, cov, V

    def lml_grad_batched(self, X, y, thetas, nparams=None, sigma_noise=None):
        X, y = L.f64(X), L.f64(y)
        n, d = X.shape
        thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        B, P = thetas.shape
        if P != d + 2 or y.size != n:
            raise ValueError("dimension mismatch")
        nparams = P if nparams is None else int(nparams)
        lml = np.zeros(B)
        grad = np.zeros((B, max(nparams, 1)))
        info = np.zeros(B, dtype=np.int32)
        sn = float("nan") if sigma_noise is None else float(sigma_noise)
        self.check(self._lib.gp_lml_grad_rbf_batched(self.h, L.dptr(X), n, d, n, L.dptr(y), L.dptr(thetas), B, nparams, sn,
                                                     L.dptr(lml), L.dptr(grad), info.ctypes.data_as(C.POINTER(C.c_int))))
        return lml, grad[:, :nparams], info

    def ep_lml_rbf_batched(self, X, y, thetas, stop_eps=0.01, max_sweeps=1000, strict=True):
        """gp_ep_lml_rbf_batched: (lml[B], sweeps[B], info[B])."""
        X = L.f64(X)
        n, d = X.shape
        yi = np.ascontiguousarray(y, dtype=np.int32).reshape(-1)
        thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        B, P = thetas.shape
        if P != d + 2 or yi.size != n:
            raise ValueError("dimension mismatch")
        lml, sw, info = np.zeros(B), np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self.check(self._lib.gp_ep_lml_rbf_batched(self.h, L.dptr(X), n, d, n, yi.ctypes.data_as(C.POINTER(C.c_int32)), L.dptr(thetas), B,
                                                   float(stop_eps), int(max_sweeps), int(bool(strict)), L.dptr(lml),
                                                   sw.ctypes.data_as(C.POINTER(C.c_int)), info.ctypes.data_as(C.POINTER(C.c_int))))
        return lml, sw, info

    def ep_lml_grad_rbf_batched(self, X, y, thetas, stop_eps=0.01, max_sweeps=1000, strict=True):
        """gp_ep_lml_grad_rbf_batched: (lml[B], grad[B, d+2], sweeps[B], info[B])."""
        X = L.f64(X)
        n, d = X.shape
        yi = np.ascontiguousarray(y, dtype=np.int32).reshape(-1)
        thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        B, P = thetas.shape
        if P != d + 2 or yi.size != n:
            raise ValueError("dimension mismatch")
        lml, grad = np.zeros(B), np.zeros((B, P))
        sw, info = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self.check(self._lib.gp_ep_lml_grad_rbf_batched(self.h, L.dptr(X), n, d, n, yi.ctypes.data_as(C.POINTER(C.c_int32)), L.dptr(thetas), B,
                                                        float(stop_eps), int(max_sweeps), int(bool(strict)), L.dptr(lml), L.dptr(grad),
                                                        sw.ctypes.data_as(C.POINTER(C.c_int)), info.ctypes.data_as(C.POINTER(C.c_int))))
        return lml, grad, sw, info

    def ep_optimize_rbf(self, X, y, theta0, stop_eps=0.01, max_sweeps=1000, strict=True, max_iter=10, history=4):
        """gp_ep_optimize_rbf: (theta*, EP LML(theta*), iterations, evaluations)."""
        X = L.f64(X)
        n, d = X.shape
        yi = np.ascontiguousarray(y, dtype=np.int32).reshape(-1)
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        if theta0.size != d + 2 or yi.size != n:
            raise ValueError("dimension mismatch")
        out, lml = np.zeros(d + 2), np.zeros(1)
        its, evs = C.c_int(0), C.c_int(0)
        self.check(self._lib.gp_ep_optimize_rbf(self.h, L.dptr(X), n, d, n, yi.ctypes.data_as(C.POINTER(C.c_int32)), L.dptr(theta0), float(stop_eps),
                                                int(max_sweeps), int(bool(strict)), int(max_iter), int(history), L.dptr(out), L.dptr(lml),
                                                C.byref(its), C.byref(evs)))
        return out, float(lml[0]), its.value, evs.value

    def optimize_rbf(self, X, y, theta0, nparams=None, sigma_noise=None, max_iter=20, history=4):
        """gp_optimize_rbf: (theta*, LML(theta*), iterations, evaluations)."""
        X, y = L.f64(X), L.f64(y)
        n, d = X.shape
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        if theta0.size != d + 2 or y.size != n:
            raise ValueError("dimension mismatch")
        nparams = d + 2 if nparams is None else int(nparams)
        out, lml = np.zeros(d + 2), np.zeros(1)
        its, evs = C.c_int(0), C.c_int(0)
        sn = float("nan") if sigma_noise is None else float(sigma_noise)
        self.check(self._lib.gp_optimize_rbf(self.h, L.dptr(X), n, d, n, L.dptr(y), L.dptr(theta0), nparams, sn, int(max_iter), int(history),
                                             L.dptr(out), L.dptr(lml), C.byref(its), C.byref(evs)))
        return out, float(lml[0]), its.value, evs.value


class RegressionModel:
    """(L, alpha, LML) of GpPredictor.preComputeComponents, resident on the GPU."""

    def __init__(self, ctx, X=None, y=None, theta=None, sigma_noise=None, gram=None, kernel="rbf"):
        self.ctx = ctx
        lib = ctx._lib
        h = C.c_void_p()
        info = C.c_int()
        sn = float("nan") if sigma_noise is None else float(sigma_noise)
        y = L.f64(y)
        if gram is not None:
            K = L.f64(gram)
            self.n, self.d = K.shape[0], 0
            if y.size != self.n:
                raise ValueError("Number of objects in training data matrix should be equal to targets vector length")
            st = lib.gp_fit_from_gram(ctx.h, L.dptr(K), self.n, self.n, L.dptr(y), C.byref(h), C.byref(info))
        elif kernel == "co2":   # gp_fit_co2: Co2Kernel on 1-D inputs, the kernel is rebuilt on the device for predictions
            x, theta = L.f64(np.asarray(X, dtype=np.float64).reshape(-1)), L.f64(theta)
            self.n, self.d = x.size, 1
            if y.size != self.n:
                raise ValueError("Number of objects in training data matrix should be equal to targets vector length")
            if theta.size != 11:
                raise ValueError("Co2Kernel takes 11 hyper-parameters")
            st = lib.gp_fit_co2(ctx.h, L.dptr(x), self.n, L.dptr(y), L.dptr(theta), sn, C.byref(h), C.byref(info))
        else:
            X, theta = L.f64(X), L.f64(theta)
            self.n, self.d = X.shape
            if y.size != self.n:
                raise ValueError("Number of objects in training data matrix should be equal to targets vector length")
            if theta.size != self.d + 2:
                raise ValueError("%d does not equal to %d" % (theta.size, self.d + 2))
            st = lib.gp_fit_rbf(ctx.h, L.dptr(X), self.n, self.d, self.n, L.dptr(y), L.dptr(theta), sn, C.byref(h), C.byref(info))
        ctx.check(st, info.value)
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx._lib.gp_model_destroy(self.h)
        self.h = None

    __del__ = close

    def L(self):
        out = np.zeros((self.n, self.n), order="F")
        self.ctx.check(self.ctx._lib.gp_model_get(self.h, L.GP_GET_L, L.dptr(out), self.n))
        return out

    def alpha(self):
        out = np.zeros(self.n)
        self.ctx.check(self.ctx._lib.gp_model_get(self.h, L.GP_GET_ALPHA, L.dptr(out), self.n))
        return out

    def lml(self):
        out = np.zeros(1)
        self.ctx.check(self.ctx._lib.gp_model_get(self.h, L.GP_GET_LML, L.dptr(out), 1))
        return float(out[0])

    def predict(self, Xs, full_cov=False):
        Xs = L.f64(Xs)
        m = Xs.shape[0]
        if Xs.shape[1] != self.d:
            raise ValueError("test data dimension mismatch")
        mean, var = np.zeros(m), np.zeros(m)
        cov = np.zeros((m, m), order="F") if full_cov else None
        self.ctx.check(self.ctx._lib.gp_predict(self.h, L.dptr(Xs), m, max(m, 1), L.dptr(mean), L.dptr(var),
                                                L.dptr(cov) if full_cov else None, max(m, 1)))
        return mean, var, cov


    def predict_from_gram(self, Ks, Kss=None, kss_diag=None):
        """gp_predict_from_gram: posterior for a model fitted from a host-built Gram matrix; (mean, var|None, cov|None)."""
        Ks = L.f64(Ks)
        m = Ks.shape[0]
        if Ks.shape[1] != self.n:
            raise ValueError("test-train matrix must be m x n")
        Kss = None if Kss is None else L.f64(Kss)
        kd = None if kss_diag is None else L.f64(kss_diag)
        if (Kss is not None and Kss.shape != (m, m)) or (kd is not None and kd.size != m):
            raise ValueError("dimension mismatch")
        mean = np.zeros(m)
        var = np.zeros(m) if (Kss is not None or kd is not None) else None
        cov = np.zeros((m, m), order="F") if Kss is not None else None
        self.ctx.check(self.ctx._lib.gp_predict_from_gram(self.h, L.dptr(Ks), m, max(m, 1), L.dptr(Kss) if Kss is not None else None, max(m, 1),
                                                          L.dptr(kd) if kd is not None else None, L.dptr(mean),
                                                          L.dptr(var) if var is not None else None, L.dptr(cov) if cov is not None else None,
                                                          max(m, 1)))
        return mean, var, cov


class SmallModelBatch:
    """gp_small_*: G small GP models over ONE set of training inputs, resident with L^-1 (GP-UCB / GP-UKF workloads)."""

    def __init__(self, ctx, X, thetas, Y=None, Ls=None, alphas=None, sigma_noise=None, capacity=0):
        self.ctx = ctx
        X = L.f64(X)
        self.n0, self.d = X.shape
        thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        self.G = thetas.shape[0]
        if thetas.shape[1] != self.d + 2:
            raise ValueError("%d does not equal to %d" % (thetas.shape[1], self.d + 2))
        h, info = C.c_void_p(), C.c_int()
        n = self.n0
        if Y is not None:
            Y = L.f64(np.asarray(Y, dtype=np.float64).reshape(n, -1))
            if Y.shape[1] != self.G:
                raise ValueError("one target column per model")
            sn = float("nan") if sigma_noise is None else float(sigma_noise)
            st = ctx._lib.gp_small_fit(ctx.h, L.dptr(X), n, self.d, n, L.dptr(Y), n, self.G, L.dptr(thetas), sn, int(capacity), C.byref(h), C.byref(info))
        else:
            Ls = np.asfortranarray(np.concatenate([L.f64(a) for a in Ls], axis=1))       # n x (G n): model g at columns [g n, (g+1) n)
            alphas = np.ascontiguousarray(np.stack([np.asarray(a, dtype=np.float64) for a in alphas]))
            if Ls.shape != (n, self.G * n) or alphas.shape != (self.G, n):
                raise ValueError("dimension mismatch")
            st = ctx._lib.gp_small_from_factors(ctx.h, L.dptr(X), n, self.d, n, L.dptr(thetas), self.G, L.dptr(Ls), n, L.dptr(alphas), int(capacity), C.byref(h))
        ctx.check(st, info.value)
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx._lib.gp_small_destroy(self.h)
        self.h = None

    __del__ = close

    @property
    def n(self):
        n = C.c_int()
        self.ctx.check(self.ctx._lib.gp_small_size(self.h, C.byref(n), None, None))
        return n.value

    def get(self, g, what):
        n = self.n
        out = np.zeros(n) if what == L.GP_SMALL_GET_ALPHA else np.zeros((n, n), order="F")
        self.ctx.check(self.ctx._lib.gp_small_get(self.h, int(g), what, L.dptr(out), n))
        return out

    def posterior(self, Xs):
        """(mean[G, m], var[G, m]) of every model at every test point, one launch."""
        Xs = L.f64(np.asarray(Xs, dtype=np.float64).reshape(-1, self.d))
        m = Xs.shape[0]
        mean, var = np.zeros((self.G, m)), np.zeros((self.G, m))
        self.ctx.check(self.ctx._lib.gp_small_posterior(self.h, L.dptr(Xs), m, max(m, 1), L.dptr(mean), L.dptr(var)))
        return mean, var

    def ucb(self, Xs, kappa, g=0):
        """(value[m], grad[m, d]) of mean + kappa sqrt(var) for model g."""
        Xs = L.f64(np.asarray(Xs, dtype=np.float64).reshape(-1, self.d))
        m = Xs.shape[0]
        val, grad = np.zeros(m), np.zeros((m, self.d))
        self.ctx.check(self.ctx._lib.gp_small_ucb(self.h, int(g), L.dptr(Xs), m, max(m, 1), float(kappa), L.dptr(val), L.dptr(grad)))
        return val, grad

    def append(self, x_new, y_new):
        x_new = np.ascontiguousarray(x_new, dtype=np.float64).reshape(-1)
        y_new = np.ascontiguousarray(np.atleast_1d(y_new), dtype=np.float64)
        if x_new.size != self.d or y_new.size != self.G:
            raise ValueError("dimension mismatch")
        info = C.c_int()
        self.ctx.check(self.ctx._lib.gp_small_append(self.h, L.dptr(x_new), L.dptr(y_new), C.byref(info)), info.value)

    def maximize_ucb(self, starts, kappa, g=0, max_iter=10, history=4):
        """(best_x[d], best_value, evaluations) over c lockstep L-BFGS runs."""
        starts = L.f64(np.asarray(starts, dtype=np.float64).reshape(-1, self.d))
        c = starts.shape[0]
        bx, bv, ev = np.zeros(self.d), C.c_double(), C.c_int()
        self.ctx.check(self.ctx._lib.gp_small_maximize_ucb(self.h, int(g), L.dptr(starts), c, c, float(kappa), int(max_iter), int(history),
                                                           L.dptr(bx), C.byref(bv), C.byref(ev)))
        return bx, bv.value, ev.value


class DistGroup:
    """gp_dist_*: this rank's RCCL communicator on `ctx`'s device (one process per GPU).  `exchange(id_bytes_or_None) -> id_bytes`
    ships rank 0's 128-byte id to every rank (torch.distributed broadcast, a file, a socket ...)."""

    def __init__(self, ctx, rank, world, exchange=None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        lib = ctx._lib
        buf = C.create_string_buffer(L.GP_DIST_ID_BYTES)
        if self.rank == 0:
            ctx.check(lib.gp_dist_unique_id(ctx.h, buf))
        ident = bytes(buf.raw)
        if exchange is not None:
            ident = exchange(ident if self.rank == 0 else None)
        if len(ident) != L.GP_DIST_ID_BYTES:
            raise ValueError("the RCCL id must be %d bytes" % L.GP_DIST_ID_BYTES)
        h = C.c_void_p()
        ctx.check(lib.gp_dist_init(ctx.h, C.create_string_buffer(ident, L.GP_DIST_ID_BYTES), self.rank, self.world, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx._lib.gp_dist_destroy(self.h)
        self.h = None

    __del__ = close

    def inject_failure(self):
        """Test hook (gp_dist_inject_failure): this rank's next collective call fails locally with GP_ENOMEM."""
        self.ctx.check(self.ctx._lib.gp_dist_inject_failure(self.h))

    def shard(self, total):
        lo, hi = C.c_int(), C.c_int()
        self.ctx.check(self.ctx._lib.gp_dist_shard(self.h, int(total), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def lml_grad_batched(self, X, y, thetas, nparams=None, sigma_noise=None):
        X, y = L.f64(X), L.f64(y)
        n, d = X.shape
        thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
        B, P = thetas.shape
        if P != d + 2 or y.size != n:
            raise ValueError("dimension mismatch")
        nparams = P if nparams is None else int(nparams)
        lml, grad, info = np.zeros(B), np.zeros((B, max(nparams, 1))), np.zeros(B, dtype=np.int32)
        sn = float("nan") if sigma_noise is None else float(sigma_noise)
        self.ctx.check(self.ctx._lib.gp_dist_lml_grad_batched(self.h, L.dptr(X), n, d, n, L.dptr(y), L.dptr(thetas), B, nparams, sn,
                                                              L.dptr(lml), L.dptr(grad), info.ctypes.data_as(C.POINTER(C.c_int))))
        return lml, grad[:, :nparams], info

    def predict(self, model, Xs):
        Xs = L.f64(Xs)
        m = Xs.shape[0]
        if Xs.shape[1] != model.d:
            raise ValueError("test data dimension mismatch")
        mean, var = np.zeros(m), np.zeros(m)
        self.ctx.check(self.ctx._lib.gp_dist_predict(self.h, model.h, L.dptr(Xs), m, max(m, 1), L.dptr(mean), L.dptr(var)))
        return mean, var


class EpClassifierState:
    """EpParameterEstimator state on the GPU: K, Sigma, L and the site parameters."""

    def __init__(self, ctx, K, y):
        self.ctx = ctx
        K = L.f64(K)
        y = np.ascontiguousarray(y, dtype=np.int32)
        self.n = K.shape[0]
        if K.shape[1] != self.n or y.size != self.n:
            raise ValueError("kernelMatrix.rows must equal targets.length")
        h = C.c_void_p()
        st = ctx._lib.gp_ep_create(ctx.h, L.dptr(K), self.n, self.n, y.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(h))
        ctx.check(st)
        self.h = h

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            self.ctx._lib.gp_ep_destroy(self.h)
        self.h = None

    __del__ = close

    def sweep(self, nsweeps=1):
        tau, nu = np.zeros(self.n), np.zeros(self.n)
        info = C.c_int()
        st = self.ctx._lib.gp_ep_sweep(self.h, int(nsweeps), L.dptr(tau), L.dptr(nu), C.byref(info))
        self.ctx.check(st, info.value)
        return tau, nu

    def load_site_params(self, tau, nu):
        tau, nu = L.f64(tau), L.f64(nu)
        if tau.size != self.n or nu.size != self.n:
            raise ValueError("site parameter length mismatch")
        info = C.c_int()
        st = self.ctx._lib.gp_ep_set_site_params(self.h, L.dptr(tau), L.dptr(nu), C.byref(info))
        self.ctx.check(st, info.value)

    def lml(self, strict=True):
        v = C.c_double()
        self.ctx.check(self.ctx._lib.gp_ep_lml(self.h, int(bool(strict)), C.byref(v)))
        return v.value

    def lml_grad_rbf(self, X, theta, strict=True):
        X, theta = L.f64(X), L.f64(theta)
        if X.shape[0] != self.n or theta.size != X.shape[1] + 2:
            raise ValueError("dimension mismatch")
        g = np.zeros(theta.size)
        self.ctx.check(self.ctx._lib.gp_ep_lml_grad_rbf(self.h, L.dptr(X), X.shape[1], self.n, L.dptr(theta), int(bool(strict)), L.dptr(g)))
        return g

    def get(self, what):
        mat = what in (L.GP_EP_GET_L, L.GP_EP_GET_SIGMA)
        out = np.zeros((self.n, self.n), order="F") if mat else np.zeros(self.n)
        self.ctx.check(self.ctx._lib.gp_ep_get(self.h, what, L.dptr(out), self.n))
        return out

    def predict(self, Ks, kss_diag):
        Ks, kd = L.f64(Ks), L.f64(kss_diag)
        m = Ks.shape[0]
        if Ks.shape[1] != self.n or kd.size != m:
            raise ValueError("dimension mismatch")
        prob = np.zeros(m)
        self.ctx.check(self.ctx._lib.gp_ep_predict(self.h, L.dptr(Ks), m, max(m, 1), L.dptr(kd), L.dptr(prob)))
        return prob
