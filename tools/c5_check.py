"""Config C5 at scale: n=32768 fit + m test points in batches; identity checks on sampled rows."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_algos_amd import _lib as L, synth
from gp_algos_amd.core import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
m = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
d = 8
p = synth.config_c5(n, d, m)
ctx = Context(0); lib = ctx._lib
dX, dy, dXs = ctx.upload(p["X"]), ctx.upload(p["y"]), ctx.upload(p["Xs"])
dmean, dvar = ctx.dev_alloc(8 * m), ctx.dev_alloc(8 * m)
theta = L.f64(p["theta"]); h, info = C.c_void_p(), C.c_int()
t0 = time.perf_counter()
ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, d, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)), info.value)
print("first fit %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
ctx.sync(); t0 = time.perf_counter()
ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), float("nan"))); ctx.sync()
tf = time.perf_counter() - t0
print("refit n=%d: %.1f ms -> %.2f TFLOP/s (n^3/3)" % (n, tf * 1e3, n ** 3 / 3 / tf / 1e12), flush=True)
ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar)); ctx.sync()
t0 = time.perf_counter()
ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar)); ctx.sync()
tp = time.perf_counter() - t0
print("predict m=%d: %.1f ms -> %.0f points/s, %.2f TFLOP/s (n^2 m)" % (m, tp * 1e3, m / tp, n * n * m / tp / 1e12), flush=True)
alpha = np.zeros(n); ctx.check(lib.gp_model_get(h, L.GP_GET_ALPHA, L.dptr(alpha), n))
lml = np.zeros(1); ctx.check(lib.gp_model_get(h, L.GP_GET_LML, L.dptr(lml), 1))
mean, var = ctx.download(dmean, (m,)), ctx.download(dvar, (m,))
# identities on sampled rows: (K alpha)_i = y_i ; posterior mean at test point j = k*_j . alpha
rng = np.random.default_rng(0); idx = rng.choice(n, 64, replace=False)
Z = p["X"] / p["theta"][1:-1]; sf2, sn2 = p["theta"][0] ** 2, p["theta"][-1] ** 2
def krow(z): return sf2 * np.exp(-0.5 * ((Z - z) ** 2).sum(axis=1))
errK = max(abs(krow(Z[i]) @ alpha + sn2 * alpha[i] - p["y"][i]) for i in idx)
Zs = p["Xs"] / p["theta"][1:-1]; jdx = rng.choice(m, 64, replace=False)
errM = max(abs(krow(Zs[j]) @ alpha - mean[j]) for j in jdx)
print("LML %.6f  max|K alpha - y| (64 rows) %.2e  max|k*.alpha - mean| (64 pts) %.2e  var range [%.3e, %.3e] finite %s" %
      (lml[0], errK, errM, var.min(), var.max(), bool(np.isfinite(mean).all() and np.isfinite(var).all())), flush=True)
assert errK < 1e-8 and errM < 1e-9 and var.min() > 0 and var.max() <= sf2 + sn2 + 1e-9
lib.gp_model_destroy(h)
print("C5 check ok")
