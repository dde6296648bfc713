"""Does the posterior GEMM slow down under sustained load?  n = 32768 fit, then `reps` consecutive 131072-point predicts, each timed
(2.06 s per batch in a 2-batch run against 2.26 s averaged over a 10^6-point request).
usage: python tools/c5_sustained.py [reps] [m]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_algos_amd import _lib as L, synth
from gp_algos_amd.core import Context
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
m = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
n, d = 32768, 8
p = synth.config_c5(n, d, m)
ctx = Context(0); lib = ctx._lib
dX, dy, dXs = ctx.upload(p["X"]), ctx.upload(p["y"]), ctx.upload(p["Xs"])
dmean, dvar = ctx.dev_alloc(8 * m), ctx.dev_alloc(8 * m)
theta = L.f64(p["theta"]); h, info = C.c_void_p(), C.c_int()
ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, d, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)), info.value)
ctx.sync()
ts = []
for r in range(reps):
    t0 = time.perf_counter()
    ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar)); ctx.sync()
    ts.append(time.perf_counter() - t0)
    print("predict %2d: %.1f ms -> %.0f points/s, %.2f TFLOP/s (n^2 m)" % (r, ts[-1] * 1e3, m / ts[-1], n * n * m / ts[-1] / 1e12), flush=True)
print("first (incl. Lw) %.1f ms, second %.1f ms, mean of the last %d: %.1f ms (+%.1f %% on the second)" %
      (ts[0] * 1e3, ts[1] * 1e3, max(1, reps - 3), np.mean(ts[3:]) * 1e3 if reps > 3 else ts[-1] * 1e3,
       100 * ((np.mean(ts[3:]) if reps > 3 else ts[-1]) / ts[1] - 1)))
lib.gp_model_destroy(h)
