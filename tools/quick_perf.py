"""Ad-hoc timing of the C2 hot path (fit n=8192 d=8, predict m points) through the C-ABI, no torch."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_algos_amd import _lib as L, synth
from gp_algos_amd.core import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
d = 8
p = synth.config_c2(n, d, m)
ctx = Context(0)
lib = ctx._lib
print("mfma f64 probe TFLOP/s:", ctx.probe_mfma_f64(), flush=True)
dX, dy, dXs = ctx.upload(p["X"]), ctx.upload(p["y"]), ctx.upload(p["Xs"])
dmean, dvar = ctx.dev_alloc(8 * m), ctx.dev_alloc(8 * m)
theta = L.f64(p["theta"])
h = C.c_void_p()
info = C.c_int()
t0 = time.perf_counter()
ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, d, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)))
print("first fit (incl. alloc) %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)


def timed(fn, reps=3):
    ctx.sync()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ctx.sync()
        best = min(best, time.perf_counter() - t)
    return best


tf = timed(lambda: ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), float("nan"))))
print("refit n=%d: %.3f ms  -> cholesky-equivalent %.2f TFLOP/s (n^3/3 over whole fit)" % (n, tf * 1e3, n ** 3 / 3 / tf / 1e12), flush=True)
ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))
tp = timed(lambda: ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar)), reps=2)
print("predict m=%d: %.3f ms -> %.1f points/s, %.2f TFLOP/s (n^2 m)" % (m, tp * 1e3, m / tp, n * n * m / tp / 1e12), flush=True)
names = {1: "gemm", 2: "syrk", 3: "gram", 4: "trsm_panel", 5: "potrf_diag", 6: "panel_upd"}
for cls in (5, 4, 6, 2, 3):
    ctx.profile(1 << cls)
    ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), float("nan")))
    k, ms, work = ctx.profile_read(cls)
    ctx.profile(0)
    print("fit   %-10s launches=%4d total=%8.3f ms avg=%8.2f us  work/s=%.3e" % (names[cls], k, ms, ms / max(k, 1) * 1e3, work / (ms * 1e-3 + 1e-12)), flush=True)
for cls in (1, 4, 3):
    ctx.profile(1 << cls)
    ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))
    k, ms, work = ctx.profile_read(cls)
    ctx.profile(0)
    print("pred  %-10s launches=%4d total=%8.3f ms avg=%8.2f us  work/s=%.3e" % (names[cls], k, ms, ms / max(k, 1) * 1e3, work / (ms * 1e-3 + 1e-12)), flush=True)
mean = ctx.download(dmean, (m,))
var = ctx.download(dvar, (m,))
print("mean[:3]", mean[:3], "var[:3]", var[:3], "finite", np.isfinite(mean).all(), np.isfinite(var).all(), "var range", var.min(), var.max())
lib.gp_model_destroy(h)
