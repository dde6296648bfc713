"""One fit + a few predicts at C2 scale (for rocprofv3 traces / PMC passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_algos_amd import _lib as L, synth
from gp_algos_amd.core import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = synth.config_c2(n, 8, m)
ctx = Context(0); lib = ctx._lib
dX, dy, dXs = ctx.upload(p["X"]), ctx.upload(p["y"]), ctx.upload(p["Xs"])
dmean, dvar = ctx.dev_alloc(8 * m), ctx.dev_alloc(8 * m)
theta = L.f64(p["theta"]); h, info = C.c_void_p(), C.c_int()
ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, 8, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)))
import time
ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))   # warm-up (workspaces, Lw)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print("n=%d m=%d predict %.2f ms -> %.0f points/s  (GPCORE_PREDICT_BATCH=%s)" % (n, m, dt * 1e3, m / dt, os.environ.get("GPCORE_PREDICT_BATCH", "65536")))
lib.gp_model_destroy(h)
