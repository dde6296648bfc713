"""Run a few refits at n (for rocprofv3 kernel traces)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_algos_amd import _lib as L, synth
from gp_algos_amd.core import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = synth.config_c2(n, 8, 0)
ctx = Context(0)
lib = ctx._lib
dX, dy = ctx.upload(p["X"]), ctx.upload(p["y"])
theta = L.f64(p["theta"])
h, info = C.c_void_p(), C.c_int()
ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, 8, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)))
import time
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), float("nan")))
ctx.sync()
dt = (time.perf_counter() - t0) / reps
print("n=%d refit %.3f ms -> %.2f TFLOP/s (n^3/3)  GPCORE_OUTER=%s" % (n, dt * 1e3, n ** 3 / 3.0 / dt / 1e12, os.environ.get("GPCORE_OUTER", "default")))
lib.gp_model_destroy(h)
