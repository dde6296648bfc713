"""Average launch time of the factorisation's chain kernels (diagonal block, panel solve, in-panel update) during refits at n,
with HIP events around each launch (gp_ctx_profile): python tools/chain_kernels.py [n] [reps]."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gp_algos_amd import _lib as L, synth  # noqa: E402
from gp_algos_amd.core import Context  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = synth.config_c2(n, 8, 0)
ctx = Context(0)
lib = ctx._lib
dX, dy = ctx.upload(p["X"]), ctx.upload(p["y"])
theta = L.f64(p["theta"])
h, info = C.c_void_p(), C.c_int()
ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, 8, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)))
lib.gp_ctx_set_lookahead(ctx.h, 0)     # one stream: the per-launch times are those of the kernels alone
for name, which in (("potrf_diag128", L.GP_PROF_POTRF_DIAG), ("trsm_panel128", L.GP_PROF_TRSM), ("in-panel update", L.GP_PROF_PANEL_UPD),
                    ("outer update", L.GP_PROF_SYRK)):
    ctx.profile(1 << which)
    for _ in range(reps):
        ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), float("nan")))
    ctx.sync()
    cnt, ms, work = ctx.profile_read(which)
    ctx.profile(0)
    print("n=%d %-16s %5d launches  avg %7.2f us  total/refit %7.3f ms" % (n, name, cnt, ms * 1e3 / max(cnt, 1), ms / reps))
lib.gp_model_destroy(h)
