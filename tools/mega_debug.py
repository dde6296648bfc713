"""Lab: one factorisation through the single-launch form at a small size, with GPCORE_MEGA_MODE selecting which task bodies run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gp_algos_amd import synth
from gp_algos_amd.core import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
p = synth.regression(n, 3, 0, 5, 6, 0, synth.ard_theta(3, 1.3, 0.9, 0.3))
ctx = Context(0)
K = ctx.gram_rbf(p["X"], p["theta"])
ctx.check(ctx._lib.gp_ctx_set_lookahead(ctx.h, 1))
t0 = time.perf_counter()
try:
    L = ctx.potrf_lower(K.copy(order="F"))
    print("mode", os.environ.get("GPCORE_MEGA_MODE", "7"), "n", n, "returned in %.3f s" % (time.perf_counter() - t0), "finite", bool(np.all(np.isfinite(L))),
          "resid %.2e" % (np.linalg.norm(L @ L.T - K) / np.linalg.norm(K)), flush=True)
except Exception as e:
    print("mode", os.environ.get("GPCORE_MEGA_MODE", "7"), "n", n, "raised after %.3f s:" % (time.perf_counter() - t0), repr(e)[:300], flush=True)
ctx.close()
