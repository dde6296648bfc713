"""Soak of the single-launch factorisation: many refits of one model, alpha and the LML of every refit compared bit for bit with the
launch-per-step form's (GPCORE_CHOL_MEGA=0 for the reference refit, then unset).  A lost hand-over between two workgroups would show as
a different factor -- or as the library's bounded wait reporting GP_EHIP -- long before it showed in a test that factors a dozen times.
usage: python tools/mega_soak.py [n] [refits] [contend]   (prints a line every 500 refits)
contend = 1: a second thread keeps a second context busy with posterior batches of another model (full-chip GEMM launches of ~1 ms)
the whole time -- the single launch's 256 workgroups then are NOT all resident together, which it must not need."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gp_algos_amd import _lib as L, synth
from gp_algos_amd.core import Context, RegressionModel
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
p = synth.config_c2(n, 8, 0)
ctx = Context(0)
os.environ["GPCORE_CHOL_MEGA"] = "0"
m = RegressionModel(ctx, p["X"], p["y"], p["theta"])
ref_alpha, ref_lml = m.alpha(), m.lml()
del os.environ["GPCORE_CHOL_MEGA"]
theta = L.f64(p["theta"])
stop, other_batches = False, 0
if len(sys.argv) > 3 and sys.argv[3] == "1":
    import threading
    octx = Context(0)
    q = synth.config_c2(4096, 8, 131072)
    om = RegressionModel(octx, q["X"], q["y"], q["theta"])
    want = om.predict(q["Xs"])

    def contend():
        global other_batches
        while not stop:
            got = om.predict(q["Xs"])
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            other_batches += 1
    th = threading.Thread(target=contend)
    th.start()
bad, t0 = 0, time.perf_counter()
for r in range(reps):
    ctx.check(ctx._lib.gp_model_refit_dev(m.h, L.dptr(theta), float("nan")))
    if r < 3 or r % 7 == 0:                          # most refits are only queued behind each other; every seventh is read back
        if not (np.array_equal(m.alpha(), ref_alpha) and m.lml() == ref_lml):
            bad += 1
            print("refit %d differs" % r, flush=True)
    if (r + 1) % 500 == 0:
        ctx.sync()
        print("n=%d: %d refits, %d differing, %.1f s" % (n, r + 1, bad, time.perf_counter() - t0), flush=True)
ctx.sync()
assert np.array_equal(m.alpha(), ref_alpha) and m.lml() == ref_lml
if len(sys.argv) > 3 and sys.argv[3] == "1":
    stop = True
    th.join()
    om.close()
    octx.close()
print("n=%d: %d refits done, %d differing%s" % (n, reps, bad, "; %d posterior batches of 131072 points (n = 4096) on a second context meanwhile" % other_batches if other_batches else ""))
m.close()
ctx.close()
sys.exit(1 if bad else 0)
