"""Lab: where the single-launch factorisation's time goes.  Runs one refit at n with GPCORE_MEGA_TRACE set and reads the per-task stamps
(claimed / dependencies met / body done / published; s_memrealtime, 100 MHz): per task type the number of tasks, mean wait for
dependencies, mean body time, mean publish time; the makespan; the busy fraction of the workgroups; the chain (diagonal blocks) alone.
usage: python tools/mega_trace.py [n]"""
import ctypes as C, os, struct, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
path = os.path.join(tempfile.gettempdir(), "mega_trace_%d.bin" % os.getpid())
from gp_algos_amd import _lib as L, synth
from gp_algos_amd.core import Context
p = synth.config_c2(n, 8, 0)
ctx = Context(0)
lib = ctx._lib
dX, dy = ctx.upload(p["X"]), ctx.upload(p["y"])
theta = L.f64(p["theta"])
h, info = C.c_void_p(), C.c_int()
ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, 8, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)))
for _ in range(2):
    ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), float("nan")))
ctx.sync()
os.environ["GPCORE_MEGA_TRACE"] = path
ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), float("nan")))
ctx.sync()
del os.environ["GPCORE_MEGA_TRACE"]
raw = open(path, "rb").read()
os.remove(path)
nt, np_, extra, ob = struct.unpack("4i", raw[:16])
tasks = np.frombuffer(raw[16:16 + 64 * nt], dtype=np.int32).reshape(nt, 16)
st = np.frombuffer(raw[16 + 64 * nt:], dtype=np.uint64).reshape(nt, 4)
wg = (st[:, 3] >> np.uint64(48)).astype(np.int64)
t = (st & np.uint64((1 << 48) - 1)).astype(np.float64) / 100.0          # us
typ, kb, q = tasks[:, 0], tasks[:, 2], tasks[:, 5]
t0 = t[typ != 3, 0].min()
t -= t0
t[~(typ != 3)] = 0.0
print("n=%d np=%d extra=%d: %d tasks, makespan %.1f us (first claim -> last publish)" % (n, np_, extra, nt, t[:, 3].max()))
live = typ != 3
classes = (("POTRF", typ == 0), ("link + POTRF", typ == 4), ("TRSM", typ == 1), ("UPD K=128 full", (typ == 2) & (kb == 1) & (q < 0)), ("UPD K=128 quarter", (typ == 2) & (kb == 1) & (q >= 0)),
           ("UPD K=512 full", (typ == 2) & (kb > 1) & (q < 0)), ("UPD K=512 quarter", (typ == 2) & (kb > 1) & (q >= 0)), ("started sum quarter", typ == 5))
busy = 0.0
for name, m in classes:
    if not m.any():
        continue
    wait, body, pub = t[m, 1] - t[m, 0], t[m, 2] - t[m, 1], t[m, 3] - t[m, 2]
    busy += body.sum() + pub.sum()
    print("  %-18s %6d tasks  wait %7.2f us (max %7.1f)  body %6.2f us  publish %5.2f us   sum(body+publish) %9.1f us" % (name, m.sum(), wait.mean(), wait.max(), body.mean(), pub.mean(), (body + pub).sum()))
nwg = len(np.unique(wg))
print("  workgroups seen %d; busy (body + publish) %.1f us per workgroup = %.2f of the makespan; waiting %.1f us per workgroup" % (nwg, busy / nwg, busy / nwg / t[:, 3].max(), (t[:, 1] - t[:, 0]).sum() / nwg))
pm = (typ == 0) | (typ == 4)
order = np.argsort(tasks[pm, 1])
ps, pe = t[pm, 1][order], t[pm, 3][order]
print("  diagonal blocks: first starts %.1f us, last published %.1f us; mean step (published k -> published k+1) %.2f us" % (ps[0], pe[-1], np.diff(pe).mean()))
gaps = ps[1:] - pe[:-1]
print("  between two diagonal blocks (published k -> started k+1): mean %.2f us, at panel boundaries %.2f us, inside panels %.2f us"
      % (gaps.mean(), gaps[ob - 1::ob].mean(), np.delete(gaps, np.arange(ob - 1, len(gaps), ob)).mean()))
lib.gp_model_destroy(h)
ctx.close()
