"""Gram kernels alone: lower-triangular symmetric Gram at n and the cross-Gram of a posterior batch (HIP events per launch).
usage: python tools/gram_perf.py [n] [m] [d] [length-scale factor]   (GPCORE_GRAM_MFMA=0 selects the per-pair form; a factor of 0.3 puts
every point beyond |z|^2 = 64 from the first one, i.e. the unit kernel on its per-pair path)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gp_algos_amd import _lib as L, synth
from gp_algos_amd.core import Context
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
p = synth.regression(n, d, m, 11, 12, 13, synth.ard_theta(d, 1.5, scale, 0.1))
ctx = Context(0); lib = ctx._lib
dX, dy, dXs = ctx.upload(p["X"]), ctx.upload(p["y"]), ctx.upload(p["Xs"])
dK = ctx.dev_alloc(8 * n * n)
theta = L.f64(p["theta"])
for uplo, name, nbytes in ((L.GP_LOWER, "lower", 8.0 * n * (n + 1) / 2 + 8.0 * n * d), (L.GP_FULL, "full", 8.0 * n * n + 8.0 * n * d)):
    for _ in range(3):
        ctx.check(lib.gp_gram_rbf_dev(ctx.h, dX, n, d, n, L.dptr(theta), dK, n, uplo))
    ctx.sync(); ctx.profile(1 << L.GP_PROF_GRAM)
    for _ in range(10):
        ctx.check(lib.gp_gram_rbf_dev(ctx.h, dX, n, d, n, L.dptr(theta), dK, n, uplo))
    ctx.profile(0); k, ms, work = ctx.profile_read(L.GP_PROF_GRAM)
    print("scale %.2f " % scale, end=""); print("gram %-5s n=%d d=%d: %.1f us/launch -> %.2f TB/s (%.0f MB algorithmic)" % (name, n, d, ms / k * 1e3, nbytes / (ms / k * 1e-3) / 1e12, nbytes / 1e6))
h, info = C.c_void_p(), C.c_int()
ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, d, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)))
dmean, dvar = ctx.dev_alloc(8 * m), ctx.dev_alloc(8 * m)
ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar)); ctx.sync()
ctx.profile(1 << L.GP_PROF_GRAM)
for _ in range(3):
    ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))
ctx.profile(0); k, ms, work = ctx.profile_read(L.GP_PROF_GRAM)
print("cross-gram m=%d x n=%d: %.1f us/launch -> %.2f TB/s (%d launches)" % (m, n, ms / k * 1e3, work / k / (ms / k * 1e-3) / 1e12, k))
