"""Where one site iteration of ep_block_kernel (the one-barrier-per-site form: run with GPCORE_EP_BLOCK=0) goes (lab): needs a library built with -DEP_STAMPS
   (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DEP_STAMPS -o gp_algos_amd/libgpcore.so gp_algos_amd/csrc/*.hip)."""
import ctypes as C
import os
import sys

os.environ.setdefault("GPCORE_EP_BLOCK", "0")   # the stamps live in the one-barrier-per-site kernel

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from gp_algos_amd import core, synth  # noqa: E402

p = synth.config_c4(4096, 8)
ctx = core.Context(0)
K = ctx.gram_rbf(p["X"], p["theta"], full=True)
ep = core.EpClassifierState(ctx, K, p["y"])
ep.sweep(3)
ctx.sync()
st = (C.c_ulonglong * 512)()
ctx._lib.gp_debug_ep_stamps.restype = C.c_int
assert ctx._lib.gp_debug_ep_stamps(st) == 0
a = np.array(list(st), dtype=np.float64).reshape(128, 4)
chain = a[:, 1] - a[:, 0]
rowend = a[:, 2] - a[:, 0]
barrier_exit = a[:, 3] - a[:, 0]
period = np.diff(a[:, 0])
print("ticks per site iteration (scalar lane start -> next start): mean %.0f  min %.0f  max %.0f" % (period.mean(), period.min(), period.max()))
print("scalar lane start -> its results published: mean %.0f" % chain.mean())
print("scalar lane start -> last row thread reaches the barrier: mean %.0f" % rowend.mean())
print("scalar lane start -> scalar lane leaves the barrier: mean %.0f" % barrier_exit.mean())
for t in (1, 5, 14, 15, 16, 17, 30, 31, 32, 64, 100, 126):
    print("t=%3d chain %5.0f  row-end %5.0f  barrier-exit %5.0f  next-start %5.0f" % (t, chain[t], rowend[t], barrier_exit[t], period[t] if t < 127 else -1))
